#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5g8
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_exec.py -x -q > $O/t.log 2>&1; echo "tests rc=$?"; tail -3 $O/t.log
B="--no-cpu-baseline --infer-steps 0 --no-force-dp --no-fp32-route --no-extras --no-instrument"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA > $O/b_$name.json 2> $O/b_$name.err; python -c "import json;d=json.load(open('$O/b_$name.json'));print('$name',d['value'],d['ms_per_step'],d['config']['final_loss'])"; }
run warm A=0
run head FC_LIB=$GRAFT_REPO_ROOT/scratch/lib_head.so
run persist1 FC_PERSIST=1
run persist0 FC_PERSIST=0
run head_b FC_LIB=$GRAFT_REPO_ROOT/scratch/lib_head.so
run persist1_b FC_PERSIST=1
EXTRA="--batch 2"
run head_B2 FC_LIB=$GRAFT_REPO_ROOT/scratch/lib_head.so
run persist1_B2 FC_PERSIST=1
EXTRA="--no-wgrad-overlap"
run head_one FC_LIB=$GRAFT_REPO_ROOT/scratch/lib_head.so
run persist1_one FC_PERSIST=1
run persist0_one FC_PERSIST=0
