#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5g7
mkdir -p $O
B="--no-cpu-baseline --infer-steps 0 --no-force-dp --no-fp32-route --no-extras --no-instrument"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B > $O/b_$name.json 2> $O/b_$name.err; python -c "import json;d=json.load(open('$O/b_$name.json'));print('$name',d['value'],d['ms_per_step'])"; }
run base A=0
run bn64_1k FC_PAIR_BN64_ROWS=1024
run bn64_4k FC_PAIR_BN64_ROWS=4096
run bn64_16k FC_PAIR_BN64_ROWS=16384
run base2 A=0
timeout 600 python -m pytest tests/test_gpu_exec.py -x -q -k train_step_through > $O/t.log 2>&1; echo "3-step rc=$?"
