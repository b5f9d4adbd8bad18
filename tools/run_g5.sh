#!/bin/bash
# r5 call 5: routing thresholds of the convolution (pair lists vs mask-sorted tables, offset split), 3-step test
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5g5
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_exec.py -x -q -s -k "train_step_through" > $O/t_exec.log 2>&1
echo "3-step test rc=$?"; grep "parameters after\|losses\|passed\|failed" $O/t_exec.log | head
B="--no-cpu-baseline --infer-steps 0 --no-force-dp --no-fp32-route --no-extras --no-instrument"
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py $B > $O/b_$name.json 2> $O/b_$name.err
  python -c "import json;d=json.load(open('$O/b_$name.json'));print('$name B=8',d['value'],d['ms_per_step'])"
}
run base A=0
run pair8k FC_PAIR_CONV_ROWS=8192 FC_SORT_MIN_ROWS=4096
run pair2k FC_PAIR_CONV_ROWS=2048 FC_SORT_MIN_ROWS=2048
run pair0 FC_PAIR_CONV_ROWS=0 FC_SORT_MIN_ROWS=512
run split512 FC_SPLIT_TILES=512
run split768 FC_SPLIT_TILES=768
run split256 FC_SPLIT_TILES=256
run base2 A=0
run bn4m FC_BN_SMALL_ELEMS=4194304
run bn256k FC_BN_SMALL_ELEMS=262144
