#!/bin/bash
# buffer addressing as the default: the new parity test, the operator / executor tests; A/B against the build with the loads pinned before the MFMAs
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5g17
mkdir -p $O
SECONDS=0
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "buffer_and_flat or conv or statistics or batchnorm_backward_sums" > $O/t_ops.log 2>&1; echo "ops rc=$? in $SECONDS s"; tail -2 $O/t_ops.log
B="--no-cpu-baseline --infer-steps 0 --no-force-dp --no-fp32-route --no-extras --no-instrument --steps 20 --warmup 5"
for rep in 1 2 3; do
  for m in buf early flat; do
    case $m in
      buf) python bench.py $B > $O/b_${m}_$rep.json 2> $O/err_${m}.log;;
      early) FC_LIB=$GRAFT_REPO_ROOT/tools/ko/FC_X6_EARLY/libfcaf3d_hip.so python bench.py $B > $O/b_${m}_$rep.json 2> $O/err_${m}.log;;
      flat) FC_FLAGS=0x8000000 python bench.py $B > $O/b_${m}_$rep.json 2> $O/err_${m}.log;;
    esac
    python -c "
import json;d=json.load(open('$O/b_${m}_$rep.json'))
print('$m rep $rep:', d['value'], d['ms_per_step'])"
  done
done
