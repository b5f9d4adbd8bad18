#!/bin/bash
# (FC_X6_PIPE / FC_X6_BUF were switches of the experiment builds of this call; the product library has neither: buffer addressing is the default, flags bit27 = flat)
# A/B of the in-wave pipelined stage loop of k_conv_x6 (FC_X6_PIPE bit 0: 128-column tiles, bit 1: 64-column tiles): bit-identity, then timing
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5g14
mkdir -p $O
for m in 0 3; do
  FC_X6_PIPE=$m timeout 300 python tools/determinism.py --points 100000 --scenes 2 --reps 2 --bench-streams 2>&1 | tail -1 | cut -c1-200 | sed "s/^/pipe $m: /"
done
B="--no-cpu-baseline --infer-steps 0 --no-force-dp --no-fp32-route --no-extras --no-instrument --steps 20 --warmup 5"
for rep in 1 2; do
  for m in 0 1 2 3; do
    FC_X6_PIPE=$m python bench.py $B > $O/b_${m}_$rep.json 2> $O/err_${m}.log
    python -c "
import json;d=json.load(open('$O/b_${m}_$rep.json'))
print('pipe $m rep $rep:', d['value'], d['ms_per_step'])"
  done
done
