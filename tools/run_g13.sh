#!/bin/bash
# A/B: wave priority in the STAGING part of k_conv_x6 (g_fc_prio mode 2) against the default (none), interleaved
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5g13
mkdir -p $O
B="--no-cpu-baseline --infer-steps 0 --no-force-dp --no-fp32-route --no-extras --no-instrument --steps 20 --warmup 5"
for rep in 1 2; do
  for m in 0 2; do
    if [ $m = 0 ]; then python bench.py $B > $O/b_${m}_$rep.json 2> $O/err.log; else FC_PRIO_MODE=$m python bench.py $B > $O/b_${m}_$rep.json 2> $O/err.log; fi
    python -c "
import json;d=json.load(open('$O/b_${m}_$rep.json'));r=d['roofline']
print('prio mode $m rep $rep:', d['value'], d['ms_per_step'], 'conv us', r['avg_launch_us'], r['frac'])"
  done
done
