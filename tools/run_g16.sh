#!/bin/bash
# (FC_X6_PIPE / FC_X6_BUF were switches of the experiment builds of this call; the product library has neither: buffer addressing is the default, flags bit27 = flat)
# A/B of buffer addressing in k_conv_x6 (FC_X6_BUF): bit-identity (gradient digest of a full forward + backward), then timing, then kernel-alone durations
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5g16
mkdir -p $O
for m in 0 1; do
  FC_X6_BUF=$m timeout 300 python tools/determinism.py --points 100000 --scenes 2 --reps 2 --bench-streams 2>&1 | tail -1 | cut -c1-200 | sed "s/^/buf $m: /"
done
B="--no-cpu-baseline --infer-steps 0 --no-force-dp --no-fp32-route --no-extras --no-instrument --steps 20 --warmup 5"
for rep in 1 2 3; do
  for m in 0 1; do
    FC_X6_BUF=$m python bench.py $B > $O/b_${m}_$rep.json 2> $O/err_${m}.log
    python -c "
import json;d=json.load(open('$O/b_${m}_$rep.json'))
print('buf $m rep $rep:', d['value'], d['ms_per_step'])"
  done
done
export TMPDIR=/tmp
cd /tmp
for m in 0 1; do
  FC_X6_BUF=$m rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$m -o p -- python $GRAFT_REPO_ROOT/bench.py $B --steps 10 --warmup 3 --no-wgrad-overlap > $O/prof_$m.log 2>&1
  f=$(find $O/prof_$m -name "*kernel_stats.csv" | head -1)
  echo "== buf $m"; grep "k_conv_x6" $f | sed 's/(float const.*X6Epi)"//' | cut -c1-120 | head -8
  cp $f $O/stats_$m.csv; rm -rf $O/prof_$m
done
