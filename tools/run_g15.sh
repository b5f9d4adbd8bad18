#!/bin/bash
# (FC_X6_PIPE / FC_X6_BUF were switches of the experiment builds of this call; the product library has neither: buffer addressing is the default, flags bit27 = flat)
# kernel-alone durations (one stream) of the pipelined against the plain stage loop
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5g15
mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --infer-steps 0 --no-force-dp --no-fp32-route --no-extras --no-instrument --steps 10 --warmup 3 --no-wgrad-overlap"
cd /tmp
for m in 0 3; do
  FC_X6_PIPE=$m rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$m -o p -- python $GRAFT_REPO_ROOT/bench.py $B > $O/prof_$m.log 2>&1
  f=$(find $O/prof_$m -name "*kernel_stats.csv" | head -1)
  echo "== pipe $m"; grep "k_conv_x6" $f | cut -c1-200 | head -8
  cp $f $O/stats_$m.csv; rm -rf $O/prof_$m
done
