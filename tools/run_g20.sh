#!/bin/bash
# how often do the 8-scene runs of one bench.py process fall into the slow mode, with the weight-gradient stream at normal (default) and at LOW priority?
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5g20
mkdir -p $O
for rep in 1 2; do
  for p in 0 1; do
    FC_WGRAD_PRIO=$p python bench.py --no-cpu-baseline --infer-steps 0 --no-fp32-route --extra-steps 8 > $O/b_${p}_$rep.json 2> $O/err.log
    python -c "
import json;d=json.load(open('$O/b_${p}_$rep.json'));c=d['config']
g=lambda k:(c.get(k) or {}).get('ms_per_step')
print('wgrad prio $p rep $rep: main', d['ms_per_step'], 'forced_dp', g('forced_dp_n1'), 'cfg4', g('config4_per_gpu'), 'bf16', g('bf16_fast_mode'), '1cm', g('literal_1cm'), '2sc', g('two_scales'), 'sun', g('sunrgbd'), 's3dis', g('s3dis'))"
  done
done
