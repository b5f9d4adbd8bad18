#!/bin/bash
# is the slow mode of the forced 1-rank averager run (33.9 ms in the final take) reproducible?
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5g18
mkdir -p $O
for rep in 1 2 3; do
  python bench.py --no-cpu-baseline --infer-steps 0 --no-fp32-route --no-extras --no-instrument > $O/b_$rep.json 2> $O/err.log
  python -c "
import json;d=json.load(open('$O/b_$rep.json'));c=d['config']
print('rep $rep:', d['value'], d['ms_per_step'], 'forced_dp', (c.get('forced_dp_n1') or {}).get('ms_per_step'), 'cfg4', (c.get('config4_per_gpu') or {}).get('ms_per_step'))"
done
