#!/bin/bash
# the slow mode of the FIRST process on a fresh box: does it depend on how the host waits (HSA_ENABLE_INTERRUPT=0: the runtime polls its signals)?
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5g21
mkdir -p $O
B="--no-cpu-baseline --infer-steps 0 --no-fp32-route --no-extras"
i=0
for e in "$1" "" "$1"; do
  i=$((i+1))
  env $e python bench.py $B > $O/b_$i.json 2> $O/err.log
  python -c "
import json;d=json.load(open('$O/b_$i.json'));c=d['config']
g=lambda k:(c.get(k) or {}).get('ms_per_step')
print('process $i [$e]: main', d['ms_per_step'], 'forced_dp', g('forced_dp_n1'), 'cfg4', g('config4_per_gpu'), 'frac', (d.get('roofline') or {}).get('frac'))"
done
