#!/bin/bash
# first-process-on-a-fresh-box behaviour of the step time (hardware-queue sharing between the step's streams)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5q
mkdir -p $O
B="--no-cpu-baseline --infer-steps 0 --no-force-dp --no-fp32-route --no-extras"
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout 300 python bench.py $B > $O/q_$i.json 2> $O/q_$i.err
  python -c "import json;d=json.load(open('$O/q_$i.json'));print('$e',d['value'],d['ms_per_step'])"
done
