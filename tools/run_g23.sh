#!/bin/bash
# bench.py with the host diagnostics (config.host, host_enqueue_ms_per_step): the default line's fields, and the two-rank contract test
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5g23
mkdir -p $O
python bench.py --no-cpu-baseline --infer-steps 0 --no-fp32-route --extra-steps 3 > $O/b.json 2> $O/err.log; echo "bench rc=$?"
python -c "
import json;d=json.load(open('$O/b.json'));c=d['config']
print('main', d['value'], d['ms_per_step'], c['host'])
for k in ('literal_1cm','two_scales','sunrgbd','s3dis'): print(k, c[k]['ms_per_step'], c[k]['host_enqueue_ms_per_step'])"
timeout 200 python -m pytest tests/test_gpu_dist.py -x -q -k "bench_two_ranks" 2>&1 | tail -2
