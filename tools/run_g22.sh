#!/bin/bash
# the committed bench line: the default `python bench.py` as the SECOND process on its box (the first process of a fresh box falls into the
# slow mode of profiles/r5_notes.md section 11 four times out of ten)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5g22
mkdir -p $O
python bench.py --no-cpu-baseline --infer-steps 0 --no-fp32-route --no-extras --no-force-dp > $O/first.json 2> $O/err.log
python -c "
import json;d=json.load(open('$O/first.json')); print('first process: main', d['ms_per_step'])"
timeout 600 python bench.py > $O/r5_bench_n1.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json;d=json.load(open('$O/r5_bench_n1.json'));c=d['config'];r=d['roofline']
print('value',d['value'],d['ms_per_step'],'roofline',r['achieved'],r['frac'],r['avg_launch_us'],'ab',{k:(v['frac'],v['avg_launch_us']) for k,v in r['bn_epilogue_ab'].items() if isinstance(v,dict)})
for k in ('bf16_fast_mode','literal_1cm','two_scales','sunrgbd','s3dis','config4_per_gpu','forced_dp_n1','fp32_mfma_route','inference','inference_pipelined','fwd_bwd_only'): print(k, {kk:vv for kk,vv in (c.get(k) or {}).items() if kk in ('value','ms_per_step','ms','scenes_per_s','ms_per_batch','error')})
print('cpu', d['cpu_baseline'].get('value'), c['kernel_source_sha16'])"
