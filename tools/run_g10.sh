#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5g10
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_dist.py -x -q -s > $O/t_dist.log 2>&1; echo "dist tests rc=$?"; grep "passed\|failed\|two ranks x\|config-4 per-GPU\|bench.py --gpus 2" $O/t_dist.log | head
timeout 1200 python bench.py > $O/r5_bench_n1.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json;d=json.load(open('$O/r5_bench_n1.json'));c=d['config'];r=d['roofline']
print('value',d['value'],d['ms_per_step'],'roofline',r['achieved'],r['frac'],r['avg_launch_us'],'unfused',r.get('without_bn_epilogues'))
print(c['kernel_source_sha16'])"
FC_BN_FUSE=0 timeout 600 python bench.py --no-cpu-baseline --infer-steps 0 --no-force-dp --no-fp32-route --no-extras > $O/b_nofuse.json 2> $O/b_nofuse.err
python -c "
import json;d=json.load(open('$O/b_nofuse.json'));r=d['roofline']
print('FC_BN_FUSE=0: value',d['value'],d['ms_per_step'],'roofline',r['achieved'],r['frac'],r['avg_launch_us'])"
