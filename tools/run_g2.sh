#!/bin/bash
# r5 call 2: batched map planning (tests + A/B), host profiles, kernel trace of the current build
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5g2
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -k "batched_map_planning or forward_train_parity or pruned or config5 or simple_test" > $O/t_plan.log 2>&1
echo "plan tests rc=$?"; tail -4 $O/t_plan.log
timeout 600 python -m pytest tests/test_gpu_exec.py tests/test_gpu_ops.py -x -q > $O/t_ops.log 2>&1
echo "ops+exec tests rc=$?"; tail -3 $O/t_ops.log
for pb in 1 0; do
  FC_PLAN_BATCH=$pb timeout 300 python tools/hostprof.py --batches 2,8 > $O/host_scannet_pb$pb.txt 2>&1
  FC_PLAN_BATCH=$pb timeout 300 python tools/hostprof.py --batches 2 --cprofile 2 --workload s3dis-500k > $O/host_s3dis_pb$pb.txt 2>&1
  grep -A2 "^=== B" $O/host_scannet_pb$pb.txt | grep -v "^--"
  grep -A2 "^=== B" $O/host_s3dis_pb$pb.txt | grep -v "^--"
  FC_PLAN_BATCH=$pb timeout 300 python bench.py --no-cpu-baseline --infer-steps 0 --no-force-dp --no-fp32-route --no-extras --no-instrument > $O/b_pb$pb.json 2> $O/b_pb$pb.err
  python -c "import json;d=json.load(open('$O/b_pb$pb.json'));print('B=8 plan_batch=$pb',d['value'],d['ms_per_step'])"
  FC_PLAN_BATCH=$pb timeout 300 python bench.py --no-cpu-baseline --infer-steps 0 --no-force-dp --no-fp32-route --no-extras --no-instrument --batch 2 > $O/b_pb${pb}_b2.json 2> $O/b_pb${pb}_b2.err
  python -c "import json;d=json.load(open('$O/b_pb${pb}_b2.json'));print('B=2 plan_batch=$pb',d['value'],d['ms_per_step'])"
  FC_PLAN_BATCH=$pb timeout 300 python bench.py --no-cpu-baseline --infer-steps 0 --no-force-dp --no-fp32-route --no-extras --no-instrument --batch 2 --workload s3dis-500k > $O/b_pb${pb}_s3.json 2> $O/b_pb${pb}_s3.err
  python -c "import json;d=json.load(open('$O/b_pb${pb}_s3.json'));print('s3dis B=2 plan_batch=$pb',d['value'],d['ms_per_step'])"
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o r5a -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --infer-steps 0 --no-force-dp --no-fp32-route --no-extras --no-instrument > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" | head -3
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/ks_r5a.csv
find $O/prof -name "*.csv" ! -name "*kernel_stats*" -delete; find $O/prof -name "*.db" -delete
python tools/kernel_stats.py 25 $O/ks_r5a.csv $O/ks_r5a.csv | head -40
