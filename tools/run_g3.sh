#!/bin/bash
# r5 call 3: BN backward sums from the dgrad epilogue (tests + A/B), plan batching modes, s3dis after the prune fix, kernel trace
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5g3
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "epilogue or statistics or second_gradient or head_backward or bf16 or non_finite or offset_split or norm or bn" > $O/t_new.log 2>&1
echo "new tests rc=$?"; tail -4 $O/t_new.log
timeout 900 python -m pytest tests/test_gpu_exec.py -x -q > $O/t_exec.log 2>&1
echo "exec tests rc=$?"; tail -4 $O/t_exec.log; grep "worst gradient" $O/t_exec.log | head
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "batched_map_planning or config5 or pruned or forward_train_parity" > $O/t_model.log 2>&1
echo "model tests rc=$?"; tail -4 $O/t_model.log
B="--no-cpu-baseline --infer-steps 0 --no-force-dp --no-fp32-route --no-extras --no-instrument"
for v in "1 1" "1 0" "0 0" "1 2"; do
  set -- $v
  FC_BN_FUSE=$1 FC_PLAN_BATCH=$2 timeout 300 python bench.py $B > $O/b_f$1_p$2.json 2> $O/b_f$1_p$2.err
  python -c "import json;d=json.load(open('$O/b_f$1_p$2.json'));print('B=8 fuse=$1 plan=$2',d['value'],d['ms_per_step'],d['config']['final_loss'])"
  FC_BN_FUSE=$1 FC_PLAN_BATCH=$2 timeout 300 python bench.py $B --batch 2 > $O/b2_f$1_p$2.json 2> $O/b2_f$1_p$2.err
  python -c "import json;d=json.load(open('$O/b2_f$1_p$2.json'));print('B=2 fuse=$1 plan=$2',d['value'],d['ms_per_step'])"
  FC_BN_FUSE=$1 FC_PLAN_BATCH=$2 timeout 300 python bench.py $B --batch 2 --workload s3dis-500k > $O/s3_f$1_p$2.json 2> $O/s3_f$1_p$2.err
  python -c "import json;d=json.load(open('$O/s3_f$1_p$2.json'));print('s3dis B=2 fuse=$1 plan=$2',d['value'],d['ms_per_step'])"
done
FC_PLAN_BATCH=1 timeout 300 python bench.py $B --batch 4 --voxel-size 0.01 > $O/cm1.json 2> $O/cm1.err
python -c "import json;d=json.load(open('$O/cm1.json'));print('1cm B=4',d['value'],d['ms_per_step'])"
timeout 300 python tools/hostprof.py --batches 2 --cprofile 2 --workload s3dis-500k > $O/host_s3dis.txt 2>&1
grep -A2 "^=== B" $O/host_s3dis.txt
timeout 300 python tools/hostprof.py --batches 2,8 > $O/host_scannet.txt 2>&1
grep -A2 "^=== B" $O/host_scannet.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o r5b -- python $GRAFT_REPO_ROOT/bench.py $B > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/ks_r5b.csv
rm -rf $O/prof
python tools/kernel_stats.py 30 $O/ks_r5b.csv $O/ks_r5b.csv | head -36
