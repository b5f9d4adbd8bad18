#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5g6
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_exec.py -x -q > $O/t_ops.log 2>&1
echo "ops+exec rc=$?"; tail -3 $O/t_ops.log
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "forward_train_parity or config5 or trajectory or train_step" > $O/t_model.log 2>&1
echo "model rc=$?"; tail -3 $O/t_model.log
B="--no-cpu-baseline --infer-steps 0 --no-force-dp --no-fp32-route --no-extras --no-instrument"
for i in 1 2; do
  timeout 300 python bench.py $B > $O/b8_$i.json 2> $O/b8_$i.err
  python -c "import json;d=json.load(open('$O/b8_$i.json'));print('B=8',d['value'],d['ms_per_step'],d['config']['final_loss'])"
done
timeout 300 python bench.py $B --batch 2 > $O/b2.json 2> $O/b2.err
python -c "import json;d=json.load(open('$O/b2.json'));print('B=2',d['value'],d['ms_per_step'])"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o r5d -- python $GRAFT_REPO_ROOT/bench.py $B --no-wgrad-overlap > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
f=$(find $GRAFT_REPO_ROOT/$O/prof -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/$O/ks_one.csv; rm -rf $GRAFT_REPO_ROOT/$O/prof
cd $GRAFT_REPO_ROOT
python - <<PY
import csv
rows=list(csv.DictReader(open('$O/ks_one.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)/30/1e6
print('one-stream kernel sum ms/step', round(tot,2), 'launches', sum(int(r['Calls']) for r in rows)/30)
for r in rows:
    n=r['Name']
    if any(k in n for k in ('k_stats','k_seg_meanvar','k_bn2_','k_sum_pairs','k_sum_parts','k_norm_','k_head_','k_bn1_')):
        print(f"{n[:36]:36s} calls/step {int(r['Calls'])/30:6.1f} avg {float(r['AverageNs'])/1e3:7.1f} max {float(r['MaxNs'])/1e3:7.1f} ms/step {float(r['TotalDurationNs'])/30/1e6:.3f}")
PY
