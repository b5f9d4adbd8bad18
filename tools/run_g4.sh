#!/bin/bash
# r5 call 4: full GPU suite on the fused build, A/B, kernel trace
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5g4
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > $O/t_all.log 2>&1
echo "all gpu tests rc=$?"; tail -6 $O/t_all.log; grep "worst gradient\|IoU gradient\|two ranks x\|bf16 fast mode,\|bench.py --gpus 2" $O/t_all.log | head -20
B="--no-cpu-baseline --infer-steps 0 --no-force-dp --no-fp32-route --no-extras --no-instrument"
for f in 1 0 1 0; do
  FC_BN_FUSE=$f timeout 300 python bench.py $B > $O/b_f$f.json 2> $O/b_f$f.err
  python -c "import json;d=json.load(open('$O/b_f$f.json'));print('B=8 fuse=$f',d['value'],d['ms_per_step'],d['config']['final_loss'])"
done
for f in 1 0; do
  FC_BN_FUSE=$f timeout 300 python bench.py $B --batch 2 > $O/b2_f$f.json 2> $O/b2_f$f.err
  python -c "import json;d=json.load(open('$O/b2_f$f.json'));print('B=2 fuse=$f',d['value'],d['ms_per_step'])"
  FC_BN_FUSE=$f timeout 300 python bench.py $B --batch 4 > $O/b4_f$f.json 2> $O/b4_f$f.err
  python -c "import json;d=json.load(open('$O/b4_f$f.json'));print('B=4 fuse=$f',d['value'],d['ms_per_step'])"
done
cd /tmp
for m in "" "--no-wgrad-overlap"; do
  tag=ov; [ -n "$m" ] && tag=one
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$tag -o r5c -- python $GRAFT_REPO_ROOT/bench.py $B $m > $GRAFT_REPO_ROOT/$O/prof_$tag.log 2>&1
  f=$(find $GRAFT_REPO_ROOT/$O/prof_$tag -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/$O/ks_$tag.csv
  rm -rf $GRAFT_REPO_ROOT/$O/prof_$tag
done
cd $GRAFT_REPO_ROOT
python tools/kernel_stats.py 30 $O/ks_ov.csv $O/ks_one.csv > $O/ks.md
head -36 $O/ks.md
