#!/bin/bash
# r5 final: every profile of the round on ONE build, one box, back to back (outputs under gpurun_out/r5final; copied into profiles/)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5final
mkdir -p $O
export TMPDIR=/tmp
python -c "from fcaf3d_amd.build import source_hash; print('kernel source', source_hash())" | tee $O/source_hash.txt
B="--no-cpu-baseline --infer-steps 0 --no-force-dp --no-fp32-route --no-extras --no-instrument"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- python $GRAFT_REPO_ROOT/bench.py $B --steps 3 --warmup 1 > $O/pmc_$c.log 2>&1
  echo "pmc $c rc=$?"
done
cd $GRAFT_REPO_ROOT
F=$(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/traffic_from_pmc.py $F $W profiles/r5_traffic.json > $O/traffic.log 2>&1; cp profiles/r5_traffic.json $O/
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
tail -12 $O/traffic.log
timeout 1200 python bench.py > $O/r5_bench_n1.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json;d=json.load(open('$O/r5_bench_n1.json'));c=d['config'];r=d['roofline']
print('value',d['value'],d['ms_per_step'],'roofline',r['achieved'],r['frac'],'traffic',r['traffic'],'alg',r['algorithmic_bytes_per_launch'])
for k in ('bf16_fast_mode','literal_1cm','two_scales','sunrgbd','s3dis','config4_per_gpu','forced_dp_n1','fp32_mfma_route','inference','inference_pipelined','fwd_bwd_only'): print(k, {kk:vv for kk,vv in (c.get(k) or {}).items() if kk in ('value','ms_per_step','ms','scenes_per_s','ms_per_batch','error')})
print('cpu', d['cpu_baseline'].get('value'), d['cpu_baseline'].get('cores'))"
cd /tmp
for m in "" "--no-wgrad-overlap"; do
  tag=ov; [ -n "$m" ] && tag=one
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o r5 -- python $GRAFT_REPO_ROOT/bench.py $B $m > $O/prof_$tag.log 2>&1
  f=$(find $O/prof_$tag -name "*kernel_stats.csv" | head -1)
  [ "$tag" = ov ] && cp $f $O/r5_kernel_stats.csv || cp $f $O/r5_kernel_stats_no_overlap.csv
  rm -rf $O/prof_$tag
done
cd $GRAFT_REPO_ROOT
python tools/kernel_stats.py 30 $O/r5_kernel_stats.csv $O/r5_kernel_stats_no_overlap.csv > $O/r5_kernel_stats_tables.md; head -16 $O/r5_kernel_stats_tables.md
timeout 600 python tools/hostprof.py --batches 2,4,8 > $O/hostprof_scannet.txt 2>&1; grep -A2 "^=== B" $O/hostprof_scannet.txt | grep -v "^--"
timeout 300 python tools/hostprof.py --batches 2 --workload s3dis-500k > $O/hostprof_s3dis.txt 2>&1; grep -A2 "^=== B" $O/hostprof_s3dis.txt
timeout 300 python tools/determinism.py --points 100000 --scenes 8 --reps 6 > $O/determinism.txt 2>&1; tail -3 $O/determinism.txt
timeout 1800 python -m pytest tests -q -m gpu > $O/t_all.log 2>&1; echo "all gpu tests rc=$?"; tail -4 $O/t_all.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
