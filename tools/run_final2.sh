#!/bin/bash
# r5 final (second take, after the buffer-addressing kernels): every profile of the round on ONE build, one box, back to back, then the
# operator / executor / two-rank tests and the full-size model tests (outputs under gpurun_out/r5final2; copied into profiles/)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5final2
mkdir -p $O
export TMPDIR=/tmp
python -c "from fcaf3d_amd.build import source_hash; print('kernel source', source_hash())" | tee $O/source_hash.txt
B="--no-cpu-baseline --infer-steps 0 --no-force-dp --no-fp32-route --no-extras --no-instrument"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- python $GRAFT_REPO_ROOT/bench.py $B --steps 3 --warmup 1 > $O/pmc_$c.log 2>&1
  echo "pmc $c rc=$?"
done
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS"
P3="SQ_INSTS_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SALU SQ_WAVES SQ_INST_LEVEL_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_WR"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/sq_$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py $B --steps 3 --warmup 1 > $O/sq_$i.log 2>&1
  echo "sq pass $i rc=$?"
done
cd $GRAFT_REPO_ROOT
F=$(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/traffic_from_pmc.py $F $W $O/r5_traffic.json > $O/traffic.log 2>&1; cp $O/r5_traffic.json profiles/r5_traffic.json
python tools/sq_from_pmc.py $O/r5_conv_pmc.json $O/r5_conv_pmc_table.md $(find $O/sq_1 $O/sq_2 $O/sq_3 -name "*counter_collection.csv" | sort) > $O/sq.log 2>&1; echo "sq fold rc=$?"
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/sq_1 $O/sq_2 $O/sq_3
tail -6 $O/traffic.log | cut -c1-200
timeout 1200 python bench.py > $O/r5_bench_n1.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json;d=json.load(open('$O/r5_bench_n1.json'));c=d['config'];r=d['roofline']
print('value',d['value'],d['ms_per_step'],'roofline',r['achieved'],r['frac'],r['avg_launch_us'],'traffic',r['traffic'],'alg',r['algorithmic_bytes_per_launch'],'ab',{k:(v['frac'],v['avg_launch_us']) for k,v in r['bn_epilogue_ab'].items() if isinstance(v,dict)})
for k in ('bf16_fast_mode','literal_1cm','two_scales','sunrgbd','s3dis','config4_per_gpu','forced_dp_n1','fp32_mfma_route','inference','inference_pipelined','fwd_bwd_only'): print(k, {kk:vv for kk,vv in (c.get(k) or {}).items() if kk in ('value','ms_per_step','ms','scenes_per_s','ms_per_batch','error')})
print('cpu', d['cpu_baseline'].get('value'), d['cpu_baseline'].get('cores'), c['kernel_source_sha16'])"
cd /tmp
for m in "" "--no-wgrad-overlap"; do
  tag=ov; [ -n "$m" ] && tag=one
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o r5 -- python $GRAFT_REPO_ROOT/bench.py $B $m > $O/prof_$tag.log 2>&1
  f=$(find $O/prof_$tag -name "*kernel_stats.csv" | head -1)
  [ "$tag" = ov ] && cp $f $O/r5_kernel_stats.csv || cp $f $O/r5_kernel_stats_no_overlap.csv
  rm -rf $O/prof_$tag
done
cd $GRAFT_REPO_ROOT
python tools/kernel_stats.py 30 $O/r5_kernel_stats.csv $O/r5_kernel_stats_no_overlap.csv > $O/r5_kernel_stats_tables.md; head -16 $O/r5_kernel_stats_tables.md | cut -c1-160
timeout 400 python tools/hostprof.py --batches 2,4,8 > $O/hostprof_scannet.txt 2>&1; grep -A2 "^=== B" $O/hostprof_scannet.txt | grep -v "^--"
timeout 200 python tools/hostprof.py --batches 2 --workload s3dis-500k > $O/hostprof_s3dis.txt 2>&1; grep -A2 "^=== B" $O/hostprof_s3dis.txt
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_exec.py tests/test_gpu_dist.py -x -q > $O/t_ops_exec_dist.log 2>&1; echo "ops+exec+dist rc=$? in $SECONDS s"; tail -2 $O/t_ops_exec_dist.log
SECONDS=0
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "kw4 or config5_backward or simple_test_parity or three_step or depth50 or iou or pruning_live" > $O/t_model.log 2>&1; echo "model subset rc=$? in $SECONDS s"; tail -2 $O/t_model.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
