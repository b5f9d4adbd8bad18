O=gpurun_out/plumb; mkdir -p $O
B="python bench.py --no-instrument --no-cpu-baseline --infer-steps 0 --steps 100 --warmup 20 --workload plumbing-20k --levels 1 --batch 1"
for i in 1 2; do
timeout 100 $B 2>/dev/null | tail -1 > $O/base_$i.json
FC_STEM_COL=0 timeout 100 $B 2>/dev/null | tail -1 > $O/nocol_$i.json
FC_DGRAD_TRANSPOSE=1 timeout 100 $B 2>/dev/null | tail -1 > $O/nowt_$i.json
timeout 100 $B --no-wgrad-overlap 2>/dev/null | tail -1 > $O/nooverlap_$i.json
FC_FLAGS=0x20000 timeout 100 $B 2>/dev/null | tail -1 > $O/r1kernel_$i.json
FC_FLAGS=0x420000 timeout 100 $B 2>/dev/null | tail -1 > $O/r1kernel_noglds_$i.json
done
timeout 200 python tools/hostprof.py --workload plumbing-20k --levels 1 --batch 1 2>&1 | grep -E "synced|enqueue" > $O/hostprof.log
