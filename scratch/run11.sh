O=gpurun_out/prio2; mkdir -p $O
B="python bench.py --no-instrument --steps 30 --warmup 10 --no-cpu-baseline --infer-steps 0"
for i in 1 2 3; do
timeout 200 $B 2>&1 | tail -1 > $O/A_$i.json
FC_PRIO_MODE=1 timeout 200 $B 2>&1 | tail -1 > $O/B_$i.json
timeout 200 $B --priority-stream 2>&1 | tail -1 > $O/C_$i.json
FC_PRIO_MODE=1 timeout 200 $B --priority-stream 2>&1 | tail -1 > $O/D_$i.json
FC_PRIO_OFF=1 timeout 200 $B 2>&1 | tail -1 > $O/E_$i.json
timeout 200 $B --no-wgrad-overlap 2>&1 | tail -1 > $O/F_$i.json
done
