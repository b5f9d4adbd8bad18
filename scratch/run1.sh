set -x
mkdir -p gpurun_out
timeout 300 tools/nbench --mode wgrad --reps 20 > gpurun_out/nb_wgrad_p.log 2>&1
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "variants_behind_flags or mfma" 2>&1 | tail -5 > gpurun_out/t_ops.log
for i in 1 2; do
FC_FLAGS=0x10000 timeout 300 python bench.py --no-instrument --steps 30 --warmup 10 --no-cpu-baseline --infer-steps 0 2>&1 | tail -1 > gpurun_out/b_r1w_$i.json
timeout 300 python bench.py --no-instrument --steps 30 --warmup 10 --no-cpu-baseline --infer-steps 0 2>&1 | tail -1 > gpurun_out/b_pw_$i.json
done
