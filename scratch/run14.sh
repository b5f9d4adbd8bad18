mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "stem or generic_fma" 2>&1 | tail -5 > gpurun_out/t_stem.log
B="python bench.py --no-instrument --steps 30 --warmup 10 --no-cpu-baseline --infer-steps 0"
for i in 1 2; do
FC_STEM_COL=0 timeout 200 $B 2>&1 | tail -1 > gpurun_out/b_nocol_$i.json
timeout 200 $B 2>&1 | tail -1 > gpurun_out/b_col_$i.json
done
