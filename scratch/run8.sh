mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -m gpu 2>&1 | tail -5 > gpurun_out/t_ops4.log
for i in 1 2; do
FC_DGRAD_TRANSPOSE=1 timeout 300 python bench.py --no-instrument --steps 30 --warmup 10 --no-cpu-baseline --infer-steps 0 2>&1 | tail -1 > gpurun_out/b_nowt_$i.json
timeout 300 python bench.py --no-instrument --steps 30 --warmup 10 --no-cpu-baseline --infer-steps 0 2>&1 | tail -1 > gpurun_out/b_wt_$i.json
done
