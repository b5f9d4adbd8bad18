mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "stem or norm or stats or instance" 2>&1 | tail -4 > gpurun_out/t_norm.log
timeout 300 python bench.py --no-cpu-baseline --infer-steps 0 2>/dev/null | tail -1 > gpurun_out/b_inst.json
B="python bench.py --no-instrument --steps 30 --warmup 10 --no-cpu-baseline --infer-steps 0"
for i in 1 2; do timeout 200 $B 2>&1 | tail -1 > gpurun_out/b_fin_$i.json; done
