mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_dist.py -q -x -m gpu -k "trajectory or dist or rank" 2>&1 | tail -5 > gpurun_out/t_clip.log
B="python bench.py --no-instrument --steps 30 --warmup 10 --no-cpu-baseline --infer-steps 0"
for i in 1 2; do
FC_FUSED_CLIP=0 timeout 200 $B 2>&1 | tail -1 > gpurun_out/b_noclipf_$i.json
timeout 200 $B 2>&1 | tail -1 > gpurun_out/b_clipf_$i.json
done
