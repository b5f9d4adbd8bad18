set -x
mkdir -p gpurun_out
timeout 400 tools/nbench --mode fwd --variants 0 --reps 20 > gpurun_out/nb_glds.log 2>&1
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "variants_behind_flags" 2>&1 | tail -5 > gpurun_out/t_ops2.log
timeout 300 python tools/torch_ops.py --steps 2 > gpurun_out/torch_ops.log 2>&1
