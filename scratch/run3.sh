set -x
mkdir -p gpurun_out
timeout 300 python tools/torch_ops.py --steps 2 > gpurun_out/torch_ops.log 2>&1
for i in 1 2; do
FC_FLAGS=0x400000 timeout 300 python bench.py --no-instrument --steps 30 --warmup 10 --no-cpu-baseline --infer-steps 0 2>&1 | tail -1 > gpurun_out/b_noglds_$i.json
timeout 300 python bench.py --no-instrument --steps 30 --warmup 10 --no-cpu-baseline --infer-steps 0 2>&1 | tail -1 > gpurun_out/b_glds_$i.json
done
