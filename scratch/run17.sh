R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final2
mkdir -p $O
cd $R
C="--no-cpu-baseline --infer-steps 0 --steps 12 --warmup 4 --probe-every 6"
timeout 300 python bench.py $C --batch 4 2>/dev/null | tail -1 > $O/w_scannet_b4.json
timeout 300 python bench.py $C --batch 16 2>/dev/null | tail -1 > $O/w_scannet_b16.json
timeout 400 python bench.py $C --batch 32 2>/dev/null | tail -1 > $O/w_scannet_b32.json
timeout 300 python bench.py $C --workload sunrgbd-100k 2>/dev/null | tail -1 > $O/w_sunrgbd.json
timeout 400 python bench.py $C --workload s3dis-500k --batch 2 2>/dev/null | tail -1 > $O/w_s3dis.json
timeout 400 python bench.py $C --voxel-size 0.01 --batch 4 2>/dev/null | tail -1 > $O/w_scannet_1cm.json
timeout 300 python bench.py $C --levels 2 2>/dev/null | tail -1 > $O/w_scannet_2lev.json
timeout 300 python bench.py $C --workload plumbing-20k --levels 1 --batch 1 2>/dev/null | tail -1 > $O/w_plumbing.json
FC_DIST_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline > $O/dp2.json 2> $O/dp2.err
