mkdir -p gpurun_out
python scratch/microbench.py 2>&1 | grep -E " us|rows" > gpurun_out/micro.log
echo "## FC_STEM_RPS=1024" >> gpurun_out/micro.log
FC_STEM_RPS=1024 python scratch/microbench.py 2>&1 | grep -E "STEM_COL True" >> gpurun_out/micro.log
echo "## FC_STEM_RPS=512" >> gpurun_out/micro.log
FC_STEM_RPS=512 python scratch/microbench.py 2>&1 | grep -E "STEM_COL True" >> gpurun_out/micro.log
echo "## FC_STEM_RPS=256" >> gpurun_out/micro.log
FC_STEM_RPS=256 python scratch/microbench.py 2>&1 | grep -E "STEM_COL True" >> gpurun_out/micro.log
