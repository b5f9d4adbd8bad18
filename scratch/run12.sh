mkdir -p gpurun_out
timeout 300 python tools/inferprof.py > gpurun_out/inferprof.log 2>&1
