R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.log
timeout 600 python bench.py > $O/r2_bench_n1.json 2> $O/r2_bench_n1.err
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r2 --output-format csv -- python $R/bench.py --no-cpu-baseline --infer-steps 0 > $O/prof.log 2>&1)
find $O/prof -name "*kernel_trace.csv" -size +30M -delete
timeout 300 python tools/hostprof.py 2>&1 | grep -E "synced phase|enqueue-only" > $O/hostprof.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
