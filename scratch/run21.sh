mkdir -p gpurun_out
rm -f gpurun_out/micro2.log
for r in 1024 2048 3072 4096 8192; do
echo "## FC_STEM_RPS=$r" >> gpurun_out/micro2.log
FC_STEM_RPS=$r python scratch/microbench.py 2>&1 | grep -E "STEM_COL True" >> gpurun_out/micro2.log
done
