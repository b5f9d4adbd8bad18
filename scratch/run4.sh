set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_n0
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES -d $R/gpurun_out/pmc_n0/a -o a --output-format csv -- $R/tools/nbench --only N0 --mode fwd --variants 0 --reps 3 > $R/gpurun_out/pmc_n0/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $R/gpurun_out/pmc_n0/b -o b --output-format csv -- $R/tools/nbench --only N0 --mode fwd --variants 0 --reps 3 > $R/gpurun_out/pmc_n0/b.log 2>&1
ls -R $R/gpurun_out/pmc_n0 | head -30
