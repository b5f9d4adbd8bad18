#!/bin/bash
# r5: SQ counter passes over the default bench workload (three separate rocprofv3 runs, <= 8 SQ counters each) -> profiles/r5_conv_pmc.{json,md}
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5g12
mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --infer-steps 0 --no-force-dp --no-fp32-route --no-extras --no-instrument --steps 3 --warmup 1"
cd /tmp
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS"
P3="SQ_INSTS_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SALU SQ_WAVES SQ_INST_LEVEL_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_WR"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1)); SECONDS=0
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/pmc_$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py $B > $O/pmc_$i.log 2>&1
  echo "pass $i rc=$? in $SECONDS s"; tail -2 $O/pmc_$i.log | cut -c1-300
done
cd $GRAFT_REPO_ROOT
FILES=$(find $O -name "*counter_collection.csv" | sort)
python tools/sq_from_pmc.py $O/r5_conv_pmc.json $O/r5_conv_pmc_table.md $FILES > $O/sq.log 2>&1; echo "fold rc=$?"; tail -5 $O/sq.log | cut -c1-400
head -2 $(echo $FILES | cut -d' ' -f1) > $O/csv_head.txt
rm -rf $O/pmc_1 $O/pmc_2 $O/pmc_3
