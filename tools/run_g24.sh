#!/bin/bash
# where the host's 12.5 ms of enqueue per 8-scene step go (cProfile over three steps) — the step is within 15 % of host-bound (bench: host_enqueue_ms_per_step 17.5 of 20.5)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5g24
mkdir -p $O
timeout 55 python tools/hostprof.py --batches 8 --cprofile 8 > $O/hostprof_cprofile_b8.txt 2>&1; echo "rc=$?"
grep -A3 "^=== B" $O/hostprof_cprofile_b8.txt | head -6
