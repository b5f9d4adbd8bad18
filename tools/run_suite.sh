#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5suite
mkdir -p $O
SECONDS=0
timeout 1500 python -m pytest tests -x -q -m gpu --durations=12 > $O/t_all.log 2>&1; echo "all gpu tests rc=$? in $SECONDS s"; tail -22 $O/t_all.log | grep -v Warning
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
