#!/bin/bash
# The ONE lease script (r6; replaces the per-call run_g*.sh of r5): every gpurun call of the round goes through it.
#   gpurun --timeout T -- 'bash tools/lease.sh <tag> <job> [<job> ...]'
# Jobs run in the order given, output under gpurun_out/<tag>/.  A job is one of
#   bench[:args]     python bench.py [args]            -> bench.json (+ one-line summary)         (FIRST job = driver conditions)
#   quick[:args]     bench.py without the untimed extras -> quick<i>.json
#   equick:ENV[:args] the same under environment switches (VAR=VAL,VAR=VAL)
#   tests[:expr]     pytest -m gpu [-k expr]           -> tests.log
#   file:<path>      pytest -m gpu <path>              -> tests_<name>.log
#   smoke            __graft_entry__.smoke()
#   trace            rocprofv3 --kernel-trace --stats over a short bench -> trace/ + kernel_stats
#   trace1           the same with every kernel on ONE stream (--no-wgrad-overlap, FC_MAP_SYNC=1)
#   pmc:<counters>   rocprofv3 --pmc <counters> over a short bench (own pass, no tracing)
#   pmcfold:<tag>    FETCH / WRITE traffic passes + three SQ counter passes, folded -> <tag>_traffic.json, <tag>_conv_pmc.{json,md}
#   host[:args]      tools/hostprof.py --batches <args>   (default: 8 --lookahead)
#   py:<script args> python <script args>
#   sh:<command>     bash -c <command>            -> sh<i>.log (native benches: tools/nbench ...)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$O"
export TMPDIR=/tmp
uptime > "$O/uptime.txt"
i=0
for job in "$@"; do
  i=$((i + 1))
  kind=${job%%:*}; arg=""; [ "$kind" != "$job" ] && arg=${job#*:}
  SECONDS=0
  case $kind in
    bench)
      timeout 900 python bench.py $arg > "$O/bench$i.json" 2> "$O/bench$i.err"; rc=$?
      python tools/lease_summary.py "$O/bench$i.json" ;;
    quick)
      timeout 600 python bench.py --no-cpu-baseline --infer-steps 0 --no-fp32-route --no-extras --no-force-dp $arg > "$O/quick$i.json" 2> "$O/quick$i.err"; rc=$?
      python tools/lease_summary.py "$O/quick$i.json" ;;
    equick)          # equick:VAR=VAL[,VAR=VAL...][:bench args] — quick with environment switches (A/B sweeps)
      envs=${arg%%:*}; qargs=""; [ "$envs" != "$arg" ] && qargs=${arg#*:}
      (export ${envs//,/ }; timeout 600 python bench.py --no-cpu-baseline --infer-steps 0 --no-fp32-route --no-extras --no-force-dp $qargs > "$O/quick$i.json" 2> "$O/quick$i.err"); rc=$?
      echo "   [$envs]"; python tools/lease_summary.py "$O/quick$i.json" ;;
    tests)
      if [ -n "$arg" ]; then timeout 2400 python -m pytest tests -x -q -m gpu -k "$arg" --durations=8 > "$O/tests$i.log" 2>&1; else
        timeout 2400 python -m pytest tests -x -q -m gpu --durations=8 > "$O/tests$i.log" 2>&1; fi; rc=$?
      tail -15 "$O/tests$i.log" | grep -v Warning ;;
    file)
      timeout 2400 python -m pytest $arg -x -q -m gpu --durations=8 > "$O/tests$i.log" 2>&1; rc=$?
      tail -15 "$O/tests$i.log" | grep -v Warning ;;
    smoke)
      timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1; rc=$?; tail -2 "$O/smoke.log" ;;
    trace|trace1)
      extra=""; [ "$kind" = trace1 ] && extra="--no-wgrad-overlap"
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$kind" -o k -- python "$GRAFT_REPO_ROOT/bench.py" --steps 6 --warmup 3 --no-cpu-baseline \
        --infer-steps 0 --no-fp32-route --no-extras --no-force-dp --no-instrument --settle-seconds 0 $extra $arg > "$O/$kind.json" 2> "$O/$kind.err"); rc=$?
      cp "$(find "$O/$kind" -name '*kernel_stats.csv' | head -1)" "$O/${kind}_kernel_stats.csv" 2>> "$O/$kind.err"
      python tools/lease_summary.py "$O/$kind.json"; head -25 "$O/${kind}_kernel_stats.csv" | cut -c1-160
      find "$O/$kind" -name '*kernel_trace.csv' -size +40M -delete ;;
    pmc)
      (cd /tmp && timeout 900 rocprofv3 --pmc $arg --output-format csv -d "$O/pmc_${arg// /_}" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline \
        --infer-steps 0 --no-fp32-route --no-extras --no-force-dp --no-instrument --settle-seconds 0 > "$O/pmc$i.json" 2> "$O/pmc$i.err"); rc=$? ;;
    pmcfold)         # pmcfold:<tag> — the round's counter passes on ONE build: FETCH_SIZE, WRITE_SIZE (own passes, MI355X_MICROARCH.md) and three
                     # passes of <= 9 SQ counters, each with --kernel-trace only; folded on the box, the raw collections deleted
      B="--no-cpu-baseline --infer-steps 0 --no-force-dp --no-fp32-route --no-extras --no-instrument --settle-seconds 0 --steps 3 --warmup 1"
      P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
      P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS"
      P3="SQ_INSTS_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SALU SQ_WAVES SQ_INST_LEVEL_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_WR"
      rc=0; j=0
      for P in "FETCH_SIZE" "WRITE_SIZE" "$P1" "$P2" "$P3"; do
        j=$((j + 1))
        (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$O/pmcpass$j" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" $B > "$O/pmcpass$j.log" 2>&1) || rc=$?
        echo "   pass $j ($P): rc=$rc"
      done
      F=$(find "$O/pmcpass1" -name "*counter_collection.csv" | head -1); W=$(find "$O/pmcpass2" -name "*counter_collection.csv" | head -1)
      python tools/traffic_from_pmc.py "$F" "$W" "$O/${arg}_traffic.json" > "$O/traffic.log" 2>&1 || rc=$?
      python tools/sq_from_pmc.py "$O/${arg}_conv_pmc.json" "$O/${arg}_conv_pmc_table.md" $(find "$O/pmcpass3" "$O/pmcpass4" "$O/pmcpass5" -name "*counter_collection.csv" | sort) > "$O/sq.log" 2>&1 || rc=$?
      rm -rf "$O"/pmcpass[1-5]
      tail -4 "$O/traffic.log" | cut -c1-200 ;;
    host)
      timeout 600 python tools/hostprof.py --batches ${arg:-8 --lookahead} > "$O/host$i.txt" 2>&1; rc=$?; tail -12 "$O/host$i.txt" ;;
    sh)
      timeout 1800 bash -c "$arg" > "$O/sh$i.log" 2>&1; rc=$?; tail -40 "$O/sh$i.log" ;;
    py)
      timeout 1800 python $arg > "$O/py$i.log" 2>&1; rc=$?; tail -25 "$O/py$i.log" ;;
    *) echo "unknown job $job"; rc=64 ;;
  esac
  echo "[$TAG] job $i ($job): rc=$rc in ${SECONDS}s"
done
uptime >> "$O/uptime.txt"
