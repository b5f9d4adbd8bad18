#!/bin/bash
# diagnostic (r6): a quick bench line while N busy-loop processes load the host's CPUs (the slow first-process runs of r6_notes.md section 13
# happened on hosts at load averages of 46-57) — does the step become host-bound under plain CPU contention?
#   tools/hostload.sh <hogs> <seconds> [bench args]
N=${1:-128}; T=${2:-90}; shift 2
for i in $(seq $N); do (timeout $T python3 -c "while True: pass" > /dev/null 2>&1 &) ; done
sleep 3
uptime
python bench.py --no-cpu-baseline --infer-steps 0 --no-fp32-route --no-extras --no-force-dp "$@"
