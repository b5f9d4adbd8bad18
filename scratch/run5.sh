mkdir -p gpurun_out
for p in 0 1 2 3 4 5 0; do
 timeout 120 tools/nbench --only N0 --mode fwd --variants 0 --reps 10 --no-check --prio $p 2>&1 | grep -E "^#|^N0|pipe |256x64 |plain " >> gpurun_out/nb_prio.log
done
