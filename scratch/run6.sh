mkdir -p gpurun_out
for p in 0 4 6 8 9 0 4; do
 timeout 120 tools/nbench --only N0 --mode fwd --variants 0 --reps 10 --no-check --prio $p 2>&1 | grep -E "^# wave|^N0|pipe " >> gpurun_out/nb_prio2.log
done
for p in 0 4 0 4; do
 timeout 200 tools/nbench --mode all --variants 0 --reps 10 --no-check --prio $p 2>&1 | grep -E "^# wave|^[LN][0-9]|variant 0 (plain|sorted|pairsL|pipeL|pipeS|glds ) |wgrad lds " >> gpurun_out/nb_prio3.log
done
