mkdir -p gpurun_out
rm -f gpurun_out/nb_sgb.log
for i in 1 2; do
 for b in tools/nbench scratch/sgb/nbench_sgb; do
  echo "## $b" >> gpurun_out/nb_sgb.log
  timeout 200 $b --mode fwd --variants 0 --reps 10 --no-check 2>&1 | grep -E "^[LN][0-9]|variant 0 (plain|sorted|pairsL|pipe |pipeS|pipeL) " >> gpurun_out/nb_sgb.log
 done
done
