cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/traf
mkdir -p $O
cd $R
for tag in plain sorted; do
  extra=""; [ $tag = sorted ] && extra="--spatial-sort"
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c -d $O/${tag}_$c -o p --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-instrument --infer-steps 0 $extra > $O/${tag}_$c.log 2>&1)
  done
  python tools/traffic_from_pmc.py $(find $O/${tag}_FETCH_SIZE -name "*counter_collection.csv") $(find $O/${tag}_WRITE_SIZE -name "*counter_collection.csv") $O/traffic_$tag.json > /dev/null
done
for i in 1 2; do
timeout 300 python bench.py --no-instrument --steps 30 --warmup 10 --no-cpu-baseline --infer-steps 0 2>&1 | tail -1 > $O/b_plain_$i.json
timeout 300 python bench.py --no-instrument --steps 30 --warmup 10 --no-cpu-baseline --infer-steps 0 --spatial-sort 2>&1 | tail -1 > $O/b_sorted_$i.json
done
# keep only small files
find $O -name "*.csv" -size +20M -delete
