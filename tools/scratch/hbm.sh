set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r03k; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --no-secondary"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$C" -o pmc -- $BENCH --steps 1 --warmup 0 --T 3 --no-profile > /dev/null 2> "$OUT/pmc_$C.err"
done
F=$(find "$OUT/pmc_FETCH_SIZE" -name "*counter_collection.csv" | head -1)
W=$(find "$OUT/pmc_WRITE_SIZE" -name "*counter_collection.csv" | head -1)
python "$REPO/tools/pmc_summary.py" "$F" "$W" 3 "$OUT/bench_pmc_hbm" > /dev/null 2> "$OUT/pmc_summary.err"
rm -rf "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE"
cd $REPO; timeout 200 python tools/op_profile.py 16 256 0 > $OUT/op_native.txt 2>&1
tail -80 $OUT/bench_pmc_hbm.txt; cat $OUT/pmc_summary.err | tail -5
