set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r03l; mkdir -p $OUT
python tools/fused_nt_probe.py 10 > $OUT/nt_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/nt_fetch" -o pmc -- python $REPO/tools/fused_nt_probe.py 2 > /dev/null 2> "$OUT/nt_fetch.err"
python $REPO/tools/pmc_kernel.py "wino4_fused64" $(find "$OUT/nt_fetch" -name "*counter_collection.csv") > $OUT/nt_fetch_summary.txt 2>&1
rm -rf $OUT/nt_fetch
cd $REPO
cat $OUT/nt_probe.txt $OUT/nt_fetch_summary.txt
