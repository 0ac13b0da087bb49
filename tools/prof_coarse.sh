cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2k
mkdir -p $OUT
VLQ_COARSE_FILTER=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $GRAFT_REPO_ROOT/tools/slice_stages.py > $OUT/log.txt 2>&1
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
grep -i "coarse\|sample\|row_norms" $OUT/kernel_stats.csv | cut -c1-220
