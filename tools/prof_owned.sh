set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2d
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
export VLQ_SCAN_SCHEDULE=2
A="--no-cpu-baseline --no-second-dataset --no-host-buffers --sigma 0.005 --rank 12 --spread 0.4 --steps 5 --warmup 2"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $REPO/bench.py $A > $OUT/bench_trace.json 2> $OUT/trace.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE TCC_REQ_sum --kernel-include-regex "vlq::" --output-format csv -d $OUT/p3 -- python $REPO/bench.py $A > $OUT/p3.json 2> $OUT/p3.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-include-regex "vlq::" --output-format csv -d $OUT/p4 -- python $REPO/bench.py $A > $OUT/p4.json 2> $OUT/p4.err
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
echo done
