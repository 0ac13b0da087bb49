cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2m
mkdir -p $OUT
M=32 NBITS=4 NPROBE=256 K=100 NLIST=65536 NEDGE=64 NB=30000000 CHECK=2 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $GRAFT_REPO_ROOT/tools/time_vlq.py 2000 5 > $OUT/log.txt 2>&1
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv | cut -c1-200
grep "^search" $OUT/log.txt
