#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03k; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for V in 0 1; do
  if [ $V = 1 ]; then export VLQ_COARSE_PLAIN=1; else unset VLQ_COARSE_PLAIN; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace$V -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-second-dataset --no-host-buffers --no-vlq > $OUT/b$V.json 2> $OUT/t$V.err
  f=$(find $OUT/trace$V -name "*kernel_stats.csv" | head -1)
  echo "== plain=$V"; grep -E "coarse_dist|coarse_select|scan16_kernel" $f | cut -d, -f1-4 | cut -c1-160
done
