#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03d; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_vlq.py -x -q -m gpu > $OUT/pytest_vlq.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_vlq.txt
export SYNTH=1 NLIST=65536 NEDGE=64 NB=1000000000 CHECK=4 ROWS=2
for W in 4 8 16; do VLQ_LINE16R_WAVES=$W timeout -k 10 300 python tools/time_vlq.py 2000 5 > $OUT/c5_w$W.log 2>&1; echo "waves $W: $(grep 'search:' $OUT/c5_w$W.log) $(grep -c VERIFIED $OUT/c5_w$W.log)"; done
export D=128
for W in 8 16; do VLQ_LINE16R_WAVES=$W timeout -k 10 300 python tools/time_vlq.py 2000 5 > $OUT/c5_d128_w$W.log 2>&1; echo "d128 waves $W: $(grep 'search:' $OUT/c5_d128_w$W.log) $(grep -c VERIFIED $OUT/c5_d128_w$W.log)"; done
ROWS=1 timeout -k 10 300 python tools/time_vlq.py 2000 5 > $OUT/c5_d128_rows1.log 2>&1; echo "d128 table rows: $(grep 'search:' $OUT/c5_d128_rows1.log)"
