#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03i; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
(while true; do sleep 60; echo "[$(date +%T)] still running; $(free -g | awk 'NR==2{print "mem used " $3 " GB"}')"; done) &
HB=$!
VLQ_RUN_SIFT1B_DRIVER=1 timeout -k 10 1150 python -m pytest tests/test_reference_drivers.py -x -q -m gpu -s -k sift1b > $OUT/pytest_sift1b.txt 2>&1; echo "pytest rc=$?"
kill $HB
tail -30 $OUT/pytest_sift1b.txt
