#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03b; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_vlq.py -x -q -m gpu > $OUT/pytest_vlq.txt 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_vlq.txt
timeout -k 10 600 python -m pytest tests/test_gpu_geometry.py -x -q -m gpu -k vlq > $OUT/pytest_geo.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_geo.txt
export SYNTH=1 NLIST=65536 NEDGE=64 NB=1000000000 CHECK=4
for R in 1 2; do ROWS=$R timeout -k 10 300 python tools/time_vlq.py 2000 5 > $OUT/c5_rows$R.log 2>&1; grep "search:\|oracle" $OUT/c5_rows$R.log; done
unset SYNTH NLIST NEDGE
export NB=16000000
for R in 1 2; do ROWS=$R timeout -k 10 300 python tools/time_vlq.py 2000 5 > $OUT/s16m_rows$R.log 2>&1; grep "search:\|oracle" $OUT/s16m_rows$R.log; done
export D=128
for R in 1 2; do ROWS=$R timeout -k 10 300 python tools/time_vlq.py 2000 5 > $OUT/s16m_d128_rows$R.log 2>&1; grep "search:\|oracle" $OUT/s16m_d128_rows$R.log; done
