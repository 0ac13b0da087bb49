#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03f; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_vlq.py tests/test_gpu_geometry.py -x -q -m gpu -k "fp16 or vlq" > $OUT/pytest_vlq.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_vlq.txt
export SYNTH=1 NLIST=65536 NEDGE=64 NB=1000000000 CHECK=4 FP16=1
timeout -k 10 300 python tools/time_vlq.py 2000 5 > $OUT/c5_fp16.log 2>&1; grep "search:\|oracle" $OUT/c5_fp16.log
unset SYNTH NLIST NEDGE; export NB=16000000
timeout -k 10 300 python tools/time_vlq.py 2000 5 > $OUT/s16m_fp16.log 2>&1; grep "search:\|oracle" $OUT/s16m_fp16.log
