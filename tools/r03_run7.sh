#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03g; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_fp16.py tests/test_cpp_shell.py -x -q -m gpu > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -25 $OUT/pytest.txt
