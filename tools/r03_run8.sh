#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03h; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $OUT/pytest_all.txt 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_all.txt
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 6000 $OUT/bench.json
