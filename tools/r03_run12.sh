#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03l; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_cpp_shell.py -x -q -m gpu -s > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; grep -n "part 2h\|part 2g\|part 2e\|FAILED\|all ok\|passed\|failed" $OUT/pytest.txt | tail -12
