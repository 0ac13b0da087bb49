#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03e; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests/test_reference_drivers.py tests/test_gpu_dist.py tests/test_gpu_owned.py tests/test_gpu_vlq.py tests/test_index_io.py -x -q -m gpu -s "tests/test_gpu_parity.py::test_encode_preassigned_matches_encode" > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.txt
