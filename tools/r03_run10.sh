#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03j; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_geometry.py -x -q -m gpu -k "coarse or full or deep1b or imi or wide or nearest" > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.txt
for V in 0 1 0 1; do
  if [ $V = 1 ]; then export VLQ_COARSE_PLAIN=1; else unset VLQ_COARSE_PLAIN; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-second-dataset --no-host-buffers --no-vlq > $OUT/bench_$V.json 2>/dev/null
  python - <<PY
import json
j=json.load(open("$OUT/bench_$V.json"))
print("plain=$V", "ms_per_step %.4f" % j["ms_per_step"], "stage", j["stage_ms"]["coarse"], j["stage_ms"]["scan"], "parity", j["parity"]["distance_bits_equal"], j["parity"]["label_mismatches"])
PY
done
