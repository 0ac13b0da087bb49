#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03c; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export SYNTH=1 NLIST=65536 NEDGE=64 NB=1000000000 CHECK=2 ROWS=2
PMC_TIMEOUT=400 bash profiles/pmc_cmd.sh $OUT/pmc_c5_rows2 "line16r" tools/time_vlq.py 2000 3 || echo "pmc failed"
grep -A22 "grid=512000" $OUT/pmc_c5_rows2/summary.txt
