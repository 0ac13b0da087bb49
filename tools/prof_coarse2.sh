#!/bin/bash
# kernel trace + MFMA / wait counters of the f32 coarse kernels (screen off).  On the GPU box from the repo root:
#   bash tools/prof_coarse2.sh <out dir under gpurun_out>      (NQ / NLIST / DIM / NPROBE as tools/time_coarse.py)
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $GRAFT_REPO_ROOT/tools/time_coarse.py > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
find $OUT/trace -name "*kernel_stats.csv" | xargs -I{} cp {} $OUT/kernel_stats.csv
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_BUSY_CU_CYCLES"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $P --kernel-include-regex "coarse_dist" --output-format csv -d "$OUT/pass$i" -- python $GRAFT_REPO_ROOT/tools/time_coarse.py > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; exit 1; }
done
python $GRAFT_REPO_ROOT/profiles/summarize_pmc.py $OUT > $OUT/pmc.txt
