#!/bin/bash
# PMC counters of scan16_kernel on long lists for one k (kernel experiments; counters only, no tracing
# domains):   bash tools/pmc_k.sh <out_dir under gpurun_out> <k>   on the GPU box, from the repo root
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export K=$2
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $P --kernel-include-regex "scan16_kernel" --output-format csv -d "$OUT/pass$i" -- python $GRAFT_REPO_ROOT/tools/long_lists.py 64000000 16384 10000 > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; exit 1; }
done
python $GRAFT_REPO_ROOT/profiles/summarize_pmc.py $OUT
