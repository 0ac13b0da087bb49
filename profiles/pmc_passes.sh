#!/bin/bash
# PMC counter passes for the vlq kernels of bench.py (one rocprofv3 run per pass:
# counters only, no tracing domains).  Usage (on the GPU box, from the repo root):
#   bash profiles/pmc_passes.sh <out_dir> [bench args]
set -e
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_LDS_ADDR_CONFLICT"
P3="FETCH_SIZE TCC_REQ_sum"
P4="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-include-regex "vlq::" --output-format csv -d "$OUT/pass$i" -- \
      python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-buffers --no-vlq --no-imi --no-deep1b "$@" > "$OUT/pass$i.json" 2> "$OUT/pass$i.err" || { tail -5 "$OUT/pass$i.err"; exit 1; }
done
