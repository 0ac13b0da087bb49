#!/bin/bash
# PMC counter passes (counters only: no tracing domains) for ANY of the timing tools, one rocprofv3 run per
# pass; workload parameters come from exported environment variables (never an `env` / `bash -c` hop
# between rocprofv3 and python).  On the GPU box, from the repo root:
#   NB=16000000 bash profiles/pmc_cmd.sh <out_dir> <kernel-include-regex> tools/time_vlq.py 2000 3
# -> <out_dir>/pass{1..4}/..., <out_dir>/summary.txt (profiles/summarize_pmc.py).
# FETCH_SIZE is in KB and counts 64 B per 128-B request on gfx950: bytes = FETCH_SIZE x 1024 x 2
# (MI355X_MICROARCH.md, HBM section).
set -e
OUT=$1; REGEX=$2; shift 2
mkdir -p "$OUT"
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_LDS_ADDR_CONFLICT"
P3="FETCH_SIZE TCC_REQ_sum"
P4="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  timeout -k 10 ${PMC_TIMEOUT:-300} rocprofv3 --pmc $P --kernel-include-regex "$REGEX" --output-format csv -d "$OUT/pass$i" -- \
      python "$REPO/$1" "${@:2}" > "$OUT/pass$i.log" 2>&1 || { tail -5 "$OUT/pass$i.log"; exit 1; }
done
python "$REPO/profiles/summarize_pmc.py" "$OUT" > "$OUT/summary.txt"
