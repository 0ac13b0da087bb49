#!/bin/bash
# VGPR / SGPR / LDS of every kernel of one source: tools/kernel_regs.sh scan16 [extra flags]
src=$1; shift
cd "$(dirname "$0")/../vector_line_quantization_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" --offload-device-only -S $src.hip -o /tmp/_regs.s 2>/dev/null
awk '/\.amdhsa_kernel /{n=$2} /amdhsa_next_free_vgpr/{v=$2} /amdhsa_next_free_sgpr/{s=$2} /\.end_amdhsa_kernel/{print v, s, n}' /tmp/_regs.s | while read v s n; do echo "$v $s $(echo $n | c++filt | sed "s/(vlq::ScanArgs, int)//")"; done
