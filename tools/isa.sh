#!/bin/bash
# Device ISA of one csrc translation unit:  tools/isa.sh bm25.hip [kernel-name-substring]  ->  /tmp/<unit>.s (+ /tmp/kernel.s)
cd "$(dirname "$0")/../myscaledb_amd/csrc" || exit 1
U=${1%.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -S --cuda-device-only -o /tmp/$U.s $U.hip 2>&1 | grep -v "warning\|^$" | head -5
if [ -n "$2" ]; then
  awk -v k="$2" 'index($0, k) && /^_Z[^ ]*:/ {p=1} p {print} p && /s_endpgm/ {exit}' /tmp/$U.s > /tmp/kernel.s
  wc -l /tmp/kernel.s
fi
