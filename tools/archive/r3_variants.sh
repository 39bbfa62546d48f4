#!/bin/bash
# build variants of the shadow scan ON the GPU box (same image: hipcc is there) and time the headline step with each
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
H=myscaledb_amd/csrc/h16_scan_kernels.hpp
cp $H /tmp/h16_orig.hpp
run() { # label
  make -C myscaledb_amd/csrc -j8 > /tmp/make.log 2>&1 || { echo "$1: build failed"; tail -5 /tmp/make.log; return; }
  timeout 300 python bench.py --headline-only --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads([x for x in sys.stdin if x.startswith('{')][-1])
print('$1', d['value'], d['ms_per_step'], 'scan', d['roofline']['step_kernels_ms']['ivf_scan'], d['roofline']['prefilter'])
"
}
{
run "base(NW8,RING4)"
sed -i 's/^constexpr int H_NW = 8; /constexpr int H_NW = 16; /' $H; run "NW16,RING4"
sed -i 's/^constexpr int H_RING = 4; /constexpr int H_RING = 6; /' $H; run "NW16,RING6"
cp /tmp/h16_orig.hpp $H; sed -i 's/^constexpr int H_RING = 4; /constexpr int H_RING = 6; /' $H; run "NW8,RING6"
} 2>&1 | tee gpurun_out/r3_variants.txt
cp /tmp/h16_orig.hpp $H
