#!/bin/bash
# Round 6: kernel-trace A/B of several BUILDS of libmsvs.so on one box: average duration of the kernels matching $KERNELS per build
#   KERNELS="coarse_gemm|coarse_tail" tools/r6_lib_trace.sh d4 e1 e2
mkdir -p gpurun_out/ab
for t in "$@"; do
  cp ab_libs/libmsvs_$t.so myscaledb_amd/libmsvs.so
  tools/prof_cmd.sh gpurun_out/ab/trace_$t.txt python $PWD/bench.py --headline-only --steps 8 --warmup 2 --no-concurrent ${BENCH_ARGS}
  echo "== $t"; grep -E "${KERNELS:-coarse}" gpurun_out/ab/trace_$t.txt | cut -c1-200
done
