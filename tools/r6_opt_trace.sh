#!/bin/bash
# Round 6: kernel-trace A/B of OPTION settings on one box:  KERNELS="coarse_gemm" tools/r6_opt_trace.sh "MSVS_X=1" "MSVS_X=2"   ("" = defaults)
mkdir -p gpurun_out/ab
i=0
for e in "$@"; do
  i=$((i+1))
  env $e tools/prof_cmd.sh gpurun_out/ab/opt_$i.txt python $PWD/bench.py --headline-only --steps 8 --warmup 2 --no-concurrent ${BENCH_ARGS}
  echo "== $e"; grep -E "${KERNELS:-coarse}" gpurun_out/ab/opt_$i.txt | cut -c1-200
  grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab/opt_$i.txt.log | head -1
done
