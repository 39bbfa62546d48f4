V=0 bash tools/r5_pmc_bm25.sh 2>&1 | grep "bm25l_kernel<1" | awk '{print $(NF-3), $(NF-1)}' > gpurun_out/pmc_v0.txt
V=12 bash tools/r5_pmc_bm25.sh 2>&1 | grep "bm25l_kernel<1" | awk '{print $(NF-3), $(NF-1)}' > gpurun_out/pmc_v12.txt
paste gpurun_out/pmc_v0.txt gpurun_out/pmc_v12.txt
