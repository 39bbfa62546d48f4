export TMPDIR=/tmp; R=$(pwd); cd /tmp
STEPS=12 rocprofv3 --kernel-trace -d /tmp/pp -o t -- python $R/tools/r5_bm25_ab.py --batches 1024 --variants 0 > /tmp/pp.log 2>&1
db=$(find /tmp/pp -name "*.db" | head -1)
python $R/tools/rocprof_durations.py $db "bm25l_kernel<0, 2"
python $R/tools/rocprof_durations.py $db "bm25l_kernel<1, 1"
python $R/tools/rocprof_around.py $db "bm25l_kernel<0, 2" 10
grep "^batch" /tmp/pp.log
