export TMPDIR=/tmp; R=$(pwd); cd /tmp
STEPS=8 timeout 120 rocprofv3 --kernel-trace -d /tmp/pp -o t -- python $R/tools/r5_bm25_ab.py --batches 1024 --variants 13 > /tmp/pp.log 2>&1
db=$(find /tmp/pp -name "*.db" | head -1)
python $R/tools/rocprof_timeline.py $db 18
grep "^batch" /tmp/pp.log
