#!/bin/bash
# Kernel trace of any command (run on the GPU box from the repo root):  tools/prof_cmd.sh <out.txt> <command ...>   (TIMELINE=N: + the last N dispatches)
OUT=$1; shift; REPO=$(pwd); mkdir -p $(dirname $OUT); export TMPDIR=/tmp; D=/tmp/prof_$$; cd /tmp
rocprofv3 --kernel-trace --stats -d $D -o t -- "$@" > $REPO/$OUT.log 2>&1
db=$(find $D -name "*.db" | head -1)
[ -n "$db" ] && python $REPO/tools/rocprof_summary.py $db > $REPO/$OUT 2>&1
[ -n "$db" ] && [ -n "$TIMELINE" ] && python $REPO/tools/rocprof_timeline.py $db $TIMELINE ${TIMELINE_SKIP:-0} > $REPO/${OUT%.txt}_timeline.txt 2>&1
rm -rf $D; cd $REPO
