#!/bin/bash
# kernel trace of the headline steps -> gpurun_out/r4/trace_<tag>.txt (median / min per kernel of the timed steps)
TAG=${1:-now}; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/r4; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
env "$@" timeout 600 rocprofv3 --kernel-trace -d /tmp/tr_$TAG -o t -- python $REPO/bench.py --headline-only --steps 30 --warmup 5 > $OUT/trace_${TAG}_bench.json 2> $OUT/trace_${TAG}.log
db=$(find /tmp/tr_$TAG -name "*.db" | head -1)
python - "$db" > $OUT/trace_$TAG.txt <<'PY'
import sqlite3, sys, statistics
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
# the last 30 steps: find the last 30 h16_scan_kernel dispatches and take everything from the prep kernel before the 30th-last one
idx = [i for i, r in enumerate(rows) if r[0].startswith('h16_scan_kernel') or 'h16_scan_kernel<' in r[0]]
if len(idx) >= 31:
    start = idx[-31] + 1
    rows = rows[start:]
nsteps = 30
agg = {}
for name, s, e in rows:
    short = name.split('(')[0][:60]
    agg.setdefault(short, []).append((e - s) / 1000.0)
tot = 0
print("%-62s %6s %9s %9s %9s" % ("kernel (last 30 steps)", "calls", "median_us", "min_us", "us/step"))
for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    per = sum(v) / nsteps
    tot += per
    print("%-62s %6d %9.2f %9.2f %9.2f" % (name, len(v), statistics.median(v), min(v), per))
span = (rows[-1][2] - rows[0][1]) / 1000.0 / nsteps
print("sum of kernels per step %.1f us; span per step %.1f us" % (tot, span))
PY
rm -rf /tmp/tr_$TAG
cat $OUT/trace_$TAG.txt | head -40
