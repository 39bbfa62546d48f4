#!/usr/bin/env python3
"""The dispatches around the slowest one of the kernels whose name contains <substring>.  Usage: tools/rocprof_around.py <results.db> <substring> [N=8]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
rows = list(c.execute("select name, start, end, grid_x / workgroup_x, grid_y / workgroup_y from kernels order by start"))
idx = max((i for i, r in enumerate(rows) if sys.argv[2] in r[0]), key=lambda i: rows[i][2] - rows[i][1])
t0 = rows[max(0, idx - n)][1]
for name, s, e, gx, gy in rows[max(0, idx - n):idx + n]:
    print("%-64s %6dx%-5d %10.1f %10.1f" % (name.replace("void ", "").replace("msvs::", "")[:64], gx, gy, (s - t0) / 1e3, (e - s) / 1e3))
