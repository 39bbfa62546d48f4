#!/usr/bin/env python3
"""Every dispatch duration (us) of the kernels whose name contains <substring>, in time order.  Usage: tools/rocprof_durations.py <results.db> <substring>"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels order by start"))
out = [(n.replace("void ", "").replace("msvs::", "")[:40], (e - s) / 1e3) for n, s, e in rows if sys.argv[2] in n]
print(" ".join("%.1f" % d for _, d in out))
