#!/usr/bin/env python3
"""Timeline of the last N kernel dispatches of a rocprofv3 kernel trace (rocpd sqlite): start offset, duration, gap to the previous end.
Usage: tools/rocprof_timeline.py <results.db> [N=24] [SKIP=0]     (SKIP: leave out that many dispatches at the end first)"""
import sqlite3
import sys


def main():
    db, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 24
    c = sqlite3.connect(db)
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    rows = list(c.execute("select name, start, end, grid_x / workgroup_x from kernels order by start"))
    rows = rows[len(rows) - skip - n:len(rows) - skip] if skip else rows[-n:]
    t0, prev = rows[0][1], None
    print("%-60s %8s %10s %10s %8s" % ("kernel", "blocks", "start_us", "dur_us", "gap_us"))
    for name, s, e, g in rows:
        nm = name.replace("void ", "").replace("msvs::", "")[:60]
        print("%-60s %8d %10.1f %10.1f %8.1f" % (nm, g, (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
        prev = e


if __name__ == "__main__":
    main()
