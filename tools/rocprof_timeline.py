#!/usr/bin/env python3
"""Timeline of the last N kernel dispatches of a rocprofv3 kernel trace (rocpd sqlite): start offset, duration, gap to the previous end.
Usage: tools/rocprof_timeline.py <results.db> [N=24] [SKIP=0]     (SKIP: leave out that many dispatches at the end first)
       TIMELINE_AFTER=<kernel-name-substring>:<i>  starts the window right after the i-th dispatch of that kernel instead (1-based) --
       e.g. inside bench.py's TIMED steps: its profiled steps later in the run carry an event record (a barrier packet, ~10 us of idle)
       at every kernel-family boundary, which the timed steps do not have."""
import os
import sqlite3
import sys


def main():
    db, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 24
    c = sqlite3.connect(db)
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    rows = list(c.execute("select name, start, end, grid_x / workgroup_x from kernels order by start"))
    after = os.environ.get("TIMELINE_AFTER")
    if after:
        sub, i = after.rsplit(":", 1)
        hits = [j for j, r in enumerate(rows) if sub in r[0]]
        at = hits[int(i) - 1] + 1
        rows = rows[at:at + n]
        print("# window: the %d dispatches after dispatch %s of a kernel named *%s* (of %d in the run)" % (n, i, sub, len(hits)))
    else:
        rows = rows[len(rows) - skip - n:len(rows) - skip] if skip else rows[-n:]
    t0, prev = rows[0][1], None
    print("%-60s %8s %10s %10s %8s" % ("kernel", "blocks", "start_us", "dur_us", "gap_us"))
    for name, s, e, g in rows:
        nm = name.replace("void ", "").replace("msvs::", "")[:60]
        print("%-60s %8d %10.1f %10.1f %8.1f" % (nm, g, (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
        prev = e


if __name__ == "__main__":
    main()
