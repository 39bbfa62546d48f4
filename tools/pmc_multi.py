#!/usr/bin/env python3
"""Per-kernel means of every counter in a rocprofv3 --pmc pass (rocpd sqlite output).
Usage: tools/pmc_multi.py <results.db> [kernel-substring]     (FETCH_SIZE is printed raw AND doubled: gfx950 counts 64 B
per 128-B request for wide coalesced streams, MI355X_MICROARCH.md section HBM)"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    rows = list(c.execute(
        "select %s, grid_size, counter_name, count(*), avg(value), max(value) from counters_collection "
        "group by %s, grid_size, counter_name order by %s, grid_size, counter_name" % (name_col, name_col, name_col)))
    print("# %s" % db)
    print("%-64s %10s %-28s %6s %16s %16s" % ("kernel", "grid", "counter", "calls", "avg", "max"))
    for nm, grid, cn, calls, avg, mx in rows:
        if pat and pat not in nm:
            continue
        nm = nm.replace("void ", "").replace("msvs::", "")
        nm = nm if len(nm) < 64 else nm[:60] + "..."
        print("%-64s %10s %-28s %6d %16.1f %16.1f" % (nm, grid, cn, calls, avg, mx))
        if cn == "FETCH_SIZE":
            print("%-64s %10s %-28s %6d %16.1f %16.1f" % (nm, grid, "FETCH_SIZE x2 (gfx950)", calls, 2 * avg, 2 * mx))


if __name__ == "__main__":
    main()
