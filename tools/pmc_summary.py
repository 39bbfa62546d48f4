#!/usr/bin/env python3
"""Per-kernel mean of one rocprofv3 --pmc counter from the rocpd sqlite output.
Usage: tools/pmc_summary.py <results.db> <COUNTER> [--double]   (--double: gfx950 FETCH_SIZE correction x2)"""
import sqlite3
import sys


def main():
    db, counter = sys.argv[1], sys.argv[2]
    factor = 2.0 if "--double" in sys.argv else 1.0
    c = sqlite3.connect(db)
    views = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    if "counters_collection" not in views:
        print("views/tables:", views)
        raise SystemExit("no counters_collection view in %s" % db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    print("# %s from %s (columns: %s)" % (counter, db, ",".join(cols)))
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    rows = list(c.execute(
        "select %s, grid_size, count(*), avg(value), min(value), max(value) from counters_collection "
        "where counter_name = ? group by %s, grid_size order by sum(value) desc" % (name_col, name_col), (counter,)))
    print("%-70s %12s %6s %16s %16s %16s" % ("kernel", "grid", "calls", "avg", "min", "max"))
    for r in rows[:25]:
        nm = r[0].replace("void ", "").replace("msvs::", "")
        nm = nm if len(nm) < 70 else nm[:66] + "..."
        print("%-70s %12s %6d %16.1f %16.1f %16.1f" % (nm, r[1], r[2], r[3] * factor, r[4] * factor, r[5] * factor))
    if factor != 1.0:
        print("# values multiplied by %.0f (gfx950 FETCH_SIZE counts 64 B per 128-B request)" % factor)


if __name__ == "__main__":
    main()
