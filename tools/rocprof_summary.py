#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace: per (kernel, grid) call count, total and
average duration.  Usage: tools/rocprof_summary.py <results.db> [> profiles/rNN_xxx.txt]
Collected with:  cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d <dir> -o bench -- python bench.py ..."""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = list(c.execute(
        "select name, grid_x/workgroup_x, grid_y/workgroup_y, grid_z/workgroup_z, vgpr_count, accum_vgpr_count, "
        "lds_size, scratch_size, count(*), sum(duration), avg(duration), min(duration), max(duration) "
        "from kernels group by name, grid_x, grid_y, grid_z order by sum(duration) desc"))
    total = sum(r[9] for r in rows) or 1
    print("# rocprofv3 --kernel-trace summary of %s (durations in microseconds)" % path)
    print("%-58s %-16s %5s %5s %7s %7s %6s %12s %10s %10s %10s %6s" % (
        "kernel", "grid(blocks)", "vgpr", "agpr", "lds", "scratch", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for r in rows[:40]:
        name = r[0].replace("void ", "").replace("msvs::", "")
        if len(name) > 57:
            name = name[:54] + "..."
        print("%-58s %-16s %5d %5d %7d %7d %6d %12.1f %10.2f %10.2f %10.2f %6.2f" % (
            name, "%dx%dx%d" % (r[1], r[2], r[3]), r[4], r[5], r[6], r[7], r[8], r[9] / 1e3, r[10] / 1e3, r[11] / 1e3,
            r[12] / 1e3, 100.0 * r[9] / total))


if __name__ == "__main__":
    main(sys.argv[1])
