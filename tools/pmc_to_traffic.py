#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/pmc_workload.py into profiles/traffic.json,
which bench.py reports as roofline.traffic.  FETCH_SIZE is in KiB and, on gfx950, counts 64 B per 128-B request for
wide coalesced streams: doubled (MI355X_MICROARCH.md, section HBM; calibrated in the same pass on gather_rows_kernel,
which reads exactly rows x dim x 4 B).  WRITE_SIZE (KiB) is taken as is.
Usage: tools/pmc_to_traffic.py <fetch.db> <write.db> <kernel substring> <batch> <rows> <dim> <round tag>"""
import json
import os
import sqlite3
import sys


def mean(db, counter, kern):
    """Per-step value of the list scan: the launches of `kern` with its LARGEST grid (the same kernel also runs the much
    smaller coarse-quantiser pass) come in pairs -- a short sample phase and the main phase (value >= half the largest);
    returns (main launches sampled, mean main + mean sample)."""
    c = sqlite3.connect(db)
    g = list(c.execute("select max(grid_size) from counters_collection where counter_name = ? and kernel_name like ?",
                       (counter, "%" + kern + "%")))[0][0]
    mx = list(c.execute("select max(value) from counters_collection where counter_name = ? and kernel_name like ? "
                        "and grid_size = ?", (counter, "%" + kern + "%", g)))[0][0] or 0.0
    q = ("select count(*), avg(value) from counters_collection where counter_name = ? and kernel_name like ? "
         "and grid_size = ? and value %s ?")
    hi = list(c.execute(q % ">=", (counter, "%" + kern + "%", g, 0.5 * mx)))[0]
    lo = list(c.execute(q % "<", (counter, "%" + kern + "%", g, 0.5 * mx)))[0]
    return hi[0], (hi[1] or 0.0) + (lo[1] or 0.0)


def main():
    fdb, wdb, kern, batch, rows, dim, tag = sys.argv[1:8]
    nf, f = mean(fdb, "FETCH_SIZE", kern)
    nw, w = mean(wdb, "WRITE_SIZE", kern)
    c = sqlite3.connect(fdb)  # the largest gather_rows_kernel call is the 1M-row list layout pass
    cal = list(c.execute("select max(value) from counters_collection where counter_name = 'FETCH_SIZE' and "
                         "kernel_name like '%gather_rows_kernel%'"))[0][0] or 0.0
    out = {"round": tag, "kernel": kern, "batch": int(batch), "rows": int(rows), "dim": int(dim),
           "launches_sampled": nf, "fetch_kib_raw": f, "fetch_bytes_per_launch": int(f * 2 * 1024),
           "write_bytes_per_launch": int(w * 1024), "hbm_bytes_per_launch": int(f * 2 * 1024 + w * 1024),
           "calibration": {"kernel": "gather_rows_kernel", "expected_bytes": int(rows) * int(dim) * 4,
                           "fetch_x2_bytes": int(cal * 2 * 1024)},
           "unit": "one search step = sample-phase launch + main-phase launch of the list scan",
           "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE --kernel-trace (separate passes), "
                     "tools/pmc_workload.py 4 %s; FETCH_SIZE x2 (gfx950), KiB -> bytes" % batch}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
