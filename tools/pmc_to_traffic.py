#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/pmc_workload.py into profiles/traffic.json,
which bench.py reports as roofline.traffic.  FETCH_SIZE is in KiB and, on gfx950, counts 64 B per 128-B request for
wide coalesced streams: doubled (MI355X_MICROARCH.md, section HBM; calibrated in the same pass on gather_rows_kernel,
which reads exactly rows x dim x 4 B).  WRITE_SIZE (KiB) is taken as is.  The list scan of a step is two launches
(h16_sample_kernel, h16_scan_kernel): the per-step value is the sum of their per-launch means.
Usage: tools/pmc_to_traffic.py <fetch.db> <write.db> <batch> <rows> <dim> <round tag> [data model]"""
import json
import os
import sqlite3
import sys

KERNELS = ("h16_scan_kernel", "h16_sample_kernel")


def mean(db, counter, kern):
    c = sqlite3.connect(db)
    n, v = list(c.execute("select count(*), avg(value) from counters_collection where counter_name = ? and kernel_name like ?",
                          (counter, "%" + kern + "%")))[0]
    return n, v or 0.0


def main():
    fdb, wdb, batch, rows, dim, tag = sys.argv[1:7]
    data = sys.argv[7] if len(sys.argv) > 7 else "blobs03"
    per = {}
    f_total = w_total = 0.0
    for kname in KERNELS:
        nf, f = mean(fdb, "FETCH_SIZE", kname)
        nw, w = mean(wdb, "WRITE_SIZE", kname)
        per[kname] = {"launches_sampled": nf, "fetch_kib_raw": f, "fetch_bytes": int(f * 2 * 1024), "write_bytes": int(w * 1024)}
        f_total += f
        w_total += w
    c = sqlite3.connect(fdb)  # the largest gather_rows_kernel call is the 1M-row list layout pass
    cal = list(c.execute("select max(value) from counters_collection where counter_name = 'FETCH_SIZE' and "
                         "kernel_name like '%gather_rows_kernel%'"))[0][0] or 0.0
    out = {"round": tag, "data": data, "kernels": per, "batch": int(batch), "rows": int(rows), "dim": int(dim),
           "fetch_bytes_per_step": int(f_total * 2 * 1024), "write_bytes_per_step": int(w_total * 1024),
           "hbm_bytes_per_step": int(f_total * 2 * 1024 + w_total * 1024),
           "calibration": {"kernel": "gather_rows_kernel", "expected_bytes": int(rows) * int(dim) * 4,
                           "fetch_x2_bytes": int(cal * 2 * 1024)},
           "unit": "one search step = h16_sample_kernel launch + h16_scan_kernel launch (the list scan)",
           "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE --kernel-trace (separate passes), tools/pmc_workload.py 4 %s; "
                     "FETCH_SIZE x2 (gfx950), KiB -> bytes" % batch}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
