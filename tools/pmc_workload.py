#!/usr/bin/env python3
"""Minimal workload for rocprofv3 counter passes (same index / data model / batch as bench.py, few steps):
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d <out> -o fetch -- python tools/pmc_workload.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d <out> -o write -- python tools/pmc_workload.py
then tools/pmc_summary.py <out>/fetch_results.db FETCH_SIZE  (gfx950: FETCH_SIZE counts 64 B per 128-B request for
wide coalesced streams -> the summary doubles it, MI355X_MICROARCH.md section HBM)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import myscaledb_amd.capi as capi  # noqa: E402
from bench import data_model, ivf_params  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    dev = torch.device("cuda", 0)
    n, d, nlist, nprobe, k = 1_000_000, 768, 1024, 32, 10
    x, q, _ = data_model(os.environ.get("PMC_DATA", "blobs03"), n, 16 * B, d, dev)  # the headline's data model
    ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, d, ivf_params(nlist, n))
    ix.train(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
    ix.add(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
    ix.build()
    oi = torch.empty((B, k), device=dev, dtype=torch.int64)
    od = torch.empty((B, k), device=dev, dtype=torch.float32)
    stream = torch.cuda.current_stream().cuda_stream
    rows = []
    for i in range(steps):
        qb = q[(i % 16) * B:(i % 16 + 1) * B]
        ix.search_device(qb.data_ptr(), B, k, nprobe, oi.data_ptr(), od.data_ptr(), stream)
        rows.append(ix.scanned_rows(qb.cpu().numpy(), nprobe))
    torch.cuda.synchronize()
    print("steps", steps, "batch", B, "rows(model, streamed) per step:", rows, "row bytes", 4 * d + 4)


if __name__ == "__main__":
    main()
