"""NQ (argv 1, default 32) queries per msvs_index_search call from HOST memory, back to back: the launch sequence and the per-call time of a small
host-pointer batch on the bench index (kernel trace: tools/prof_cmd.sh)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myscaledb_amd.capi as capi  # noqa: E402
import bench  # noqa: E402


def main():
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    n, d, k, nlist, nprobe = 1_000_000, 768, 10, 1024, 32
    dev = torch.device("cuda", 0)
    capi.set_device(0)
    x, q, _ = bench.data_model("blobs03", n, 4096, d, dev)
    ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, d, bench.ivf_params(nlist, n))
    ix.train(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
    ix.add(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
    ix.build()
    qh = q.cpu().numpy()
    for i in range(20):
        ix.search(qh[(i % 16) * nq:(i % 16 + 1) * nq], k, "nprobe=%d" % nprobe)
    t = time.perf_counter()
    for i in range(200):
        ix.search(qh[(i % 16) * nq:(i % 16 + 1) * nq], k, "nprobe=%d" % nprobe)
    print("%d queries per host call: %.1f us per call" % (nq, (time.perf_counter() - t) / 200 * 1e6))


if __name__ == "__main__":
    main()
