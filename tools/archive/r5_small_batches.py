"""Round 5: a small batch through the host-pointer entry (msvs_index_search) against the same batch through the device entry + a stream sync:
what the copies and the synchronisation cost (headline index: IVFFLAT nlist 1024, 1M x 768, blobs03, nprobe 32)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myscaledb_amd.capi as capi  # noqa: E402
import bench  # noqa: E402


def main():
    n, d, k, nlist, nprobe = 1_000_000, 768, 10, 1024, 32
    dev = torch.device("cuda", 0)
    capi.set_device(0)
    x, q, _ = bench.data_model("blobs03", n, 4096, d, dev)
    ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, d, bench.ivf_params(nlist, n))
    ix.train(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
    ix.add(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
    ix.build()
    qh = q.cpu().numpy()
    stream = torch.cuda.current_stream().cuda_stream
    oi = torch.empty((4096, k), device=dev, dtype=torch.int64)
    od = torch.empty((4096, k), device=dev, dtype=torch.float32)
    sp = "nprobe=%d" % nprobe
    # the one-launch coarse quantiser and plan of small batches against the multi-launch forms: same ids and distances
    for nq in (3, 5, 8, 17, 32, 64, 100, 128, 255):
        capi.set_option("coarse_few", "0")
        capi.set_option("plan_fused", "0")
        want = [ix.search(qh[i * nq:(i + 1) * nq], k, sp) for i in range(4)]
        capi.set_option("coarse_few", None)
        capi.set_option("plan_fused", None)
        got = [ix.search(qh[i * nq:(i + 1) * nq], k, sp) for i in range(4)]
        ok = all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(want, got))
        print("nq %3d: one-launch forms == multi-launch forms: %s" % (nq, ok), flush=True)
    for opts in ({},) + tuple(dict(v.split("=") for v in a.split(",")) for a in sys.argv[1:]):
        for k_, v_ in opts.items():
            capi.set_option(k_, v_)
        for nq in (1, 2, 4, 8, 16, 32, 64, 256):
            for i in range(10):
                ix.search(qh[i * nq:(i + 1) * nq], k, sp)
            t = time.perf_counter()
            for i in range(200):
                ix.search(qh[(i % 16) * nq:(i % 16 + 1) * nq], k, sp)
            host = (time.perf_counter() - t) / 200
            for i in range(10):
                ix.search_device(q[i * nq:(i + 1) * nq].data_ptr(), nq, k, nprobe, oi.data_ptr(), od.data_ptr(), stream)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for i in range(200):
                ix.search_device(q[(i % 16) * nq:(i % 16 + 1) * nq].data_ptr(), nq, k, nprobe, oi.data_ptr(), od.data_ptr(), stream)
                torch.cuda.synchronize()
            devs = (time.perf_counter() - t) / 200
            t = time.perf_counter()
            for i in range(200):
                ix.search_device(q[(i % 16) * nq:(i % 16 + 1) * nq].data_ptr(), nq, k, nprobe, oi.data_ptr(), od.data_ptr(), stream)
            torch.cuda.synchronize()
            devp = (time.perf_counter() - t) / 200
            print("%-24s nq %4d: host call %.1f us; device call + sync %.1f us; device calls back to back %.1f us per call" % (opts, nq, host * 1e6, devs * 1e6, devp * 1e6), flush=True)
        for k_ in opts:
            capi.set_option(k_, None)


if __name__ == "__main__":
    main()
