"""Round 5: concurrent single-query host callers (the bench's latency.threads_N leg alone), per option set given as k=v,k=v arguments."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myscaledb_amd.capi as capi  # noqa: E402
import myscaledb_amd.host as mhost  # noqa: E402
import bench  # noqa: E402


def times():
    import ctypes as C
    out = (C.c_uint64 * 4)()
    capi.lib().msvs_debug_combine_times(out)
    return tuple(out)


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
    sp = "nprobe=%d" % nprobe
    for opts in ({},) + tuple(dict(v.split("=") for v in a.split(",")) for a in sys.argv[1:]):
        for k_, v_ in opts.items():
            capi.set_option(k_, v_)
        for c in (8, 16, 32, 64, 128):
            per = 20000 // c
            mhost.concurrent_search(ix, qh, c, 50, k, sp)
            b0 = capi.combine_stats()
            t0 = times()
            sec, al, _, _ = mhost.concurrent_search(ix, qh, c, per, k, sp)
            b1 = capi.combine_stats()
            t1 = times()
            nb = int(b1[1] - b0[1])
            print("%-36s threads %3d: %8.0f QPS  p50 %6.1f us  p99 %7.1f us  batches %5d of %.1f queries" % (
                opts, c, c * per / sec, np.percentile(al, 50), np.percentile(al, 99), nb, (b1[2] - b0[2]) / max(nb, 1)),
                  " per batch: wall %.0f us = hand-over %.0f + gather %.0f + call %.0f + distribute %.0f + rest" % (
                      (sec * 1e6 / max(nb, 1),) + tuple((a - b) / 1e3 / max(nb, 1) for a, b in zip(t1, t0))), flush=True)
        for k_ in opts:
            capi.set_option(k_, None)


if __name__ == "__main__":
    main()
