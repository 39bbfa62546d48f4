"""A/B of the list-scan kernels on BASELINE config 4's shard shape (d = 1536, inner product), one index, several batch sizes:
    python tools/c4_ab.py [rows] [nlist] [nprobe]
Prints ms per batch with the LDS-tile kernel (h16_reg=0) and the register-tile kernel (h16_reg=1 / 2), ids compared."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myscaledb_amd.capi as capi  # noqa: E402
from bench import make_data, make_queries  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
nlist = int(sys.argv[2]) if len(sys.argv) > 2 else max(256, rows // 1536)
nprobe = int(sys.argv[3]) if len(sys.argv) > 3 else 64
d = int(os.environ.get("AB_DIM", 1536))
metric = capi.METRIC_IP if os.environ.get("AB_METRIC", "ip") == "ip" else capi.METRIC_L2
dev = torch.device("cuda", 0)
capi.set_device(0)
model, x = make_data(rows, d, 1234, dev, blobs=nlist)
ix = capi.Index(capi.INDEX_IVFFLAT, metric, d, "ncentroids=%d,kmeans_iters=6,train_sample=%d" % (nlist, nlist * 48))
ix.train(x.data_ptr(), n=rows, mem=capi.MEM_DEVICE)
for lo in range(0, rows, 1_000_000):
    hi = min(rows, lo + 1_000_000)
    ix.add(x[lo:hi].data_ptr(), n=hi - lo, mem=capi.MEM_DEVICE)
ix.build()
del x
stream = torch.cuda.current_stream().cuda_stream
k = 10
for B in (4096, 1024, 256, 64):
    q = make_queries(model, B * 4, 4321, dev)
    oi = torch.empty((B, k), device=dev, dtype=torch.int64)
    od = torch.empty((B, k), device=dev, dtype=torch.float32)
    ref = None
    for reg in ("0", "1", "2"):
        capi.set_option("h16_reg", reg)
        it = [0]

        def run():
            b = it[0] % 4
            it[0] += 1
            ix.search_device(q[b * B:(b + 1) * B].data_ptr(), B, k, nprobe, oi.data_ptr(), od.data_ptr(), stream)
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        f0 = capi.prefilter_stats()
        t = time.perf_counter()
        n = 8
        for _ in range(n):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / n
        f1 = capi.prefilter_stats()
        it[0] = 0
        run()
        torch.cuda.synchronize()
        ids = oi.cpu().numpy().copy()
        same = True if ref is None else bool((ids == ref).all())
        ref = ids if ref is None else ref
        capi.profile_reset()
        capi.profile_enable(True)
        for _ in range(4):
            run()
        capi.profile_enable(False)
        cnt, ms = capi.profile_get("ivf_scan")
        print("rows=%d d=%d nlist=%d nprobe=%d B=%d h16_reg=%s : %.3f ms/batch %.0f QPS  list scan %.3f ms  same_ids=%s fallbacks=%d/%d"
              % (rows, d, nlist, nprobe, B, reg, dt * 1e3, B / dt, ms / max(cnt, 1), same, f1[1] - f0[1], f1[0] - f0[0]), flush=True)
    capi.set_option("h16_reg", None)
