"""Few-query latency of the IVFFLAT search on the bench workload (1M x 768, nlist 1024, nprobe 32, k 10):
host-pointer entry (msvs_index_search: query in, ids/distances out, includes everything between), lat_path on/off,
device entry + synchronise, and the per-call GPU time of the two-launch path (HIP events)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myscaledb_amd.capi as capi  # noqa: E402
from bench import make_data, make_queries  # noqa: E402

dev = torch.device("cuda", 0)
n, d, nlist, nprobe, k = int(os.environ.get("SWEEP_ROWS", 1_000_000)), 768, 1024, 32, 10
model, x = make_data(n, d, 1234, dev)
ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, d, "ncentroids=1024,kmeans_iters=10,train_sample=65536")
ix.train(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
ix.add(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
ix.build()
stream = torch.cuda.current_stream().cuda_stream
qd = make_queries(model, 1024, 777, dev)
qh = qd.cpu().numpy()
params = "nprobe=%d" % nprobe


def host_calls(nq, calls=2000):
    for i in range(100):
        ix.search(qh[i:i + nq], k, params)
    lat = []
    for i in range(calls):
        j = (i * nq) % (1024 - nq)
        t = time.perf_counter()
        ix.search(qh[j:j + nq], k, params)
        lat.append(time.perf_counter() - t)
    return np.percentile(lat, 50) * 1e6, np.percentile(lat, 99) * 1e6


for lp in ("1", "0"):
    capi.set_option("lat_path", lp)
    for nq in (1, 4):
        p50, p99 = host_calls(nq)
        print("lat_path=%s host-pointer search, %d query/call: p50 %.1f us  p99 %.1f us" % (lp, nq, p50, p99), flush=True)
    oi = torch.empty((1, k), device=dev, dtype=torch.int64)
    od = torch.empty((1, k), device=dev, dtype=torch.float32)
    for i in range(50):
        ix.search_device(qd[i:i + 1].data_ptr(), 1, k, nprobe, oi.data_ptr(), od.data_ptr(), stream)
    torch.cuda.synchronize()
    lat = []
    for i in range(500):
        t = time.perf_counter()
        ix.search_device(qd[i:i + 1].data_ptr(), 1, k, nprobe, oi.data_ptr(), od.data_ptr(), stream)
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t)
    print("lat_path=%s device entry + synchronize: p50 %.1f us  p99 %.1f us" % (lp, np.percentile(lat, 50) * 1e6, np.percentile(lat, 99) * 1e6))
    t = time.perf_counter()
    for i in range(500):
        ix.search_device(qd[i:i + 1].data_ptr(), 1, k, nprobe, oi.data_ptr(), od.data_ptr(), stream)
    torch.cuda.synchronize()
    print("lat_path=%s device entry back to back: %.1f us/query" % (lp, (time.perf_counter() - t) / 500 * 1e6), flush=True)
capi.set_option("lat_path", "1")
capi.profile_reset()
capi.profile_enable(True)
for i in range(200):
    ix.search_device(qd[i:i + 1].data_ptr(), 1, k, nprobe, oi.data_ptr(), od.data_ptr(), stream)
    torch.cuda.synchronize()
capi.profile_enable(False)
c, ms = capi.profile_get("lat_search")
print("two launches, HIP events around both: %.1f us (%d calls)" % (ms / max(c, 1) * 1e3, c))
