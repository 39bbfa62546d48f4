"""Single-query (batch 1) latency of msvs_index_search_device and its per-kernel split (HIP events)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myscaledb_amd.capi as capi
from bench import make_data, make_queries
dev = torch.device('cuda', 0)
n, d, nlist, nprobe, k = 1_000_000, 768, 1024, 32, 10
model, x = make_data(n, d, 1234, dev)
ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, d, "ncentroids=1024,kmeans_iters=10,train_sample=65536")
ix.train(x.data_ptr(), n=n, mem=capi.MEM_DEVICE); ix.add(x.data_ptr(), n=n, mem=capi.MEM_DEVICE); ix.build()
stream = torch.cuda.current_stream().cuda_stream
q = make_queries(model, 256, 777, dev)
oi = torch.empty((1, k), device=dev, dtype=torch.int64); od = torch.empty((1, k), device=dev, dtype=torch.float32)
for i in range(20):
    ix.search_device(q[i:i + 1].data_ptr(), 1, k, nprobe, oi.data_ptr(), od.data_ptr(), stream)
torch.cuda.synchronize()
lat = []
for i in range(200):
    t = time.perf_counter()
    ix.search_device(q[i % 256:i % 256 + 1].data_ptr(), 1, k, nprobe, oi.data_ptr(), od.data_ptr(), stream)
    torch.cuda.synchronize()
    lat.append(time.perf_counter() - t)
print("p50 %.1f us  p99 %.1f us" % (np.percentile(lat, 50) * 1e6, np.percentile(lat, 99) * 1e6))
# enqueue-only cost (host side of the 4 launches)
t = time.perf_counter()
for i in range(200):
    ix.search_device(q[i % 256:i % 256 + 1].data_ptr(), 1, k, nprobe, oi.data_ptr(), od.data_ptr(), stream)
host = (time.perf_counter() - t) / 200
torch.cuda.synchronize()
print("host enqueue %.1f us/query; back-to-back throughput %.1f us/query" % (host * 1e6, 0))
capi.profile_reset(); capi.profile_enable(True)
for i in range(100):
    ix.search_device(q[i % 256:i % 256 + 1].data_ptr(), 1, k, nprobe, oi.data_ptr(), od.data_ptr(), stream)
torch.cuda.synchronize(); capi.profile_enable(False)
for name in ("flat_scan", "merge", "ivf_plan", "ivf_scan"):
    c, ms = capi.profile_get(name)
    print("%-10s calls %4d  avg %.1f us" % (name, c, ms / max(c, 1) * 1e3))
