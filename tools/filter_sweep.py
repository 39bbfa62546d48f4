"""Filtered search: bit test inside the scan vs scan of a compacted view, over the selectivity of the filter.
Bench workload (1M x 768, nlist 1024, nprobe 32, k 10), stream-ordered device entry, per batch size.
    python tools/filter_sweep.py
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myscaledb_amd.capi as capi  # noqa: E402
from bench import make_data, make_queries  # noqa: E402

dev = torch.device("cuda", 0)
n, d, nlist, nprobe, k = 1_000_000, 768, 1024, 32, 10
model, x = make_data(n, d, 1234, dev)
ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, d, "ncentroids=1024,kmeans_iters=10,train_sample=65536")
ix.train(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
ix.add(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
ix.build()
stream = torch.cuda.current_stream().cuda_stream
price = np.random.default_rng(3).integers(0, 10000, n).astype(np.int32)
d_price = torch.from_numpy(price).to(dev)
t = time.perf_counter()
for _ in range(20):
    f = capi.Filter.from_predicate((np.int32, n), "<", 100, device_ptr=d_price.data_ptr())
    f.close()
print("predicate -> bitmap on the device (1M x int32 column, incl. population count + sync): %.1f us" % ((time.perf_counter() - t) / 20 * 1e6))
for B in (16, 256, 4096):
    q = make_queries(model, 4 * B, 4321, dev)
    oi = torch.empty((B, k), device=dev, dtype=torch.int64)
    od = torch.empty((B, k), device=dev, dtype=torch.float32)
    for frac in (0.001, 0.01, 0.05, 0.1, 0.2, 0.5, 0.9):
        flt = capi.Filter.from_predicate((np.int32, n), "<", int(frac * 10000), device_ptr=d_price.data_ptr())
        res = {}
        for name, below in (("bit test", "0"), ("compacted view", "1")):
            capi.set_option("filter_compact_below", below)
            for i in range(3):
                ix.search_filter_device(q[(i % 4) * B:(i % 4 + 1) * B].data_ptr(), B, k, nprobe, flt, oi.data_ptr(), od.data_ptr(), stream)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for i in range(12):
                ix.search_filter_device(q[(i % 4) * B:(i % 4 + 1) * B].data_ptr(), B, k, nprobe, flt, oi.data_ptr(), od.data_ptr(), stream)
            torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t) / 12
        capi.set_option("filter_compact_below", None)
        print("batch %5d  pass fraction %.3f : bit test %.3f ms/step   compacted view %.3f ms/step   ratio %.2f"
              % (B, frac, res["bit test"] * 1e3, res["compacted view"] * 1e3, res["bit test"] / res["compacted view"]), flush=True)
        flt.close()
