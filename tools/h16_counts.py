"""Candidates per query that the fp16-shadow list scan appended (below the sample cut), bench mixture vs iid data:

    python tools/h16_counts.py            # prints count percentiles for both data models at 4096 queries per step
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myscaledb_amd.capi as capi  # noqa: E402
from bench import make_data, make_queries  # noqa: E402

dev = torch.device("cuda", 0)
n, d, nlist, nprobe, k, B = 1_000_000, 768, 1024, 32, 10, 4096
stream = torch.cuda.current_stream().cuda_stream
for name in ("mixture", "iid"):
    model, x = make_data(n, d, 1234, dev)
    q = make_queries(model, B, 4321, dev)
    if name == "iid":
        x = torch.randn((n, d), device=dev, dtype=torch.float32, generator=torch.Generator(device=dev).manual_seed(1234))
        q = torch.randn((B, d), device=dev, dtype=torch.float32, generator=torch.Generator(device=dev).manual_seed(4321))
    ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, d, "ncentroids=%d,kmeans_iters=10,train_sample=65536" % nlist)
    ix.train(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
    ix.add(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
    ix.build()
    oi = torch.empty((B, k), device=dev, dtype=torch.int64)
    od = torch.empty((B, k), device=dev, dtype=torch.float32)
    ix.search_device(q.data_ptr(), B, k, nprobe, oi.data_ptr(), od.data_ptr(), stream)
    torch.cuda.synchronize()
    keys = np.zeros((B, 1), np.uint64)
    cnt = np.zeros(B, np.uint32)
    rc = capi.lib().msvs_debug_h16_keys(keys.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_size_t(1),
                                        cnt.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_size_t(B))
    assert rc == 0, capi.last_error()
    off = ix.export(with_vecs=False)[1]
    print(name, "candidates per query: mean %.0f  p10 %d  p50 %d  p90 %d  p99 %d  max %d  sum %d" % (
        cnt.mean(), *np.percentile(cnt, [10, 50, 90, 99]).astype(int), cnt.max(), cnt.sum()), flush=True)
    if off is not None:
        ln = np.diff(off)
        print(name, "list lengths: min %d  p50 %d  p90 %d  max %d" % (ln.min(), np.median(ln), np.percentile(ln, 90), ln.max()), flush=True)
    del ix, x
