import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myscaledb_amd.capi as capi
from bench import data_model
dev = torch.device("cuda", 0)
n, d, nlist, nprobe, k, B = 1_000_000, 768, 1024, 32, 10, 1024
x, q, _ = data_model("blobs03", n, 8 * B, d, dev)
ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, d, "ncentroids=%d,kmeans_iters=10,train_sample=65536" % nlist)
ix.train(x.data_ptr(), n=n, mem=capi.MEM_DEVICE); ix.add(x.data_ptr(), n=n, mem=capi.MEM_DEVICE); ix.build()
oi = torch.empty((B, k), device=dev, dtype=torch.int64); od = torch.empty((B, k), device=dev, dtype=torch.float32)
for pre in ("0", "1"):
    capi.set_option("h16_preprune", pre)
    for b in range(3):
        f0 = capi.prefilter_stats()
        ix.search_device(q[b * B:(b + 1) * B].data_ptr(), B, k, nprobe, oi.data_ptr(), od.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        f1 = capi.prefilter_stats()
        keys = np.zeros((B, 4096), np.uint64); cnt = np.zeros(B, np.uint32)
        rc = capi.lib().msvs_debug_h16_keys(keys.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_size_t(4096), cnt.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_size_t(B))
        print("preprune", pre, "batch", b, "fallbacks", f1[1] - f0[1], "rc", rc, "cand count min/p50/p90/max", cnt.min(), int(np.median(cnt)), int(np.percentile(cnt, 90)), cnt.max(),
              "queries with < 40 candidates:", int((cnt < 40).sum()), "with < 11:", int((cnt < 11).sum()))
