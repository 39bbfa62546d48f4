"""k-means quality on iid N(0,1)^768 rows (SURVEY 8d's first data model): list balance and recall@10 against the training
sample size and iteration count.   python tools/iid_train_sweep.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myscaledb_amd.capi as capi  # noqa: E402

dev = torch.device("cuda", 0)
capi.set_device(0)
n, d, nlist, k = 1_000_000, 768, 1024, 10
g = torch.Generator(device=dev).manual_seed(1234)
x = torch.randn((n, d), generator=g, device=dev, dtype=torch.float32)
g = torch.Generator(device=dev).manual_seed(4321)
q = torch.randn((1000, d), generator=g, device=dev, dtype=torch.float32).cpu().numpy()
fl = capi.Index(capi.INDEX_FLAT, capi.METRIC_L2, d)
fl.add(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
fl.build()
gt, _ = fl.search(q, k)
fl.close()
for sample, iters in ((65536, 10), (262144, 20), (1_000_000, 20), (1_000_000, 40)):
    t = time.time()
    ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, d, "ncentroids=%d,kmeans_iters=%d,train_sample=%d,seed=7" % (nlist, iters, sample))
    ix.train(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
    ix.add(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
    ix.build()
    torch.cuda.synchronize()
    bt = time.time() - t
    st = ix.list_stats()
    _, off, _, _ = ix.export(with_vecs=False)
    lens = np.diff(off)
    rec = {}
    for npb in (32, 128, 256):
        got, _ = ix.search(q, k, "nprobe=%d" % npb)
        rec[npb] = round(float(np.mean([len(set(a) & set(b)) / k for a, b in zip(got.tolist(), gt.tolist())])), 3)
        rows = ix.scanned_rows(q[:200], npb)[0] / 200
        rec["rows/q@%d" % npb] = int(rows)
    print("sample=%d iters=%d build %.1fs: min %d max %d imbalance %.2f lists<=10: %d, >4000: %d ; recall %s"
          % (sample, iters, bt, st["min_len"], st["max_len"], st["imbalance"], int((lens <= 10).sum()), int((lens > 4000).sum()), rec), flush=True)
    ix.close()
