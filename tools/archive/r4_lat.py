"""Single-query latency with / without the radius pruning of the few-query path, on the three data models; parity against the
unpruned path on 1000 queries."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myscaledb_amd.capi as capi
from bench import data_model, ivf_params
dev = torch.device("cuda", 0)
n, d, nlist, nprobe, k = 1_000_000, 768, 1024, 32, 10
for kind in ("blobs03", "latent32", "iid"):
    x, q, _ = data_model(kind, n, 2048, d, dev)
    ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, d, ivf_params(nlist, n))
    ix.train(x.data_ptr(), n=n, mem=capi.MEM_DEVICE); ix.add(x.data_ptr(), n=n, mem=capi.MEM_DEVICE); ix.build()
    qh = q.cpu().numpy()
    res = {}
    for lp in ("0", "1"):
        capi.set_option("lat_prune", lp)
        for i in range(100):
            ix.search(qh[i:i + 1], k, "nprobe=%d" % nprobe)
        lat, out = [], []
        for i in range(2000):
            t = time.perf_counter()
            r = ix.search(qh[i % 2048:i % 2048 + 1], k, "nprobe=%d" % nprobe)
            lat.append(time.perf_counter() - t)
            if i < 1000:
                out.append(r)
        res[lp] = out
        two = ix.search(qh[:2], k, "nprobe=%d" % nprobe)
        print("%s lat_prune=%s: p50 %.1f us p99 %.1f us" % (kind, lp, np.percentile(lat, 50) * 1e6, np.percentile(lat, 99) * 1e6), flush=True)
    same = all((a[0] == b[0]).all() and (a[1].view(np.uint32) == b[1].view(np.uint32)).all() for a, b in zip(res["0"], res["1"]))
    print("   pruned == unpruned on 1000 queries (ids and distance bits):", same)
    capi.set_option("lat_prune", None)
    ix.close(); del x
