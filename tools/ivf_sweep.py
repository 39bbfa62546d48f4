"""Tuning sweep of the IVF scan decomposition (MSVS_IVF_T / _RPB / _GRID / _XCD knobs), e.g.\n    python tools/ivf_sweep.py B=256 B=256,T=8,RPB=1024 B=1024\nResults of round 1: profiles/r01_ivf_tuning_sweep.txt."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import myscaledb_amd.capi as capi
from bench import make_data, make_queries
dev = torch.device('cuda', 0)
n, d, nlist, nprobe, k = 1_000_000, 768, 1024, 32, 10
model, x = make_data(n, d, 1234, dev)
ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, d, "ncentroids=1024,kmeans_iters=10,train_sample=65536")
ix.train(x.data_ptr(), n=n, mem=capi.MEM_DEVICE); ix.add(x.data_ptr(), n=n, mem=capi.MEM_DEVICE); ix.build()
stream = torch.cuda.current_stream().cuda_stream


def run(B, steps=20):
    q = make_queries(model, 8 * B, 4321, dev)
    oi = torch.empty((B, k), device=dev, dtype=torch.int64); od = torch.empty((B, k), device=dev, dtype=torch.float32)
    for i in range(3):
        ix.search_device(q[(i % 8) * B:(i % 8 + 1) * B].data_ptr(), B, k, nprobe, oi.data_ptr(), od.data_ptr(), stream)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(steps):
        ix.search_device(q[(i % 8) * B:(i % 8 + 1) * B].data_ptr(), B, k, nprobe, oi.data_ptr(), od.data_ptr(), stream)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / steps
    return dt, oi.cpu().numpy().copy()


ref = {}
for c in sys.argv[1:]:
    kv = dict(p.split('=') for p in c.split(','))
    B = int(kv.pop('B'))
    for kk in ("MSVS_IVF_T", "MSVS_IVF_RPB", "MSVS_IVF_GRID", "MSVS_IVF_XCD", "MSVS_IVF_WT", "MSVS_IVF_MFMA",
               "MSVS_IVF_EPS_SCALE", "MSVS_IVF_NQG"):
        os.environ.pop(kk, None)
    for a, b in kv.items():
        os.environ["MSVS_IVF_" + a] = b
    f0 = capi.prefilter_stats()
    dt, ids = run(B)
    f1 = capi.prefilter_stats()
    same = (ref.setdefault(B, ids) == ids).all()
    capi.profile_reset(); capi.profile_enable(True)
    run(B, steps=5)
    capi.profile_enable(False)
    fam = {}
    for name in ("flat_scan", "coarse_pass", "merge", "ivf_plan", "ivf_sample_scan", "ivf_scan", "rerank", "fallback_scan",
                 "fallback_merge"):
        c, ms = capi.profile_get(name)
        if c:
            fam[name] = round(ms / 8, 4)  # 3 warmup + 5 steps
    print("B=%d %s : %.3f ms/step  %.0f QPS  same_ids=%s  prefilter(q,fallback)=%s  kernels ms/step %s"
          % (B, kv, dt * 1e3, B / dt, same, (f1[0] - f0[0], f1[1] - f0[1]), fam), flush=True)
