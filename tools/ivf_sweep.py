"""Tuning sweep of the IVF search on the bench workload (1M x 768, nlist 1024, nprobe 32) through msvs_set_option knobs:

    python tools/ivf_sweep.py B=4096 B=4096,ivf_h16=0 B=256,h16_ncb=2 B=1024,h16_grid=2048

Every argument is one configuration: B = queries per step, the other key=value pairs are option names.
Prints ms/step, QPS, whether the ids equal the first configuration's at that B, fallbacks and kernel-family times.
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myscaledb_amd.capi as capi  # noqa: E402
from bench import make_data, make_queries  # noqa: E402

dev = torch.device("cuda", 0)
n, d, nlist, nprobe, k = int(os.environ.get("SWEEP_ROWS", 1_000_000)), 768, 1024, int(os.environ.get("SWEEP_NPROBE", 32)), int(os.environ.get("SWEEP_K", 10))
BLOBS = os.environ.get("SWEEP_DATA") == "blobs03"  # SURVEY 8d's clustered model: 1024 gaussian blobs, sigma 0.3, in R^768
IID = os.environ.get("SWEEP_IID") == "1"  # rows and queries iid N(0,1)^768 instead of the bench mixture
model, x = make_data(n, d, 1234, dev)
if IID:
    g = torch.Generator(device=dev).manual_seed(1234)
    x = torch.randn((n, d), device=dev, dtype=torch.float32, generator=g)
if BLOBS:
    centres = torch.randn((1024, d), device=dev, dtype=torch.float32, generator=torch.Generator(device=dev).manual_seed(99))
    g = torch.Generator(device=dev).manual_seed(1234)
    for lo in range(0, n, 131072):
        hi = min(n, lo + 131072)
        x[lo:hi] = centres[torch.randint(0, 1024, (hi - lo,), generator=g, device=dev)] + 0.3 * torch.randn((hi - lo, d), generator=g, device=dev, dtype=torch.float32)
ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, d, "ncentroids=%d,kmeans_iters=10,train_sample=65536" % nlist)
ix.train(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
ix.add(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
ix.build()
stream = torch.cuda.current_stream().cuda_stream
FAMILIES = ("flat_scan", "coarse_pass", "merge", "ivf_plan", "ivf_prep", "ivf_sample_scan", "ivf_scan", "rerank",
            "fallback_scan", "fallback_merge")


def run(B, steps=20):
    q = make_queries(model, 8 * B, 4321, dev)
    if IID:
        q = torch.randn((8 * B, d), device=dev, dtype=torch.float32, generator=torch.Generator(device=dev).manual_seed(4321))
    if BLOBS:
        gq = torch.Generator(device=dev).manual_seed(4321)
        q = centres[torch.randint(0, 1024, (8 * B,), generator=gq, device=dev)] + 0.3 * torch.randn((8 * B, d), generator=gq, device=dev, dtype=torch.float32)
    oi = torch.empty((B, k), device=dev, dtype=torch.int64)
    od = torch.empty((B, k), device=dev, dtype=torch.float32)
    for i in range(3):
        ix.search_device(q[(i % 8) * B:(i % 8 + 1) * B].data_ptr(), B, k, nprobe, oi.data_ptr(), od.data_ptr(), stream)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(steps):
        ix.search_device(q[(i % 8) * B:(i % 8 + 1) * B].data_ptr(), B, k, nprobe, oi.data_ptr(), od.data_ptr(), stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    return dt, oi.cpu().numpy().copy()


def dump_stamps():
    """Per-item wall-clock stamps of the last shadow main launch (option h16_stamps): where a workgroup's time goes."""
    import ctypes as C
    import numpy as np
    buf = np.zeros(4096 * 64 * 4, np.uint64)
    grid = C.c_uint32(0)
    fn = capi.lib().msvs_debug_h16_stamps
    fn.argtypes = [C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_uint32)]
    capi._check(fn(buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size, C.byref(grid)))
    g = grid.value
    st = buf[:g * 256].reshape(g, 64, 4).astype(np.int64)
    nit = np.minimum(st[:, 0, 0], 63)
    t0s, tile, strm, nval, lists, ends, firsts = [], [], [], [], [], [], []
    for b in range(g):
        for i in range(1, int(nit[b]) + 1):
            t0, t1, t2, meta = st[b, i]
            t0s.append(t0); tile.append(t1 - t0); strm.append(t2 - t1); nval.append((meta >> 8) & 0xFFFFFF); lists.append(meta >> 32)
        if nit[b]:
            firsts.append(st[b, 1, 0]); ends.append(st[b, int(nit[b]), 2])
    t0s, tile, strm, nval = map(np.asarray, (t0s, tile, strm, nval))
    start, end = min(firsts), max(ends)
    span = (end - start) / 100.0  # us (100 MHz)
    busy = (tile.sum() + strm.sum()) / 100.0 / g
    print("   stamps: grid %d, items %d (per wg min/mean/max %d/%.1f/%d), launch span %.1f us; per-wg mean: tile load %.1f us, rows %.1f us, "
          "other+idle %.1f us; wg finish spread: p10 %.1f p50 %.1f p90 %.1f max %.1f us after start"
          % (g, len(tile), nit.min(), nit.mean(), nit.max(), span, tile.sum() / 100.0 / g, strm.sum() / 100.0 / g, span - busy,
             *[(np.percentile(ends, q) - start) / 100.0 for q in (10, 50, 90)], span))
    print("   per item: tile load us p10/p50/p90 %.1f/%.1f/%.1f; rows us p10/p50/p90 %.1f/%.1f/%.1f; queries per item p10/p50/p90/max %d/%d/%d/%d; "
          "items with <=32 / <=48 / <=64 queries: %.2f / %.2f / %.2f"
          % (*[np.percentile(tile, q) / 100.0 for q in (10, 50, 90)], *[np.percentile(strm, q) / 100.0 for q in (10, 50, 90)],
             *[np.percentile(nval, q) for q in (10, 50, 90)], nval.max(), (nval <= 32).mean(), (nval <= 48).mean(), (nval <= 64).mean()))
    first_wave = (np.asarray(firsts) - start) / 100.0
    print("   first item popped us after start: p50 %.1f max %.1f" % (np.percentile(first_wave, 50), first_wave.max()))


ref = {}
for c in sys.argv[1:]:
    kv = dict(p.split("=") for p in c.split(","))
    B = int(kv.pop("B"))
    for a, b in kv.items():
        capi.set_option(a, b)
    f0, c0 = capi.prefilter_stats(), capi.coarse_stats()
    dt, ids = run(B)
    f1, c1 = capi.prefilter_stats(), capi.coarse_stats()
    same = bool((ref.setdefault(B, ids) == ids).all())
    capi.profile_reset()
    capi.profile_enable(True)
    run(B, steps=5)
    capi.profile_enable(False)
    fam = {}
    for name in FAMILIES:
        cnt, ms = capi.profile_get(name)
        if cnt:
            fam[name] = round(ms / 8, 4)  # 3 warm-up + 5 steps
    capi.profile_reset()
    if kv.get("h16_stamps") == "1":
        dump_stamps()
    for a in kv:
        capi.set_option(a, None)
    if kv.get("rerank_stats") == "1":
        import ctypes as C
        import numpy as np
        st = np.zeros(6, np.uint64)
        capi.lib().msvs_debug_rerank_stats(st.ctypes.data_as(C.POINTER(C.c_uint64)))
        print("   re-rank candidates beyond e_k +- eps (cumulative): lists %d of %d, coarse %d of %d; rows skipped: lists %d, coarse %d"
              % tuple(int(x) for x in st))
    print("B=%d %s : %.3f ms/step  %.0f QPS  same_ids=%s  prefilter(q,fallback)=%s coarse(q,fallback)=%s  kernels ms/step %s"
          % (B, kv, dt * 1e3, B / dt, same, (f1[0] - f0[0], f1[1] - f0[1]), (c1[0] - c0[0], c1[1] - c0[1]), fam), flush=True)
