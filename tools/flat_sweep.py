"""FLAT index batches (exhaustive scan): 1M x 768 iid rows, queries per step B, option sets as in ivf_sweep.py:

    python tools/flat_sweep.py B=4096 B=4096,flat_h16=0 B=64 B=64,flat_h16=0

Prints ms/step, QPS, the MFMA / HBM rates of the shadow scan, whether the ids equal the first configuration's at that B."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myscaledb_amd.capi as capi  # noqa: E402

dev = torch.device("cuda", 0)
n, d, k = int(os.environ.get("SWEEP_ROWS", 1_000_000)), int(os.environ.get("SWEEP_DIM", 768)), int(os.environ.get("SWEEP_K", 10))
x = torch.randn((n, d), device=dev, dtype=torch.float32, generator=torch.Generator(device=dev).manual_seed(1234))
ix = capi.Index(capi.INDEX_FLAT, capi.METRIC_L2, d)
ix.add(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
ix.build()
stream = torch.cuda.current_stream().cuda_stream
FAMILIES = ("flat_pass", "flat_shadow_scan", "table_scan", "flat_scan", "merge", "rerank", "fallback_scan", "fallback_merge")
qpool = torch.randn((8192, d), device=dev, dtype=torch.float32, generator=torch.Generator(device=dev).manual_seed(4321))


def run(B, steps):
    oi = torch.empty((B, k), device=dev, dtype=torch.int64)
    od = torch.empty((B, k), device=dev, dtype=torch.float32)
    nb = max(1, 8192 // B)
    for i in range(2):
        ix.search_device(qpool[(i % nb) * B:(i % nb + 1) * B].data_ptr(), B, k, 0, oi.data_ptr(), od.data_ptr(), stream)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(steps):
        ix.search_device(qpool[(i % nb) * B:(i % nb + 1) * B].data_ptr(), B, k, 0, oi.data_ptr(), od.data_ptr(), stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps, oi.cpu().numpy().copy()


ref = {}
for c in sys.argv[1:]:
    kv = dict(p.split("=") for p in c.split(","))
    B = int(kv.pop("B"))
    for a, b in kv.items():
        capi.set_option(a, b)
    steps = 5 if B >= 1024 else 20
    f0 = capi.prefilter_stats()
    dt, ids = run(B, steps)
    f1 = capi.prefilter_stats()
    same = bool((ref.setdefault(B, ids) == ids).all())
    capi.profile_reset()
    capi.profile_enable(True)
    run(B, 3)
    capi.profile_enable(False)
    fam = {}
    for name in FAMILIES:
        cnt, ms = capi.profile_get(name)
        if cnt:
            fam[name] = round(ms / 5, 4)  # 2 warm-up + 3 steps
    capi.profile_reset()
    for a in kv:
        capi.set_option(a, None)
    sc = fam.get("flat_shadow_scan", 0)
    dpad = (d + 63) // 64 * 64
    extra = ""
    if sc:
        extra = " shadow scan: %.0f TF/s (%.3f of 2500), %.0f GB/s of shadow bytes (%.3f of 8000)" % (
            2.0 * B * n * dpad / (sc * 1e-3) / 1e12, 2.0 * B * n * dpad / (sc * 1e-3) / 1e12 / 2500,
            n * (2 * d + 8) / (sc * 1e-3) / 1e9, n * (2 * d + 8) / (sc * 1e-3) / 1e9 / 8000)
    print("B=%d %s : %.3f ms/step  %.0f QPS  same_ids=%s  pass(q,fallback)=%s  kernels ms/step %s%s"
          % (B, kv, dt * 1e3, B / dt, same, (f1[0] - f0[0], f1[1] - f0[1]), fam, extra), flush=True)
