"""Round 5: one query per call over a FLAT index of 1M x 768 iid rows (the metric's operating point on random vectors): p50 of the
host-pointer call with the fp16 shadow pass (flat_few=1) and with the canonical f32 scan (flat_few=0), small device batches."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myscaledb_amd.capi as capi  # noqa: E402


def main():
    n, d, k = 1_000_000, 768, 10
    dev = torch.device("cuda", 0)
    capi.set_device(0)
    g = torch.Generator(device=dev).manual_seed(1234)
    x = torch.randn((n, d), device=dev, dtype=torch.float32, generator=g)
    q = torch.randn((2048, d), device=dev, dtype=torch.float32, generator=torch.Generator(device=dev).manual_seed(4321))
    fl = capi.Index(capi.INDEX_FLAT, capi.METRIC_L2, d)
    fl.add(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
    fl.build()
    qh = q.cpu().numpy()
    gt, gd = fl.search(qh[:256], k)  # batch path
    stream = torch.cuda.current_stream().cuda_stream
    oi = torch.empty((256, k), device=dev, dtype=torch.int64)
    od = torch.empty((256, k), device=dev, dtype=torch.float32)
    shadow_b = n * (2 * d + 8)
    for few in (("1",) if "--few-only" in sys.argv else ("1", "0")):
        capi.set_option("flat_few", few)
        for i in range(30):
            fl.search(qh[i:i + 1], k)
        lat = np.empty(1500)
        ok = True
        for i in range(1500):
            t = time.perf_counter()
            gi_, gd_ = fl.search(qh[i % 256:i % 256 + 1], k)
            lat[i] = time.perf_counter() - t
            ok = ok and bool((gi_[0] == gt[i % 256]).all()) and bool((gd_[0].view(np.uint32) == gd[i % 256].view(np.uint32)).all())
        p50 = float(np.percentile(lat, 50))
        print("flat_few=%s single query host call: p50 %.1f us p99 %.1f us; shadow bytes / p50 = %.2f TB/s (%.3f of HBM); == batch result: %s"
              % (few, p50 * 1e6, float(np.percentile(lat, 99)) * 1e6, shadow_b / p50 / 1e12, shadow_b / p50 / 8e12, ok), flush=True)
        for b in (1, 4, 16, 64):
            def st(i):
                fl.search_device(q[(i % 8) * b:(i % 8 + 1) * b].data_ptr(), b, k, 0, oi.data_ptr(), od.data_ptr(), stream)
            for i in range(3):
                st(i)
            torch.cuda.synchronize()
            capi.profile_reset()
            capi.profile_enable(True)
            t = time.perf_counter()
            for i in range(20):
                st(i)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / 20
            capi.profile_enable(False)
            c_, ms = capi.profile_get("flat_shadow_scan")
            c2, ms2 = capi.profile_get("flat_pass")
            capi.profile_reset()
            print("  flat_few=%s device batch %3d: %.1f us/step; flat_shadow_scan %.1f us (%d launches), flat_pass %.1f us" % (few, b, dt * 1e6, ms / 20 * 1e3, c_, ms2 / 20 * 1e3), flush=True)
    capi.set_option("flat_few", None)


if __name__ == "__main__":
    main()
