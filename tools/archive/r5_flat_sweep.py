"""Round 5: single-query FLAT shadow scan over 1M x 768 -- the knobs of the scan launch side by side (device-resident query, search_device)."""
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
    x = torch.randn((n, d), device=dev, dtype=torch.float32, generator=torch.Generator(device=dev).manual_seed(1234))
    q = torch.randn((256, d), device=dev, dtype=torch.float32, generator=torch.Generator(device=dev).manual_seed(4321))
    fl = capi.Index(capi.INDEX_FLAT, capi.METRIC_L2, d)
    fl.add(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
    fl.build()
    stream = torch.cuda.current_stream().cuda_stream
    oi = torch.empty((64, k), device=dev, dtype=torch.int64)
    od = torch.empty((64, k), device=dev, dtype=torch.float32)
    variants = [{}] + [dict(v.split("=") for v in a.split(",")) for a in sys.argv[1:]]
    for b in (1, 4):
        for opts in variants:
            for k_, v_ in opts.items():
                capi.set_option(k_, v_)
            def st(i):
                fl.search_device(q[(i % 8) * b:(i % 8 + 1) * b].data_ptr(), b, k, 0, oi.data_ptr(), od.data_ptr(), stream)
            for i in range(5):
                st(i)
            torch.cuda.synchronize()
            capi.profile_reset()
            capi.profile_enable(True)
            t = time.perf_counter()
            for i in range(40):
                st(i)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / 40
            capi.profile_enable(False)
            c_, ms = capi.profile_get("flat_shadow_scan")
            c2, ms2 = capi.profile_get("flat_pass")
            capi.profile_reset()
            print("batch %d %-40s %.1f us/step; flat_shadow_scan %.1f us, flat_pass %.1f us" % (b, opts, dt * 1e6, ms / 40 * 1e3, ms2 / 40 * 1e3), flush=True)
            for k_ in opts:
                capi.set_option(k_, None)


if __name__ == "__main__":
    main()
