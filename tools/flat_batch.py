"""Exhaustive FLAT shadow pass, 4096 queries over 1M x 768 iid rows (the metric's operating point on random vectors): time per step,
fraction of the dense fp16 MFMA peak.   python tools/flat_batch.py [steps] [batch] [opt=val,opt=val ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myscaledb_amd.capi as capi  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    n, d, k = int(os.environ.get("FLAT_ROWS", 1_000_000)), 768, 10
    dev = torch.device("cuda", 0)
    capi.set_device(0)
    x = torch.randn((n, d), device=dev, dtype=torch.float32, generator=torch.Generator(device=dev).manual_seed(1234))
    if os.environ.get("FLAT_ZERO"):
        x.zero_()
    q = torch.randn((2 * B, d), device=dev, dtype=torch.float32, generator=torch.Generator(device=dev).manual_seed(4321))
    fl = capi.Index(capi.INDEX_FLAT, capi.METRIC_L2, d)
    fl.add(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
    fl.build()
    stream = torch.cuda.current_stream().cuda_stream
    oi = torch.empty((B, k), device=dev, dtype=torch.int64)
    od = torch.empty((B, k), device=dev, dtype=torch.float32)
    variants = [{}] + [dict(v.split("=") for v in a.split(",")) for a in sys.argv[3:]]
    for opts in variants:
        for k_, v_ in opts.items():
            capi.set_option(k_, v_)

        def st(i):
            fl.search_device(q[(i % 2) * B:(i % 2 + 1) * B].data_ptr(), B, k, 0, oi.data_ptr(), od.data_ptr(), stream)
        for i in range(2):
            st(i)
        torch.cuda.synchronize()
        capi.profile_reset()
        capi.profile_enable(True)
        t = time.perf_counter()
        for i in range(steps):
            st(i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / steps
        capi.profile_enable(False)
        _, ms = capi.profile_get("flat_shadow_scan")
        capi.profile_reset()
        fl_ = 2.0 * B * n * d
        print("batch %d %-30s %.3f ms/step (%.0f TF/s, %.3f of 2500); flat_shadow_scan %.3f ms (%.3f)" % (
            B, opts, dt * 1e3, fl_ / dt / 1e12, fl_ / dt / 2.5e15, ms / steps, fl_ / (ms / steps * 1e-3) / 2.5e15 if ms else 0), flush=True)
        for k_ in opts:
            capi.set_option(k_, None)
    if os.environ.get("FLAT_CHECK"):
        ids = oi.cpu().numpy()
        ref = torch.cdist(q[(steps - 1) % 2 * B:(steps - 1) % 2 * B + 64], x).topk(k, largest=False).indices.cpu().numpy()
        print("top-k agreement with torch.cdist on 64 queries:", float((ids[:64] == ref).mean()))


if __name__ == "__main__":
    main()
