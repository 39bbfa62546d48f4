"""Round 5: which synthetic mixture puts recall@10 >= 0.95 at nprobe 4 .. 32 for IVFFLAT nlist 1024 over 1M x 768?
    python tools/r5_mid_model.py blobs:scale:latent_dim[:spread] ...      (recall by nprobe + the pruning's share per model)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myscaledb_amd.capi as capi  # noqa: E402
import bench  # noqa: E402


def main():
    n, d, k, nlist, B = 1_000_000, 768, 10, 1024, 4096
    dev = torch.device("cuda", 0)
    capi.set_device(0)
    stream = torch.cuda.current_stream().cuda_stream
    oi = torch.empty((B, k), device=dev, dtype=torch.int64)
    od = torch.empty((B, k), device=dev, dtype=torch.float32)
    for spec in sys.argv[1:]:
        f = spec.split(":")
        blobs, scale, ld = int(f[0]), float(f[1]), int(f[2])
        g = torch.Generator(device=dev).manual_seed(99)
        centres = scale * torch.randn((blobs, ld), generator=g, device=dev, dtype=torch.float32)
        proj = torch.randn((ld, d), generator=g, device=dev, dtype=torch.float32) / (ld ** 0.5)

        def sample(m, seed):
            gg = torch.Generator(device=dev).manual_seed(seed)
            x = torch.empty((m, d), device=dev, dtype=torch.float32)
            for lo in range(0, m, 65536):
                hi = min(m, lo + 65536)
                z = torch.randint(0, blobs, (hi - lo,), generator=gg, device=dev)
                lat = centres[z] + torch.randn((hi - lo, ld), generator=gg, device=dev, dtype=torch.float32)
                x[lo:hi] = lat @ proj + 0.05 * torch.randn((hi - lo, d), generator=gg, device=dev, dtype=torch.float32)
            return x
        x, q = sample(n, 1234), sample(B, 4321)
        ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, d, bench.ivf_params(nlist, n))
        ix.train(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
        ix.add(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
        ix.build()
        fl = capi.Index(capi.INDEX_FLAT, capi.METRIC_L2, d)
        fl.add(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
        fl.build()
        qh = q[:1000].cpu().numpy()
        gt, _ = fl.search(qh, k)
        out = []
        for npb in (1, 2, 4, 8, 16, 32, 64):
            got, _ = ix.search(qh, k, "nprobe=%d" % npb)
            r = bench.recall_at_k(got, gt, k)
            p0 = capi.debug_prune_stats()
            ix.search_device(q.data_ptr(), B, k, npb, oi.data_ptr(), od.data_ptr(), stream)
            torch.cuda.synchronize()
            p1 = capi.debug_prune_stats()
            dp = [b - a for a, b in zip(p0, p1)]
            out.append("np%d: recall %.3f prune %s" % (npb, r, dp))
        print(spec, ix.list_stats(), flush=True)
        for o_ in out:
            print("   ", o_, flush=True)
        ix.close()
        fl.close()
        del x, q


if __name__ == "__main__":
    main()
