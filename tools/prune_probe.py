"""Why do (query, list) pairs survive the probe pruning on SURVEY 8d's blobs?  Restates the rule in numpy on the exported index
(1M x 768, 1024 blobs sigma 0.3, nlist 1024, nprobe 32) for a few hundred queries and prints what the survivors look like."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myscaledb_amd.capi as capi  # noqa: E402
from bench import data_model, ivf_params  # noqa: E402

dev = torch.device("cuda", 0)
n, d, nlist, nprobe, k, nq = 1_000_000, 768, 1024, 32, 10, 512
x, q, _ = data_model("blobs03", n, nq, d, dev)
ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, d, ivf_params(nlist, n, os.environ.get("PROBE_PARAMS", "")))
ix.train(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
ix.add(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
ix.build()
cent, off, _, lids = ix.export(with_vecs=False)
centres = torch.randn((1024, d), generator=torch.Generator(device=dev).manual_seed(99), device=dev, dtype=torch.float32)
xs = x[torch.from_numpy(lids).to(dev)]  # rows in storage order
ct = torch.from_numpy(cent).to(dev)
# blob of every stored row (nearest true centre), per-list purity and radius
lens = np.diff(off)
blob = torch.empty(n, dtype=torch.int64, device=dev)
for lo in range(0, n, 65536):
    hi = min(n, lo + 65536)
    blob[lo:hi] = torch.cdist(xs[lo:hi], centres).argmin(1)
list_of = torch.from_numpy(np.repeat(np.arange(nlist), lens)).to(dev)
rad = torch.zeros(nlist, device=dev)
dist_c = (xs - ct[list_of]).norm(dim=1)
rad.scatter_reduce_(0, list_of, dist_c, reduce="amax")
nblobs = np.array([len(torch.unique(blob[off[l]:off[l + 1]])) for l in range(nlist)])
major = np.array([float((blob[off[l]:off[l + 1]] == torch.mode(blob[off[l]:off[l + 1]]).values).float().mean()) if lens[l] else 0 for l in range(nlist)])
print("list stats", ix.list_stats())
print("lists by number of blobs they hold:", {int(b): int((nblobs == b).sum()) for b in np.unique(nblobs)})
print("radius: p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile(rad.cpu().numpy(), [10, 50, 90, 100])))
dqc = torch.cdist(q, ct)  # [nq, nlist]
probes = dqc.topk(nprobe, largest=False).indices
surv_lists = []
kept = 0
for i in range(nq):
    pl = probes[i].tolist()
    samp = torch.cat([xs[off[l]:min(off[l] + 32, off[l + 1])] for l in pl])
    ds = ((samp - q[i]) ** 2).sum(1)
    U = ds.kthvalue(k).values.item()
    for l in pl:
        dc = dqc[i, l].item()
        r = rad[l].item()
        keep = not (dc > r and (dc - r) ** 2 > U)
        if keep:
            kept += 1
            surv_lists.append(l)
surv_lists = np.array(surv_lists)
print("pairs kept %d of %d (%.3f); per query %.1f" % (kept, nq * nprobe, kept / (nq * nprobe), kept / nq))
print("kept pairs by blobs-in-list:", {int(b): int((nblobs[surv_lists] == b).sum()) for b in np.unique(nblobs[surv_lists])})
print("kept pairs: list radius p10/p50/p90 %.1f/%.1f/%.1f, list length p50 %d" % (*np.percentile(rad.cpu().numpy()[surv_lists], [10, 50, 90]), np.median(lens[surv_lists])))
hot = np.bincount(surv_lists, minlength=nlist)
top = np.argsort(-hot)[:10]
print("hottest lists (kept pairs of %d queries, blobs, len, radius, majority share):" % nq,
      [(int(hot[l]), int(nblobs[l]), int(lens[l]), round(float(rad[l]), 1), round(major[l], 2)) for l in top])
print("lists with kept pairs from >= 5%% of the queries: %d" % int((hot >= 0.05 * nq).sum()))
