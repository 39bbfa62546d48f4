"""One shard of an 8-way sharded index (lists list_id % 8 == 0 of the bench index) searched by 4096 queries that belong to ITS lists:
what the searching side of a routed step scans (32 queries per list, ~150 survivors each: the launch is its survivors' appends, not
its 190 MB of rows).
    python tools/shard_scan_probe.py"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import myscaledb_amd.capi as capi  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
capi.set_device(0)
n, d, nlist, nprobe, k, B, W = 1_000_000, 768, 1024, 32, 10, 4096, 8
x, q_all, _ = bench.data_model("blobs03", n, 16 * B, d, dev)
t = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, d, bench.ivf_params(nlist, n))
t.train(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
t.add(x[:nlist].contiguous().data_ptr(), n=nlist, mem=capi.MEM_DEVICE)
t.build()
cent = t.export()[0]
t.close()
ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, d, bench.ivf_params(nlist, n, ",shard_rank=0,shard_world=%d" % W))
ix.set_centroids(cent)
ix.add(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
ix.build()
# queries whose nearest centroid is one of this shard's lists
c = torch.from_numpy(cent).to(dev)
near = torch.cdist(q_all, c).argmin(1)
mine = q_all[(near % W) == 0][:2 * B].contiguous()
oi = torch.empty((B, k), device=dev, dtype=torch.int64)
od = torch.empty((B, k), device=dev, dtype=torch.float32)
st = torch.cuda.current_stream().cuda_stream
for tag, knobs in (("default", {}), ("grouped appends up to 8-query tiles (round 6's first value)", {"h16_group_appends": "8"}), ("one atomic per record", {"h16_group_appends": "0"}),
                   ("target 100", {"h16_target": "100"})):
    for a, b in knobs.items():
        capi.set_option(a, b)
    for i in range(4):
        ix.search_device(mine[(i % 2) * B:(i % 2 + 1) * B].data_ptr(), B, k, nprobe, oi.data_ptr(), od.data_ptr(), st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        ix.search_device(mine[(i % 2) * B:(i % 2 + 1) * B].data_ptr(), B, k, nprobe, oi.data_ptr(), od.data_ptr(), st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    capi.profile_reset(); capi.profile_enable(True)
    for i in range(4):
        ix.search_device(mine[(i % 2) * B:(i % 2 + 1) * B].data_ptr(), B, k, nprobe, oi.data_ptr(), od.data_ptr(), st)
    torch.cuda.synchronize()
    capi.profile_enable(False)
    fam = {f: round(capi.profile_get(f)[1] / 4, 4) for f in ("ivf_scan", "ivf_sample_scan", "ivf_plan", "rerank")}
    print("%s: %.4f ms per step, families %s" % (tag, dt * 1e3, fam))
    for a in knobs:
        capi.set_option(a, None)
