"""How often does the coarse stage of a batched search meet a query whose band cannot be formed (coarse_tail_kernel's slow queue)?
Per data model of the bench: 6 searches of 4096 queries with the inline form switched off, msvs_coarse_stats' fallback counter
counts only the queue's SECOND stage, so the probe reads the stamp words through timing instead: ms per step with the inline form
always on (window 0) / off / default, 70 steps each (the default's retry after 64 searches is inside).
    python tools/slow_queue_probe.py [blobs03|iid|mid|bench]"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import myscaledb_amd.capi as capi  # noqa: E402
from bench import data_model, ivf_params  # noqa: E402

dev = torch.device("cuda", 0)
for name in (sys.argv[1:] or ["blobs03", "mid", "iid"]):
    n, d, nlist, nprobe, k, B = 1_000_000, 768, 1024, 32, 10, 4096
    x, q_all, desc = data_model(name, n, 8 * B, d, dev)
    ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, d, ivf_params(nlist, n, ""))
    ix.train(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
    ix.add(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
    ix.build()
    oi = torch.empty((B, k), device=dev, dtype=torch.int64)
    od = torch.empty((B, k), device=dev, dtype=torch.float32)
    st = torch.cuda.current_stream().cuda_stream

    def run(steps):
        for i in range(3):
            ix.search_device(q_all[(i % 8) * B:(i % 8 + 1) * B].data_ptr(), B, k, nprobe, oi.data_ptr(), od.data_ptr(), st)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(steps):
            ix.search_device(q_all[(i % 8) * B:(i % 8 + 1) * B].data_ptr(), B, k, nprobe, oi.data_ptr(), od.data_ptr(), st)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / steps * 1e3

    res = {}
    for tag, knobs in (("inline always", {"coarse_slow_window": "0"}), ("queue always", {"coarse_slow_inline": "0"}), ("default", {})):
        for a, b in knobs.items():
            capi.set_option(a, b)
        res[tag] = round(run(70), 4)
        for a in knobs:
            capi.set_option(a, None)
    print(name, res, flush=True)
    ix.close()
    del x, q_all
    torch.cuda.empty_cache()
