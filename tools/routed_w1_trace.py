"""One rank, routed search through the REAL RCCL group path (option route_self_rccl: the rank's own piece travels through
ncclSend / ncclRecv to itself): the launch sequence of a routed step for the kernel trace (tools/prof_cmd.sh), and its time.
    python tools/routed_w1_trace.py [async]"""
import ctypes
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import myscaledb_amd.capi as capi  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
capi.set_device(0)
n, d, nlist, nprobe, k, B = 1_000_000, 768, 1024, 32, 10, 4096
x, q_all, _ = bench.data_model("blobs03", n, 8 * B, d, dev)
ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, d, bench.ivf_params(nlist, n))
ix.train(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
ix.add(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
ix.build()
capi.set_option("route_self_rccl", "1")
comm = capi.Comm(1, 0)
st = torch.cuda.current_stream().cuda_stream
slots = [(torch.empty((B, k), device=dev, dtype=torch.int64), torch.empty((B, k), device=dev, dtype=torch.float32), ctypes.c_uint64(0)) for _ in range(3)]
use_async = len(sys.argv) > 1 and sys.argv[1] == "async"


def step(i):
    s = slots[i % 3]
    qp = q_all[(i % 8) * B:(i % 8 + 1) * B].data_ptr()
    if use_async:
        ix.shard_search_routed_device_async(comm, qp, B, k, nprobe, s[0].data_ptr(), s[1].data_ptr(), st, served=s[2], want_event=False)
    else:
        ix.shard_search_routed_device(comm, qp, B, k, nprobe, s[0].data_ptr(), s[1].data_ptr(), st)


for i in range(5):
    step(i)
comm.drain(st)
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(20):
    step(i)
comm.drain(st)
torch.cuda.synchronize()
print("routed W = 1 (%s): %.4f ms per 4096-query step" % ("async" if use_async else "sync", (time.perf_counter() - t) / 20 * 1e3))
ix.search_device(q_all[:B].data_ptr(), B, k, nprobe, slots[0][0].data_ptr(), slots[0][1].data_ptr(), st)
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(20):
    ix.search_device(q_all[(i % 8) * B:(i % 8 + 1) * B].data_ptr(), B, k, nprobe, slots[0][0].data_ptr(), slots[0][1].data_ptr(), st)
torch.cuda.synchronize()
print("plain search_device: %.4f ms per step" % ((time.perf_counter() - t) / 20 * 1e3))
