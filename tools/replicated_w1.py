import sys, time, torch
sys.path.insert(0, "/root/repo")
import os
sys.path.insert(0, os.getcwd())
import myscaledb_amd.capi as capi
import bench
dev = torch.device("cuda", 0); capi.set_device(0)
n, d, nlist, nprobe, k, B = 1_000_000, 768, 1024, 32, 10, 4096
x, q_all, _ = bench.data_model("blobs03", n, 8 * B, d, dev)
ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, d, bench.ivf_params(nlist, n))
ix.train(x.data_ptr(), n=n, mem=capi.MEM_DEVICE); ix.add(x.data_ptr(), n=n, mem=capi.MEM_DEVICE); ix.build()
comm = capi.Comm(1, 0)
st = torch.cuda.current_stream().cuda_stream
oi = torch.empty((B, k), device=dev, dtype=torch.int64); od = torch.empty((B, k), device=dev, dtype=torch.float32)
for name, fn in (("replicated async", lambda qp: ix.shard_search_device_async(comm, qp, B, k, nprobe, oi.data_ptr(), od.data_ptr(), st)),
                 ("replicated sync", lambda qp: ix.shard_search_device(comm, qp, B, k, nprobe, oi.data_ptr(), od.data_ptr(), st))):
    for i in range(5):
        fn(q_all[(i % 8) * B:(i % 8 + 1) * B].data_ptr())
    comm.drain(st); torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(20):
        fn(q_all[(i % 8) * B:(i % 8 + 1) * B].data_ptr())
    comm.drain(st); torch.cuda.synchronize()
    print("%s W = 1: %.4f ms per step" % (name, (time.perf_counter() - t) / 20 * 1e3))
