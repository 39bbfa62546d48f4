"""Routed sharded search on ONE GPU shared by W gloo processes: how many (query, rank) pairs does a step route, step by step?
    python tools/routed_probe.py W [rows=200000] [batch=2048]
(the bench data model, d = 768, 1024 lists, nprobe 32; not a measurement of speed: the exchange goes through the host)"""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, n, B):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import myscaledb_amd.capi as capi
    from myscaledb_amd import sharded
    import bench
    capi.set_device(0)
    dev = torch.device("cuda", 0)
    d, nlist, nprobe, k = 768, 1024, 32, 10
    x, q_all, _ = bench.data_model("blobs03", n, 8 * B, d, dev)
    t = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, d, bench.ivf_params(nlist, n))
    t.train(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
    t.add(x[:nlist].contiguous().data_ptr(), n=nlist, mem=capi.MEM_DEVICE)
    t.build()
    cent = t.export()[0]
    t.close()
    ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, d, bench.ivf_params(nlist, n, ",shard_rank=%d,shard_world=%d" % (rank, world)))
    ix.set_centroids(cent)
    ix.add(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
    ix.build()
    comm = sharded.gloo_comm()
    oi = torch.empty((B, k), device=dev, dtype=torch.int64)
    od = torch.empty((B, k), device=dev, dtype=torch.float32)
    capi.set_option("rerank_stats", "1")
    out = []
    for i in range(6):
        j = (i * world + rank) % 8
        p0 = capi.debug_prune_stats()
        served = ix.shard_search_routed_device(comm, q_all[j * B:(j + 1) * B].data_ptr(), B, k, nprobe, oi.data_ptr(), od.data_ptr())
        torch.cuda.synchronize()
        p1 = capi.debug_prune_stats()
        out.append((served, p1[0] - p0[0], p1[1] - p0[1]))
    print("rank %d of %d: (served, pairs dropped, pairs seen) per step: %s" % (rank, world, out), flush=True)
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    W = int(sys.argv[1])
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(worker, args=(W, port, n, B), nprocs=W, join=True)
