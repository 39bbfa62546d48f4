"""Round 5: NQ (argv 1, default 32) queries through the device entry, back to back -- the launch sequence of a small batch (kernel trace)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myscaledb_amd.capi as capi  # noqa: E402
import bench  # noqa: E402


def main():
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    n, d, k, nlist, nprobe = 1_000_000, 768, 10, 1024, 32
    dev = torch.device("cuda", 0)
    capi.set_device(0)
    x, q, _ = bench.data_model("blobs03", n, 4096, d, dev)
    ix = capi.Index(capi.INDEX_IVFFLAT, capi.METRIC_L2, d, bench.ivf_params(nlist, n))
    ix.train(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
    ix.add(x.data_ptr(), n=n, mem=capi.MEM_DEVICE)
    ix.build()
    stream = torch.cuda.current_stream().cuda_stream
    oi = torch.empty((4096, k), device=dev, dtype=torch.int64)
    od = torch.empty((4096, k), device=dev, dtype=torch.float32)
    for i in range(30):
        ix.search_device(q[(i % 16) * nq:(i % 16 + 1) * nq].data_ptr(), nq, k, nprobe, oi.data_ptr(), od.data_ptr(), stream)
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
