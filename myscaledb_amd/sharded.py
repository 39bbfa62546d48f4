"""Multi-GPU driver of the hot path: IVF lists (or FLAT row ranges) are sharded one shard per GPU / process, every
rank scans its local shard for the whole query batch, then ONE all-gather of the per-rank partial top-k
(ids i64 + distances f32, nq*k*12 B per rank) and the canonical merge.  This is the GPU analogue of the reference's
per-part search + MergeTreeBaseSearchManager::getTotalTopSearchResultImpl
(src/VectorIndex/Storages/MergeTreeBaseSearchManager.cpp:207-299) and of its Distributed-table scatter/gather.
BM25 adds one all-reduce(sum) of (N, total tokens, df per term) before scoring, mirroring
src/VectorIndex/Common/BM25InfoInDataParts.cpp:40-93.

torch.distributed is only the transport (backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU tests);
the merge runs in libmsvs.so (device tensors) or libmsvs_host.so (host tensors)."""
import ctypes as C

import torch
import torch.distributed as dist

from . import capi


def exchange_and_merge(local_ids, local_dis, metric, group=None, stream=None):
    """local_ids int64 [nq,k], local_dis float32 [nq,k] (same device on every rank) -> merged (ids, dis) on all ranks.
    metric: capi.METRIC_L2 (ascending; also for cosine distances) or capi.METRIC_IP (descending).
    ONE collective either way: the partial lists travel as one packed buffer {ids | dis} per rank (PackedExchange on the
    GPU; on host tensors -- the gloo tests -- one all_gather of the packed bytes, merged by libmsvs_host.so)."""
    world = dist.get_world_size(group)
    nq, k = local_ids.shape
    if local_ids.is_cuda:
        px = PackedExchange(nq, k, local_ids.device, group)
        px.ids.copy_(local_ids)
        px.dis.copy_(local_dis)
        oi, od = px.run(metric, stream)
        return oi.clone(), od.clone()
    from . import host
    packed = torch.cat([local_ids.contiguous().view(torch.uint8).reshape(-1), local_dis.contiguous().view(torch.uint8).reshape(-1)])
    parts = [torch.empty_like(packed) for _ in range(world)]
    dist.all_gather(parts, packed, group=group)
    il = torch.stack([p_[:nq * k * 8].view(torch.int64).view(nq, k) for p_ in parts])
    dl = torch.stack([p_[nq * k * 8:].view(torch.float32).view(nq, k) for p_ in parts])
    oi, od = host.merge_topk(il.numpy(), dl.numpy(), metric)
    return torch.from_numpy(oi), torch.from_numpy(od)


class PackedExchange:
    """Pre-allocated buffers for the steady-state exchange on the GPU: the search writes its partial ids / distances
    into ONE packed buffer {ids[nq*k] i64 | dis[nq*k] f32}, a single RCCL all-gather moves it, and
    msvs_merge_topk_device_strided merges the gathered buffer in place (no repacking kernels, no second collective)."""

    def __init__(self, nq, k, device, group=None):
        self.nq, self.k, self.group = nq, k, group
        self.world = dist.get_world_size(group)
        self.part_bytes = (nq * k * 12 + 7) // 8 * 8
        self.local = torch.zeros(self.part_bytes, dtype=torch.uint8, device=device)
        self.ids = self.local[:nq * k * 8].view(torch.int64).view(nq, k)
        self.dis = self.local[nq * k * 8:nq * k * 12].view(torch.float32).view(nq, k)
        self.gathered = torch.empty(self.world * self.part_bytes, dtype=torch.uint8, device=device)
        self.out_ids = torch.empty((nq, k), dtype=torch.int64, device=device)
        self.out_dis = torch.empty((nq, k), dtype=torch.float32, device=device)

    def run(self, metric, stream=None):
        """all-gather self.local (filled by the search) and merge -> (out_ids, out_dis)."""
        dist.all_gather_into_tensor(self.gathered, self.local, group=self.group)
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        base = self.gathered.data_ptr()
        capi._check(capi.lib().msvs_merge_topk_device_strided(
            C.c_void_p(base), C.c_size_t(self.part_bytes // 8), C.c_void_p(base + self.nq * self.k * 8),
            C.c_size_t(self.part_bytes // 4), C.c_size_t(self.world), C.c_size_t(self.nq), C.c_size_t(self.k),
            int(metric), C.c_void_p(self.out_ids.data_ptr()), C.c_void_p(self.out_dis.data_ptr()),
            C.c_void_p(s) if s else None))
        return self.out_ids, self.out_dis


def all_reduce_bm25_stats(total_docs, total_tokens, df, group=None, device="cpu", comm=None):
    """Sum (N, total tokens, df[terms]) over the ranks: the one exchange step of sharded BM25.  With a libmsvs communicator
    (capi.Comm) the sum runs through the C-ABI the C++ host would call (msvs_host_all_reduce_bm25_stats ->
    msvs_comm_all_reduce_u64); without one (CPU tests over gloo) through torch.distributed."""
    if comm is not None:
        from . import host
        return host.all_reduce_bm25_stats(comm, total_docs, total_tokens, df)
    t = torch.tensor([int(total_docs), int(total_tokens)] + [int(x) for x in df], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    v = t.cpu().tolist()
    return v[0], v[1], v[2:]


class ShardedIndex:
    """An IVFFLAT / FLAT index whose lists live on `world` GPUs (list_id % world == rank stays local)."""

    def __init__(self, index_type, metric, dim, params="", group=None):
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.group = group
        self.metric = metric
        p = (params + "," if params else "") + "shard_rank=%d,shard_world=%d" % (self.rank, self.world)
        self.index = capi.Index(index_type, metric, dim, p)

    def set_centroids(self, centroids):
        self.index.set_centroids(centroids)

    def add(self, x, ids=None, n=None, mem=capi.MEM_HOST):
        self.index.add(x, ids, n=n, mem=mem)

    def build(self):
        self.index.build()

    def search_device(self, q, k, nprobe):
        """q: CUDA float32 tensor [nq, dim].  Returns merged (ids, dis) CUDA tensors, identical on every rank."""
        nq = q.shape[0]
        ids = torch.empty((nq, k), dtype=torch.int64, device=q.device)
        dis = torch.empty((nq, k), dtype=torch.float32, device=q.device)
        stream = torch.cuda.current_stream().cuda_stream
        self.index.search_device(q.data_ptr(), nq, k, nprobe, ids.data_ptr(), dis.data_ptr(), stream)
        order = capi.METRIC_IP if self.metric == capi.METRIC_IP else capi.METRIC_L2
        return exchange_and_merge(ids, dis, order, self.group, stream)


def rccl_comm(group=None):
    """A communicator OWNED BY libmsvs (RCCL through rccl.h, msvs_comm_init) for the ranks of a torch.distributed
    group: torch only carries the 128-byte unique id from rank 0 to the others (bootstrap, not data path)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if world == 1:
        return capi.Comm(1, 0)
    box = [capi.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    return capi.Comm(world, rank, id=box[0])


def torch_comm(group=None):
    """Fallback transport for bench.py: the all-gather of msvs_shard_search_device done by torch.distributed's own NCCL
    (= RCCL) process group on device staging tensors, ordered on the caller's stream (the calls are issued from the thread
    that owns torch's current stream).  Used only when libmsvs could not create its own communicator."""
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    stage = {}

    def all_gather(d_send, d_recv, nbytes, stream):
        if nbytes not in stage:
            stage[nbytes] = (torch.empty(nbytes, dtype=torch.uint8, device="cuda"),
                             torch.empty(nbytes * world, dtype=torch.uint8, device="cuda"))
        snd, rcv = stage[nbytes]
        if hip.hipMemcpyAsync(C.c_void_p(snd.data_ptr()), C.c_void_p(d_send), nbytes, 3, C.c_void_p(stream)) != 0:  # device -> device
            return 1
        dist.all_gather_into_tensor(rcv, snd, group=group)
        return hip.hipMemcpyAsync(C.c_void_p(d_recv), C.c_void_p(rcv.data_ptr()), nbytes * world, 3, C.c_void_p(stream))

    return capi.Comm(world, rank, all_gather=all_gather)


def gloo_comm(group=None):
    """The same communicator with the all-gather done by torch.distributed on HOST copies (gloo): lets two processes
    that share one GPU run the product's sharded search in tests.  Ordered after the stream's work by a device sync."""
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    world, rank = dist.get_world_size(group), dist.get_rank(group)

    def all_gather(d_send, d_recv, nbytes, _stream):
        torch.cuda.synchronize()
        mine = torch.empty(nbytes, dtype=torch.uint8)
        if hip.hipMemcpy(C.c_void_p(mine.data_ptr()), C.c_void_p(d_send), nbytes, 2) != 0:  # device -> host
            return 1
        parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        allb = torch.cat(parts)
        return hip.hipMemcpy(C.c_void_p(d_recv), C.c_void_p(allb.data_ptr()), nbytes * world, 1)  # host -> device

    return capi.Comm(world, rank, all_gather=all_gather)
