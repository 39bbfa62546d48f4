// latency_kernels.hpp -- the few-query (nq <= 4) IVFFLAT search in TWO launches (and, at the end, the one-launch coarse quantiser of
// small batches built from the same pieces: coarse_few_kernel).
//
// A single query of BASELINE config 2 moves ~99 MB (3 MB of centroids + 32 lists): 20 us of HBM time.  The general path
// spends four launches on it (centroid scan, merge, list scan, merge); the two one-block merge kernels alone are 40 of
// its 80 us -- a cold single block pays every latency (instruction fetch, first loads, LDS round trips) back to back.
//
//   lat_coarse_kernel : grid (c_blocks, nq).  Block b: canonical distances of c_rows centroids to the query (scan_rows,
//                       the same arithmetic as every canonical scan: the probes equal the oracle's) -> the block's
//                       top-nprobe list; block 0 also copies the query into device memory (it may live in pinned host
//                       memory).  Every block announces itself on a device counter; the LAST one to arrive merges the
//                       lists -> probes[nq][nprobe].
//   lat_scan_kernel   : grid (items + nprobe, nq).  Block i scans the i-th equal slice of the query's probed rows
//                       (scan_rows), publishes its top-k and arrives; the last block merges the slices' lists -> ids /
//                       distances (device or pinned host memory) and flips the completion word a host thread may be
//                       spinning on.
//   (Tried: no merge in stage 1, every stage-2 block merging the centroid lists itself -- 11 us in front of every
//   block's scan instead of 13 us once.)
// Visibility across the 8 XCDs (one L2 each) without fences: a block's list goes out with agent-scope (write-through)
// stores that are waited for (vmcnt) before the block's arrival, and the last block reads the lists with agent-scope
// loads.  A __threadfence() per block writes back / invalidates a whole L2 each time: 185 us per search measured.
// Results are bit-identical to the general path's (same scan_rows, same total order of keys).
// profiles/r02_latency.txt has the steps that led here.
#pragma once

#include "mfma_scan_kernels.hpp"

#pragma clang fp contract(off)

namespace msvs
{

constexpr uint32_t LAT_MAX_Q = 4;  // queries per call on this path
constexpr uint32_t LAT_MAX_K = 64; // k and nprobe (one register of a wavefront top-k)
constexpr uint32_t LAT_COARSE_MAX_Q = 128; // coarse_few_kernel: queries per call
constexpr int LAT_COARSE_T = 4;            // ... and per block (one wavefront each in the selection)

struct LatParams
{
    const float4 * Q; // [nq][ld4] scan-ready queries (padded; normalised for cosine), any device-visible memory
    float4 * dq;      // [nq][ld4] device copy made by stage 1
    uint32_t nq, ld4, k, nprobe, nlist;
    // stage 1
    const float4 * C;     // centroids [nlist][ld4]
    uint32_t c_rows;      // centroids per block (a multiple of 16), c_blocks * nprobe <= HEADS_CAP
    uint32_t c_blocks;    // ceil(nlist / c_rows)
    uint64_t * c_partial; // [nq][c_blocks][nprobe]
    int32_t * probes;     // [nq][nprobe]
    uint32_t * cut;       // [nq][nprobe + 3]: stage 1's last block leaves stage 2's work partition here (lat_cut: items before list p,
                          // then their total, then the rows per item) -- every stage-2 block read it off the probe list itself before,
                          // a serial loop of nprobe dependent loads per block (2.9 us of a 35 us search)
    float * probe_dis;    // [nq][nprobe] canonical distance of the query to the probe's centroid (what the radius pruning reads)
    // radius pruning of stage 2 (round 4; L2, no filter; nullptr: off): lat_cut gives no work to a probed list that provably holds
    // none of the k nearest rows -- (||q - c_l|| - r_l)^2 beyond the smallest (||q - c_p|| + r_p)^2 over probed lists of >= k rows
    const float * radius; // [nlist]
    float cmax, xmax;     // max |c|^2, max |x|^2
    double c_canon;       // relative rounding of a canonical distance
    // stage 2
    const float4 * Y;
    const uint32_t * ids;
    const int64_t * list_off;
    const uint64_t * alive;
    uint32_t nbits;
    uint32_t items;     // work items a query's probed rows are cut into (lat_cut); the grid has items + nprobe of them
    uint32_t * hint_rows; // nullable, pinned host memory: stage 1 leaves the rows the query's surviving lists hold (the host sizes the
                          // NEXT call's stage-2 grid by it: a hint, read without synchronisation)
    uint64_t * partial; // [nq][items + nprobe][k]
    int64_t * out_ids;  // [nq][k]
    float * out_dis;
    int cosine;
    int reg_select;  // stage 1: probe list by the register selection (lat_select_probes) instead of the list merge
    uint32_t * done; // [2] arrival counters of the two stages; zero between calls
    uint32_t * done_q; // coarse_few_kernel: [ceil(nq / T)] arrival counters of the query groups; zero between calls
    uint32_t * flag; // nullable: host-visible completion word, set to `seq` after the results are written
    uint32_t seq;
    unsigned long long * dbg; // nullable: [16] wall-clock stamps (100 MHz) of the two last blocks (experiments)
};

/// All BLOCK threads: publish this block's list (k keys in LDS) and report whether it is the last block to arrive.
__device__ __forceinline__ bool lat_publish_and_arrive(const uint64_t * lds_list, uint64_t * dst, uint32_t k, uint32_t * counter,
                                                       uint32_t expected)
{
    __shared__ uint32_t s_last;
    if (dst && threadIdx.x < k)
        __hip_atomic_store(dst + threadIdx.x, lds_list[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0); // the write-through stores are acknowledged
    __syncthreads();
    if (threadIdx.x == 0)
        s_last = atomicAdd(counter, 1u) == expected - 1;
    __syncthreads();
    return s_last != 0;
}

__device__ __forceinline__ uint64_t lat_load_key(const uint64_t * p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

/// Top-k of n_lists ascending lists of k keys each (global memory), by all BLOCK threads; returns the merged list (LDS,
/// k keys, KEY_NONE padded).  lds: lat_merge_lds(n_lists, k) bytes.  FRESH: the lists were written by other blocks of
/// this launch (agent-scope loads); else by an earlier launch (plain loads).
/// Few enough keys: all of them into LDS with every load in flight at once, then ONE wavefront pops the k smallest list
/// heads (wave_heads_merge: a wave-wide minimum and one lane's rescan per result).  More (large k): wavefront top-k over
/// all keys, 8 loads in flight per thread, + rank merge.
/// A loop of "load one key, offer it" pays a memory round trip per key: 25 us for 10k keys.
inline size_t lat_merge_lds(uint32_t n_lists, uint32_t k)
{
    const size_t total = (size_t)n_lists * k;
    return total <= HEADS_CAP ? total * 8 + (size_t)k * 8 + (size_t)n_lists * 2 + 16 : (size_t)5 * k * 8;
}

/// wave_heads_merge with the list heads in REGISTERS: lane l owns lists l, l + 64, ... (at most 8: P <= 512) and keeps
/// their current heads; a round is a wave-wide minimum, then the winning lane advances ONE list (one LDS read) and
/// re-minimises its 8 registers -- against a rescan of its lists through two dependent LDS reads each.
__device__ __forceinline__ void wave_heads_merge_regs(const uint64_t * keys, uint32_t P, uint32_t L, uint64_t * out, uint32_t k,
                                                      uint32_t lane)
{
    uint64_t head[8];
    uint32_t pos[8];
#pragma unroll
    for (int j = 0; j < 8; j++)
    {
        const uint32_t l = lane + 64 * j;
        head[j] = l < P ? keys[(size_t)l * L] : KEY_NONE;
        pos[j] = 0;
    }
    auto best_of = [&]() {
        uint64_t b = head[0];
#pragma unroll
        for (int j = 1; j < 8; j++)
            b = head[j] < b ? head[j] : b;
        return b;
    };
    uint64_t best = best_of();
    for (uint32_t r = 0; r < k; r++)
    {
        const uint64_t m = wave_min_u64(best);
        if (lane == 0)
            out[r] = m;
        // equal keys in two lanes (the same id at the same distance in two lists): one copy per round, lowest lane first
        const uint64_t tie = __ballot(m != KEY_NONE && best == m);
        if (tie && lane == (uint32_t)__builtin_ctzll(tie))
        {
            bool done = false;
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (!done && head[j] == m)
                {
                    pos[j]++;
                    head[j] = pos[j] < L ? keys[(size_t)(lane + 64 * j) * L + pos[j]] : KEY_NONE;
                    done = true;
                }
            best = best_of();
        }
    }
}

/// max_lists: what the launch sized the LDS for (lat_merge_lds(max_lists, k)); the path is chosen from IT, not from
/// the lists actually present.
template <bool FRESH>
__device__ __forceinline__ uint64_t * lat_merge_lists(const uint64_t * src, uint32_t n_lists, uint32_t max_lists, uint32_t k,
                                                      uint64_t * lds)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t total = n_lists * k;
    auto load = [&](const uint64_t * p) { return FRESH ? lat_load_key(p) : *p; };
    __syncthreads(); // whoever used this LDS region before is done
    if (max_lists * k <= HEADS_CAP)
    {
        uint64_t * keys = lds;
        uint64_t * outk = lds + total;
        uint16_t * idx = reinterpret_cast<uint16_t *>(outk + k);
        // HEADS_CAP = 24 keys per thread: every load of the merge in flight at once (three batches of 8 cost three memory
        // round trips, ~2 us each through the L2-bypassing loads)
        static_assert(HEADS_CAP == 24 * BLOCK, "one batch covers the staged keys");
        uint64_t key[24];
#pragma unroll
        for (int u = 0; u < 24; u++)
        {
            const uint32_t i = u * BLOCK + tid;
            key[u] = i < total ? load(src + i) : KEY_NONE;
        }
#pragma unroll
        for (int u = 0; u < 24; u++)
        {
            const uint32_t i = u * BLOCK + tid;
            if (i < total)
                keys[i] = key[u];
        }
        __syncthreads();
        if (wave == 0)
        {
            if (n_lists > 64 && n_lists <= 512) // several lists per lane: register heads (6.9 us against 12.7 for 512 x 10;
                                                // one list per lane, 32 x 32: 14.4 against 11.4)
                wave_heads_merge_regs(keys, n_lists, k, outk, k, lane);
            else
                wave_heads_merge(keys, n_lists, k, idx, outk, k, lane);
        }
        __syncthreads();
        return outk;
    }
    // more keys than the LDS of a block takes (large k): wavefront top-k over all of them, 8 loads in flight per thread
    WaveTopK<1> top;
    top.init();
    for (uint32_t i0 = 0; i0 < total; i0 += 8 * BLOCK)
    {
        uint64_t key[8];
#pragma unroll
        for (int u = 0; u < 8; u++)
        {
            const uint32_t i = i0 + u * BLOCK + tid;
            key[u] = i < total ? load(src + i) : KEY_NONE;
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
            top.offer(key[u], k, lane);
    }
    top.store(lds + wave * k, k, lane);
    __syncthreads();
    uint64_t * merged = lds + 4 * k;
    block_rank_merge(lds, k, merged, k, tid);
    return merged;
}

/// The probe list of one query by ONE wavefront without any merge: all `total` (<= 64 NW) centroid keys of the query in
/// registers (one batch of agent-scope loads), the np-th smallest distance word by wave_kth_word (32 rounds of NW compares,
/// no LDS, no barrier), ties at that word broken exactly by a second search over the ids (the probes must be the
/// oracle's: smallest (distance, id) first) -- 11.3 -> ~4 us for 1024 keys against popping 32 heads off 32 sorted
/// lists.  The probes leave in arbitrary order (the list scan's result does not depend on it).
template <int NW, int METRIC = -1>
__device__ inline void lat_select_probes(const uint64_t * src, uint32_t total, uint32_t np, int32_t * probes, float * dis, uint32_t lane,
                                         uint32_t * hist)
{
    uint32_t hi[NW], lo[NW];
#pragma unroll
    for (int u = 0; u < NW; u++)
    {
        const uint32_t i = u * WAVE + lane;
        const uint64_t key = i < total ? lat_load_key(src + i) : KEY_NONE;
        hi[u] = (uint32_t)(key >> 32);
        lo[u] = (uint32_t)key;
    }
    const uint32_t H = wave_kth_word<NW>(hi, np, hist, lane);
    uint32_t below = 0, ties = 0;
#pragma unroll
    for (int u = 0; u < NW; u++)
    {
        below += (uint32_t)__popcll(__ballot(hi[u] < H));
        ties += (uint32_t)__popcll(__ballot(hi[u] == H));
    }
    uint32_t L = 0xFFFFFFFFu; // ties with id <= L are taken
    if (below + ties > np)    // the tie group straddles the np-th place: its (np - below) smallest ids
    {
        uint32_t tl[NW];
#pragma unroll
        for (int u = 0; u < NW; u++)
            tl[u] = hi[u] == H ? lo[u] : 0xFFFFFFFFu;
        L = wave_kth_word<NW>(tl, np - below, hist, lane);
    }
    uint32_t run = 0;
#pragma unroll
    for (int u = 0; u < NW; u++)
    {
        const bool take = hi[u] < H || (hi[u] == H && lo[u] <= L);
        const uint64_t mask = __ballot(take);
        const uint32_t pos = run + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        if (take && pos < np)
        {
            probes[pos] = hi[u] == 0xFFFFFFFFu && lo[u] == 0xFFFFFFFFu ? -1 : (int32_t)lo[u];
            // (the two-launch search: an L2 key's high word, read by the radius pruning of L2 searches only; METRIC given: the key's
            // value as the merge launch of the batched coarse quantiser leaves it -- the general path's pre-pruning reads it for
            // L2 and cosine searches)
            if constexpr (METRIC < 0)
                dis[pos] = ord2f(hi[u]);
            else
                dis[pos] = key_value<METRIC>((uint64_t)hi[u] << 32 | lo[u]);
        }
        run += (uint32_t)__popcll(mask);
    }
}

/// The rows of a query's probed lists are cut into work items of equal size WHATEVER the list lengths are (a grid shaped
/// by the longest list leaves half its blocks without rows): rows per item = the probed rows / `items`, rounded up to the
/// 16 rows a block scans per step; list p gets ceil(len_p / rpb) items, at most items + nprobe in total.  Computed ONCE, by one
/// wavefront of stage 1's last block (lat_make_cut: lane = probe, wave-wide sums), read back by every block of stage 2.
struct LatCut
{
    uint32_t rpb, total;
};
__device__ __forceinline__ void lat_make_cut(const LatParams & p, uint32_t q, uint32_t lane)
{
    const int32_t l = lane < p.nprobe ? p.probes[(size_t)q * p.nprobe + lane] : -1;
    const uint32_t len = l >= 0 ? (uint32_t)(p.list_off[l + 1] - p.list_off[l]) : 0u;
    uint64_t rows = len;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
        rows += (uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)rows, o) | ((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(rows >> 32), o) << 32);
    if (p.hint_rows && lane == 0)
        __hip_atomic_store(p.hint_rows, rows > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    uint32_t rpb = (uint32_t)((rows + p.items - 1) / p.items);
    rpb = rpb < 16 ? 16 : (rpb + 15) / 16 * 16;
    const uint32_t mine = (len + rpb - 1) / rpb;
    uint32_t incl = mine; // inclusive prefix sum over the lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1)
    {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
        if (lane >= (uint32_t)o)
            incl += up;
    }
    uint32_t * dst = p.cut + (size_t)q * (p.nprobe + 3);
    if (lane < p.nprobe)
        dst[lane] = incl - mine;
    const uint32_t total = (uint32_t)__shfl((int)incl, 63);
    if (lane == 0)
    {
        dst[p.nprobe] = total;
        dst[p.nprobe + 1] = rpb;
    }
}
__device__ __forceinline__ LatCut lat_cut(const LatParams & p, uint32_t q, int32_t * s_probe /* [nprobe] */,
                                          uint32_t * s_first /* [nprobe + 1] */)
{
    __shared__ uint32_t s_rpb;
    const uint32_t tid = threadIdx.x;
    const uint32_t * src = p.cut + (size_t)q * (p.nprobe + 3);
    __syncthreads();
    if (tid < p.nprobe)
        s_probe[tid] = p.probes[(size_t)q * p.nprobe + tid];
    if (tid <= p.nprobe)
        s_first[tid] = src[tid];
    if (tid == 0)
        s_rpb = src[p.nprobe + 1];
    __syncthreads();
    return LatCut{s_rpb, s_first[p.nprobe]};
}

/// dynamic LDS: ld4 * 16 + max(5 * nprobe * 8, lat_merge_lds(c_blocks, nprobe)) bytes
template <int METRIC>
__global__ __launch_bounds__(BLOCK) void lat_coarse_kernel(const LatParams p)
{
    constexpr int SELM = -1; // (probe distances as L2 key words: lat_select_probes)
    float4 * qs = reinterpret_cast<float4 *>(msvs_smem);
    uint64_t * lds_merge = reinterpret_cast<uint64_t *>(msvs_smem + (size_t)p.ld4 * 16);
    const uint32_t tid = threadIdx.x, q = blockIdx.y, b = blockIdx.x, np = p.nprobe;
    ScanParams a{};
    a.Y = p.C;
    a.Q = p.Q;
    a.ld4 = p.ld4;
    a.k = np;
    a.nq = p.nq;
    __shared__ uint64_t s_list[LAT_MAX_K];
    const unsigned long long t0 = wall_clock64();
    uint32_t qidx[1] = {q};
    uint64_t * out[1] = {s_list};
    stage_queries<1>(a, qidx, qs);
    if (b == 0)
        for (uint32_t c = tid; c < p.ld4; c += BLOCK)
            p.dq[(size_t)q * p.ld4 + c] = qs[c];
    const uint32_t rb = b * p.c_rows, re = rb + p.c_rows < p.nlist ? rb + p.c_rows : p.nlist;
    scan_rows<METRIC, 1, 1>(a, rb, re, qs, lds_merge, out);
    __syncthreads();
    const unsigned long long t1 = wall_clock64();
    if (!lat_publish_and_arrive(s_list, p.c_partial + ((size_t)q * p.c_blocks + b) * np, np, p.done, gridDim.x * gridDim.y))
        return;
    const unsigned long long t2 = wall_clock64();
    const uint32_t total = p.c_blocks * np;
    if (total <= 32 * WAVE && p.reg_select)
    {
        __shared__ __attribute__((aligned(16))) uint32_t s_hist[BLOCK / WAVE][256];
        uint32_t * hist = p.reg_select == 3 ? nullptr : s_hist[tid >> 6];
        // wavefront w takes queries w, w + 4, ...
        for (uint32_t qq = tid >> 6; qq < p.nq; qq += BLOCK / WAVE)
        {
            const uint64_t * src = p.c_partial + (size_t)qq * total;
            int32_t * dst = p.probes + (size_t)qq * np;
            float * dd = p.probe_dis + (size_t)qq * np;
            if (total <= 4 * WAVE)
                lat_select_probes<4, SELM>(src, total, np, dst, dd, tid & 63, hist);
            else if (total <= 8 * WAVE)
                lat_select_probes<8, SELM>(src, total, np, dst, dd, tid & 63, hist);
            else if (total <= 16 * WAVE)
                lat_select_probes<16, SELM>(src, total, np, dst, dd, tid & 63, hist);
            else
                lat_select_probes<32, SELM>(src, total, np, dst, dd, tid & 63, hist);
        }
        __syncthreads();
    }
    else
        for (uint32_t qq = 0; qq < p.nq; qq++)
        {
            const uint64_t * merged = lat_merge_lists<true>(p.c_partial + (size_t)qq * p.c_blocks * np, p.c_blocks, p.c_blocks, np, lds_merge);
            if (tid < np)
            {
                p.probes[(size_t)qq * np + tid] = merged[tid] == KEY_NONE ? -1 : (int32_t)(uint32_t)merged[tid];
                p.probe_dis[(size_t)qq * np + tid] = key_value<METRIC>(merged[tid]);
            }
        }
    // radius pruning (LatParams::radius), once per search: lane i of wavefront qq weighs probe i of query qq; the probes it rules
    // out become -1 and get no work items in stage 2
    if (METRIC == M_L2 && p.radius && !p.alive)
    {
        __syncthreads(); // (the probes and their distances were written by this block: visible after the barrier)
        for (uint32_t qq = tid >> 6; qq < p.nq; qq += BLOCK / WAVE)
        {
            const uint32_t lane = tid & 63;
            const int32_t l = lane < np ? p.probes[(size_t)qq * np + lane] : -1;
            double ub = 1e300, lb = 0.0;
            if (l >= 0)
            {
                const double dc2 = (double)p.probe_dis[(size_t)qq * np + lane], r = (double)p.radius[l];
                if (dc2 == dc2 && dc2 >= 0.0 && dc2 < 1e30 && r == r && r < 1e18)
                {
                    // the centroid distance is canonical f32: widened by its rounding; |q| <= ||q - c|| + |c|
                    const double sc = sqrt((double)p.cmax * 1.001), sq = sqrt(dc2 * (1.0 + 4.0 * (p.c_canon + 4e-7))) * (1.0 + 1e-6) + sc,
                                 sx = sqrt((double)p.xmax * 1.001); // (dc2 is a canonical f32 value: widened by its own error before the root)
                    const double eps_c = (p.c_canon + 4e-7) * (sq + sc) * (sq + sc) + 1e-30;
                    const double slack = 2.0 * (p.c_canon + 4e-7) * (sq + sx) * (sq + sx) + 1e-30;
                    if ((uint64_t)(p.list_off[l + 1] - p.list_off[l]) >= p.k)
                    {
                        const double hi = sqrt(dc2 + eps_c) * (1.0 + 1e-7) + r;
                        ub = hi * hi * (1.0 + 1e-7) + slack; // every row of the list is within this: k of them bound the k-th best
                    }
                    if (dc2 - eps_c > 0.0)
                    {
                        const double lo = sqrt(dc2 - eps_c) * (1.0 - 1e-7);
                        if (lo > r)
                            lb = (lo - r) * (lo - r) * (1.0 - 1e-7);
                    }
                }
            }
            double U = ub;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1)
            {
                const double other = __shfl_xor(U, o);
                U = other < U ? other : U;
            }
            if (l >= 0 && lb > U) // strictly beyond the k-th best: no row of this list is a result (ties included)
                p.probes[(size_t)qq * np + lane] = -1;
        }
    }
    // stage 2's work partition, from the final probe lists (wavefront qq: query qq)
    __syncthreads();
    for (uint32_t qq = tid >> 6; qq < p.nq; qq += BLOCK / WAVE)
        lat_make_cut(p, qq, tid & 63);
    if (tid == 0)
        p.done[0] = 0;
    if (p.dbg && tid == 0)
    {
        p.dbg[0] = t0;
        p.dbg[1] = t1;
        p.dbg[2] = t2;
        p.dbg[3] = wall_clock64();
    }
}

/// grid (items + nprobe, nq); dynamic LDS: ld4 * 16 + max(5 * k * 8, lat_merge_lds(items + nprobe, k)) bytes
template <int METRIC>
__global__ __launch_bounds__(BLOCK) void lat_scan_kernel(const LatParams p)
{
    float4 * qs = reinterpret_cast<float4 *>(msvs_smem);
    uint64_t * lds_merge = reinterpret_cast<uint64_t *>(msvs_smem + (size_t)p.ld4 * 16);
    __shared__ int32_t s_probe[LAT_MAX_K];
    __shared__ uint32_t s_first[LAT_MAX_K + 1];
    __shared__ uint64_t s_list[LAT_MAX_K];
    const uint32_t tid = threadIdx.x, k = p.k, item = blockIdx.x, q = blockIdx.y;
    const unsigned long long t0 = wall_clock64();
    const LatCut cut = lat_cut(p, q, s_probe, s_first);
    const unsigned long long t1 = wall_clock64();
    const bool work = item < cut.total; // uniform
    if (work)
    {
        uint32_t pr = 0; // the list this item belongs to (nprobe <= 64 steps)
        while (s_first[pr + 1] <= item)
            pr++;
        const int32_t list = s_probe[pr];
        const int64_t lb = p.list_off[list] + (int64_t)(item - s_first[pr]) * cut.rpb;
        int64_t le = p.list_off[list + 1];
        if (le > lb + cut.rpb)
            le = lb + cut.rpb;
        ScanParams a{};
        a.Y = p.Y;
        a.ids = p.ids;
        a.alive = p.alive;
        a.nbits = p.nbits;
        a.Q = p.dq;
        a.ld4 = p.ld4;
        a.k = k;
        a.nq = p.nq;
        uint32_t qidx[1] = {q};
        uint64_t * out[1] = {s_list};
        stage_queries<1>(a, qidx, qs);
        scan_rows<METRIC, 1, 1>(a, (uint32_t)lb, (uint32_t)le, qs, lds_merge, out);
        __syncthreads();
    }
    const unsigned long long t2 = wall_clock64();
    if (!lat_publish_and_arrive(s_list, work ? p.partial + ((size_t)q * gridDim.x + item) * k : nullptr, k, p.done + 1,
                                gridDim.x * gridDim.y))
        return;
    const unsigned long long t3 = wall_clock64();
    for (uint32_t qq = 0; qq < p.nq; qq++)
    {
        const uint32_t lists = qq == q ? cut.total : lat_cut(p, qq, s_probe, s_first).total;
        const uint64_t * merged = lat_merge_lists<true>(p.partial + (size_t)qq * gridDim.x * k, lists, gridDim.x, k, lds_merge);
        if (tid < k)
        {
            const uint64_t key = merged[tid];
            const float v = key_value<METRIC>(key);
            // system scope: the destination may be pinned host memory a host thread reads as soon as the word flips
            __hip_atomic_store(p.out_ids + (size_t)qq * k + tid, key == KEY_NONE ? (int64_t)-1 : (int64_t)(uint32_t)key, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(p.out_dis + (size_t)qq * k + tid, p.cosine ? __fsub_rn(1.0f, v) : v, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (tid == 0)
        p.done[1] = 0;
    if (p.dbg && tid == 0)
    {
        p.dbg[4] = t0;
        p.dbg[5] = t1;
        p.dbg[6] = t2;
        p.dbg[7] = t3;
        p.dbg[8] = wall_clock64();
    }
    if (p.flag)
    {
        __builtin_amdgcn_s_waitcnt(0); // results acknowledged before the word the host polls
        __syncthreads();
        if (tid == 0) // release at system scope: the result stores of the whole block (ordered before this by the barrier) are
                      // visible to the host thread that acquires the word
            __hip_atomic_store(p.flag, p.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

/// The canonical coarse quantiser of a SMALL batch in one launch (search_entry.hip: coarse_few_launch; step 1 of the general search for
/// up to LAT_COARSE_MAX_Q queries): grid (c_blocks, ceil(nq / T)).  Block (b, g) scans c_rows centroids for the T queries of group g
/// (scan_rows: the canonical arithmetic, one pass over the rows for the T of them -- one query per block re-reads the 3 MB table per
/// query: 36 us of L2 -> CU streaming at 64 queries), publishes their T lists and arrives at the GROUP's counter; the last block of a
/// group selects its queries' probes, one wavefront per query (lat_select_probes: exact top-nprobe in the oracle's order of keys; the
/// probes leave in arbitrary order).  The batched kernels spend a scan and a merge launch on this: 22 + 14 us at 32 queries.
/// dynamic LDS: T * ld4 * 16 + T * 5 * nprobe * 8 bytes; c_blocks * nprobe <= 32 * WAVE.
template <int METRIC, int T>
__global__ __launch_bounds__(BLOCK) void coarse_few_kernel(const LatParams p)
{
    static_assert(T <= BLOCK / WAVE, "one wavefront per query in the selection");
    constexpr int SELM = METRIC; // (probe distances as key values)
    float4 * qs = reinterpret_cast<float4 *>(msvs_smem);
    uint64_t * lds_merge = reinterpret_cast<uint64_t *>(msvs_smem + (size_t)T * p.ld4 * 16);
    const uint32_t tid = threadIdx.x, g = blockIdx.y, b = blockIdx.x, np = p.nprobe, q0 = g * T;
    ScanParams a{};
    a.Y = p.C;
    a.Q = p.Q;
    a.ld4 = p.ld4;
    a.k = np;
    a.nq = p.nq;
    __shared__ uint64_t s_list[T][LAT_MAX_K];
    __shared__ uint32_t s_last;
    uint32_t qidx[T];
    uint64_t * out[T];
#pragma unroll
    for (int t = 0; t < T; t++)
    {
        qidx[t] = q0 + t < p.nq ? q0 + t : p.nq - 1;
        out[t] = s_list[t];
    }
    stage_queries<T>(a, qidx, qs);
    const uint32_t rb = b * p.c_rows, re = rb + p.c_rows < p.nlist ? rb + p.c_rows : p.nlist;
    scan_rows<METRIC, T, 1>(a, rb, re, qs, lds_merge, out);
    __syncthreads();
    // the T lists go out with agent-scope stores that are waited for before the arrival (lat_publish_and_arrive)
#pragma unroll
    for (int t = 0; t < T; t++)
        if (q0 + t < p.nq && tid < np)
            __hip_atomic_store(p.c_partial + ((size_t)(q0 + t) * p.c_blocks + b) * np + tid, s_list[t][tid], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0)
        s_last = atomicAdd(p.done_q + g, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!s_last)
        return;
    const uint32_t total = p.c_blocks * np, w = tid >> 6, q = q0 + w;
    __shared__ __attribute__((aligned(16))) uint32_t s_hist[BLOCK / WAVE][256];
    uint32_t * hist = p.reg_select == 3 ? nullptr : s_hist[w];
    if (w < (uint32_t)T && q < p.nq)
    {
        const uint64_t * src = p.c_partial + (size_t)q * total;
        int32_t * dst = p.probes + (size_t)q * np;
        float * dd = p.probe_dis + (size_t)q * np;
        if (total <= 4 * WAVE)
            lat_select_probes<4, SELM>(src, total, np, dst, dd, tid & 63, hist);
        else if (total <= 8 * WAVE)
            lat_select_probes<8, SELM>(src, total, np, dst, dd, tid & 63, hist);
        else if (total <= 16 * WAVE)
            lat_select_probes<16, SELM>(src, total, np, dst, dd, tid & 63, hist);
        else
            lat_select_probes<32, SELM>(src, total, np, dst, dd, tid & 63, hist);
    }
    if (tid == 0)
        p.done_q[g] = 0;
}

/// coarse_few_kernel without the per-block top-k, for tables of at most 32 * WAVE centroids (the selection's registers hold every
/// key of a query): a block's nprobe-best list is as long as its slice of the table as soon as nprobe >= c_rows (32 of 32 on the bench
/// index), so the blocks write the canonical key of EVERY (query, centroid) -- 16 lanes per centroid, the arithmetic and order of
/// scan_rows / the re-rank (canonical_update over the lane's columns in ascending order, the component sums, row16_tree_sum) -- and
/// the group's last block selects out of the nlist keys.  No top-k registers, no LDS merge: the launch is the row loads and the
/// selection.  grid (ceil(nlist / c_rows), ceil(nq / T)); c_partial: [nq][nlist]; dynamic LDS: T * ld4 * 16 bytes.
template <int METRIC, int T>
__global__ __launch_bounds__(BLOCK) void coarse_dense_kernel(const LatParams p)
{
    static_assert(T <= BLOCK / WAVE, "one wavefront per query in the selection");
    float4 * qs = reinterpret_cast<float4 *>(msvs_smem);
    const uint32_t tid = threadIdx.x, g = tid & 15, grp = tid >> 4, gq = blockIdx.y, b = blockIdx.x, np = p.nprobe, q0 = gq * T;
    const uint32_t ld4 = p.ld4, jfull = ld4 >> 4, jtail = ld4 & 15;
    __shared__ uint32_t s_last;
#pragma unroll
    for (int t = 0; t < T; t++)
    {
        const float4 * src = p.Q + (size_t)(q0 + t < p.nq ? q0 + t : p.nq - 1) * ld4;
        for (uint32_t c = tid; c < ld4; c += BLOCK)
            qs[t * ld4 + c] = src[c];
    }
    __syncthreads();
    const uint32_t rb = b * p.c_rows, re = rb + p.c_rows < p.nlist ? rb + p.c_rows : p.nlist;
    for (uint32_t base = rb; base < re; base += BLOCK / 16)
    {
        const uint32_t r = base + grp;
        const bool rv = r < re;
        const float4 * yrow = p.C + (size_t)(rv ? r : re - 1) * ld4 + g;
        const float4 * qrow = qs + g;
        float4 acc[T];
#pragma unroll
        for (int t = 0; t < T; t++)
            acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        uint32_t j = 0;
        for (; j + 6 <= jfull; j += 6) // six 16-byte loads in flight per lane, then the queries over them
        {
            float4 y[6];
#pragma unroll
            for (int u = 0; u < 6; u++)
                y[u] = yrow[(j + u) * 16];
#pragma unroll
            for (int t = 0; t < T; t++)
#pragma unroll
                for (int u = 0; u < 6; u++)
                    canonical_update<METRIC>(acc[t], qrow[t * ld4 + (j + u) * 16], y[u]);
        }
        for (; j < jfull; j++)
        {
            const float4 y1 = yrow[j * 16];
#pragma unroll
            for (int t = 0; t < T; t++)
                canonical_update<METRIC>(acc[t], qrow[t * ld4 + j * 16], y1);
        }
        if (g < jtail)
        {
            const float4 y1 = yrow[jfull * 16];
#pragma unroll
            for (int t = 0; t < T; t++)
                canonical_update<METRIC>(acc[t], qrow[t * ld4 + jfull * 16], y1);
        }
#pragma unroll
        for (int t = 0; t < T; t++)
        {
            float s = __fadd_rn(__fadd_rn(acc[t].x, acc[t].y), __fadd_rn(acc[t].z, acc[t].w));
            s = row16_tree_sum(s);
            if (rv && g == 0 && q0 + t < p.nq) // (agent scope: the last block of the group may run on another XCD)
                __hip_atomic_store(p.c_partial + (size_t)(q0 + t) * p.nlist + r, make_key<METRIC>(s, r), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0)
        s_last = atomicAdd(p.done_q + gq, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!s_last)
        return;
    const uint32_t total = p.nlist, w = tid >> 6, q = q0 + w;
    __shared__ __attribute__((aligned(16))) uint32_t s_hist[BLOCK / WAVE][256];
    uint32_t * hist = p.reg_select == 3 ? nullptr : s_hist[w];
    if (w < (uint32_t)T && q < p.nq)
    {
        const uint64_t * src = p.c_partial + (size_t)q * total;
        int32_t * dst = p.probes + (size_t)q * np;
        float * dd = p.probe_dis + (size_t)q * np;
        if (total <= 4 * WAVE)
            lat_select_probes<4, METRIC>(src, total, np, dst, dd, tid & 63, hist);
        else if (total <= 8 * WAVE)
            lat_select_probes<8, METRIC>(src, total, np, dst, dd, tid & 63, hist);
        else if (total <= 16 * WAVE)
            lat_select_probes<16, METRIC>(src, total, np, dst, dd, tid & 63, hist);
        else
            lat_select_probes<32, METRIC>(src, total, np, dst, dd, tid & 63, hist);
    }
    if (tid == 0)
        p.done_q[gq] = 0;
}

}
