/*
 * msvs.h -- C-ABI of libmsvs.so: the MI355X (gfx950) vector-scan / BM25 hot path for MyScaleDB.
 *
 * Plain pointers and sizes only.  Every entry point replaces one call the MyScaleDB host makes
 * into its (absent) native libraries; the reference call site is cited at each declaration
 * (paths relative to the MyScaleDB tree).  INTEGRATION.md shows the host-side shims.
 *
 * Conventions kept from the reference:
 *   - ids are int64 row offsets local to the data part, -1 = "no result" (host tests `> -1`,
 *     src/VectorIndex/Storages/MergeTreeVSManager.cpp:1507,1523); on the device they must fit u32
 *     (the host stores labels in ColumnUInt32, MergeTreeVSManager.cpp:418-420);
 *   - results are sorted best-first; L2 is the SQUARED distance; cosine is 1 - <x^,y^>;
 *     IP is the raw dot product, larger is better;
 *   - unfilled slots carry the heap's neutral value: +FLT_MAX (L2), -FLT_MAX (IP),
 *     1 - (-FLT_MAX) (cosine, like VIWithDataPart.h:374-380);
 *   - ties are broken by ascending id (the reference leaves this to Faiss' heap);
 *   - filter bitmaps are LSB-first uint64 words, bit i of word i/64 <=> id i, 1 = candidate
 *     (the shim converts Search::DenseBitmap through is_member()/to_vector()).
 *
 * Threading: every function may be called concurrently from many host threads (the reference
 * allows 2 x cores concurrent searches, ScanThreadLimiter.h); index search is re-entrant on a
 * shared immutable index.  Errors: return value != MSVS_OK and a thread-local message in
 * msvs_last_error() -- the shim rethrows it as VIException (VICommon.h:75-104) so the host's
 * existing fallback logic applies.  There is NO CPU fallback inside this library.
 */
#ifndef MSVS_H
#define MSVS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSVS_API __attribute__((visibility("default")))

typedef enum msvs_status {
    MSVS_OK = 0,
    MSVS_ERR_INVALID_ARGUMENT = 1,
    MSVS_ERR_NOT_IMPLEMENTED = 2, /* DB::ErrorCodes::NOT_IMPLEMENTED, BruteForceSearch.h:89-92 */
    MSVS_ERR_DEVICE = 3,          /* HIP runtime error; message carries hipGetErrorString */
    MSVS_ERR_OUT_OF_MEMORY = 4,
    MSVS_ERR_NOT_READY = 5, /* index->ready() == false, VIWithDataPart.cpp:876-879 */
    MSVS_ERR_UNSUPPORTED_K = 6,
    MSVS_ERR_ID_RANGE = 7,
    MSVS_ERR_IO = 8,
    MSVS_ERR_ABORTED = 236 /* = DB::ErrorCodes::ABORTED (src/Common/ErrorCodes.cpp: M(236, ABORTED)): the build was cancelled
                              through msvs_index_set_cancel -- what the host itself throws at VIPartReader.h:175-176 */
} msvs_status;

enum msvs_metric { MSVS_METRIC_L2 = 0, MSVS_METRIC_IP = 1, MSVS_METRIC_COSINE = 2, MSVS_METRIC_HAMMING = 3, MSVS_METRIC_JACCARD = 4 };
enum msvs_index_type { MSVS_INDEX_FLAT = 0, MSVS_INDEX_IVFFLAT = 1 };
enum msvs_mem { MSVS_MEM_HOST = 0, MSVS_MEM_DEVICE = 1 };

/* largest k (and nprobe) one device top-k pass supports (what msvs_index_search_device accepts) */
#define MSVS_MAX_K 256
/* largest k of the host-pointer entry points: beyond MSVS_MAX_K they run exact rounds of MSVS_MAX_K per query,
 * each excluding the rows already returned (covers the reference's k + deleted-rows over-fetch and LIMIT 1000) */
#define MSVS_MAX_K_ROUNDS 4096

MSVS_API const char * msvs_last_error(void);
MSVS_API const char * msvs_version(void);
MSVS_API int msvs_device_count(int * count);
MSVS_API int msvs_set_device(int ordinal);
MSVS_API int msvs_device_synchronize(void);

/* ---------------------------------------------------------------------------------------------
 * Seam A2 -- brute force.  Replaces the body of
 *   VectorIndex::tryBruteForceSearch<FloatVector>(x, y, d, k, nx, ny, result_id, distance, metric)
 *   (src/VectorIndex/Common/BruteForceSearch.h:63-92) = faiss::knn_L2sqr / faiss::knn_inner_product.
 * x: nx*d queries, y: ny*d base rows (row-major f32, HOST pointers); ids/dis: nx*k, caller-owned.
 * metric: MSVS_METRIC_L2 or MSVS_METRIC_IP; anything else -> MSVS_ERR_NOT_IMPLEMENTED like the
 * reference (cosine is composed above this call by searchWithoutIndex, VIWithDataPart.h:341-382).
 */
MSVS_API int msvs_knn_f32(const float * x, const float * y, size_t d, size_t k, size_t nx, size_t ny, int metric,
                          int64_t * ids, float * dis);
/* Same with a row filter (LSB-first bitmap over the ny base rows, 1 = candidate): lets the host's searchWrapper
 * (src/VectorIndex/Storages/MergeTreeVSManager.cpp:1612-1633) drop lightweight-deleted rows inside the scan instead of
 * over-fetching k + delete_id_num and filtering afterwards -- the two are equivalent (top-k of the alive rows). */
MSVS_API int msvs_knn_f32_filtered(const float * x, const float * y, size_t d, size_t k, size_t nx, size_t ny,
                                   int metric, const uint64_t * alive_bits, int64_t * ids, float * dis);

/* VectorDataset<FloatVector>::normalize() (src/VectorIndex/Common/VectorDataset.h:98-117) on the device:
 * sequential f32 sum of squares, rows with sum < FLT_EPSILON untouched, x /= sqrt(sum). In place, HOST pointer. */
MSVS_API int msvs_normalize_f32(float * x, size_t n, size_t d);
/* Binary vectors (FixedString(N) columns, nbytes = N = dimension / 8): replaces
 *   faiss::hammings_knn_mc(x, y, nx, ny, k, d / 8, (int32_t *)distance, result_id, nullptr)
 *   jaccard_knn(x, y, nx, ny, k, d / 8, distance, result_id, nullptr)
 * behind tryBruteForceSearch<BinaryVector> (src/VectorIndex/Common/BruteForceSearch.h:94-110).
 * metric: MSVS_METRIC_HAMMING (popcount(x ^ y), reported as a float like the reference's Float32 distance column) or
 * MSVS_METRIC_JACCARD ((|x | y| - |x & y|) / |x | y|, 1 for two zero vectors); anything else -> MSVS_ERR_NOT_IMPLEMENTED.
 * alive_bits: nullable LSB-first bitmap over the ny rows.  Results ascending by (distance, row), unfilled slots id -1
 * with distance FLT_MAX.  k <= MSVS_MAX_K. */
MSVS_API int msvs_knn_bin(const uint8_t * x, const uint8_t * y, size_t nbytes, size_t k, size_t nx, size_t ny,
                          int metric, const uint64_t * alive_bits, int64_t * ids, float * dis);

/* Seam A1 for BinaryVector -- Search::VectorIndex<IS, OS, Bitmap, DataType::BinaryVector> (VICommon.h:142-143; the host creates
 * it at VIWithDataPart.cpp:431-446 and searches it at :928-935; index types BinaryFLAT / BinaryMSTG, goldens 00038): an
 * exhaustive Hamming / Jaccard scan (msvs_knn_bin's arithmetic) over rows kept RESIDENT on the device.  rows: n * nbytes
 * (a FixedString(N) column, 8 bits per byte); ids: the rows' labels (part row offsets; NULL = staging order); alive_bits:
 * nullable LSB-first bitmap over LABELS (nbits of them).  Files "data_bin" / "id_list" through the caller's stream openers like
 * the float index.  Results ascending by (distance, label), unfilled slots id -1 / FLT_MAX. */
typedef struct msvs_bin_index msvs_bin_index_t;
MSVS_API int msvs_bin_index_create(size_t nbytes, int metric, msvs_bin_index_t ** out);
MSVS_API void msvs_bin_index_free(msvs_bin_index_t * index);
MSVS_API int msvs_bin_index_add(msvs_bin_index_t * index, const uint8_t * rows, const int64_t * ids, size_t n);
MSVS_API size_t msvs_bin_index_num_data(const msvs_bin_index_t * index);
MSVS_API int msvs_bin_index_search(const msvs_bin_index_t * index, const uint8_t * x, size_t nx, size_t k, const uint64_t * alive_bits,
                                   size_t nbits, int64_t * ids, float * dis);
struct msvs_io;
MSVS_API int msvs_bin_index_serialize_io(const msvs_bin_index_t * index, const struct msvs_io * io);
MSVS_API int msvs_bin_index_load_io(const struct msvs_io * io, msvs_bin_index_t ** out);

/* Resident blocks for the brute-force path: the GPU analogue of VICacheManager / VIWithMeta
 * (src/VectorIndex/Cache/VICacheObject.h:40-162) for the dense block a mark of a part turns into
 * (MergeTreeVSManager.cpp:1380-1392).  msvs_knn_f32 moves that block over PCIe on every query; here it is uploaded
 * once, keyed by (part key, mark) -- CacheKey::toString() is a natural part key -- and kept in HBM in an LRU bounded by
 * bytes.  Blocks are immutable: lightweight deletes arrive per search as the row_exists bitmap like in the reference
 * (searchWrapper, MergeTreeVSManager.cpp:1612-1633); a dropped / mutated part is evicted by key prefix.
 * upload: returns the resident block PINNED (an identical key already resident is returned instead -- concurrent
 * part-pool threads race benignly); normalize != 0 stores VectorDataset::normalize()'d rows (the cosine composition of
 * searchWithoutIndex normalises the block in place before the IP search).  lookup: *out = NULL when absent.
 * release unpins; a pinned block is never freed, an evicted one disappears at its last release.  Thread-safe. */
typedef struct msvs_cache msvs_cache_t;
typedef struct msvs_block msvs_block_t;
MSVS_API int msvs_cache_create(size_t capacity_bytes, msvs_cache_t ** out);
MSVS_API void msvs_cache_free(msvs_cache_t * cache);
MSVS_API int msvs_block_upload(msvs_cache_t * cache, const char * part_key, uint64_t mark, const float * rows, size_t n,
                               size_t d, int normalize, msvs_block_t ** out);
MSVS_API int msvs_block_lookup(msvs_cache_t * cache, const char * part_key, uint64_t mark, msvs_block_t ** out);
MSVS_API void msvs_block_release(msvs_block_t * block);
/* Shape and stored form of a resident block (rows, dimension, 1 = rows were normalised at upload); any out pointer may be NULL.
 * A caller that finds a block by key checks these against what it is about to search: a block uploaded for Cosine holds
 * normalised rows and must not serve an L2 / IP search of the same part (msvs_host.cpp keys the two forms apart as well). */
MSVS_API int msvs_block_info(const msvs_block_t * block, size_t * n, size_t * d, int * normalized);
MSVS_API int msvs_cache_evict(msvs_cache_t * cache, const char * key_prefix, size_t * evicted);
MSVS_API int msvs_cache_stats(msvs_cache_t * cache, size_t * bytes, size_t * blocks, uint64_t * hits, uint64_t * misses,
                              uint64_t * evictions);
/* msvs_knn_f32[_filtered] against a resident block: x nx*d (HOST), alive_bits nullable over the block's rows. */
MSVS_API int msvs_knn_resident(const msvs_block_t * block, const float * x, size_t k, size_t nx, int metric,
                               const uint64_t * alive_bits, int64_t * ids, float * dis);

/* ---------------------------------------------------------------------------------------------
 * Seam A1 -- vector index object.  Replaces Search::VectorIndex<...,FloatVector>:
 *   createVectorIndex(name, type, metric, dim, total_vec, params, ...)   VIWithDataPart.cpp:415-446
 *   build(reader, threads, cancel)  [train + chunked add]                VIWithDataPart.h:295-339, VIPartReader.h:170-304
 *   search(DataSet{nq,dim}, k, params, first_stage_only, DenseBitmap*)   VIWithDataPart.cpp:922-926
 *   ready(), numData(), getResourceUsage(), serialize()/load()           VIWithDataPart.cpp:368-385,472-479,698-700
 * The index data lives in HBM; search never touches host copies of the vectors.
 */
typedef struct msvs_index msvs_index_t;

/* params: "key=value" pairs separated by ',' or a flat JSON object, e.g. "ncentroids=1024" /
 * {"ncentroids":"1024"}.  IVFFLAT build params: ncentroids (default 1024), kmeans_iters (10),
 * train_sample (ncentroids*64), seed (1234).  Sharding for multi-GPU: shard_rank / shard_world
 * (lists with list_id % shard_world == shard_rank are kept on this device). */
MSVS_API int msvs_index_create(int index_type, int metric, size_t dim, const char * params, msvs_index_t ** out);
MSVS_API void msvs_index_free(msvs_index_t * index);

/* IVFFLAT: k-means over the training rows (ignored by FLAT).  mem = MSVS_MEM_HOST|MSVS_MEM_DEVICE. */
MSVS_API int msvs_index_train(msvs_index_t * index, const float * x, size_t n, int mem);
/* Alternative to train: adopt given coarse centroids (nlist*dim). */
MSVS_API int msvs_index_set_centroids(msvs_index_t * index, const float * centroids, size_t nlist, int mem);
/* One chunk of the build feed: n rows (n*dim f32) + their ids (nullable => consecutive from the current count).
 * Mirrors Search::DataChunk{data, n, dim} + setDataID(ids) (VIPartReader.h:296-303). */
MSVS_API int msvs_index_add(msvs_index_t * index, const float * x, const int64_t * ids, size_t n, int mem);
/* The factory's check_cancelled callback (VIWithDataPart.cpp:425-430: BaseDaemon::isCancelled): polled between k-means
 * iterations of train, at every add and between the phases of build; nonzero -> the call returns MSVS_ERR_ABORTED
 * ("Cancelled building vector index") and the index stays unbuilt.  NULL clears it. */
MSVS_API int msvs_index_set_cancel(msvs_index_t * index, int (*is_cancelled)(void * ctx), void * ctx);
/* Finalise: assign rows to lists, lay lists out contiguously (ascending id inside a list), drop staging. */
MSVS_API int msvs_index_build(msvs_index_t * index);
MSVS_API int msvs_index_ready(const msvs_index_t * index);
MSVS_API size_t msvs_index_num_data(const msvs_index_t * index);
MSVS_API size_t msvs_index_num_lists(const msvs_index_t * index);
MSVS_API size_t msvs_index_memory_usage(const msvs_index_t * index);

/* search(): queries nq*dim f32 (HOST); params e.g. "nprobe=32" (IVFFLAT search param, validated like
 * parseVSParameters.cpp:43-222); alive_bits nullable, nbits = number of valid bits; ids/dis nq*k (HOST). */
MSVS_API int msvs_index_search(const msvs_index_t * index, const float * queries, size_t nq, int k,
                               const char * params, const uint64_t * alive_bits, size_t nbits, int64_t * ids,
                               float * dis);
/* Same with DEVICE pointers, enqueued on `hip_stream` (hipStream_t, NULL = default stream) without host
 * synchronisation: the form a GPU-resident host pipeline (and bench.py) uses. */
MSVS_API int msvs_index_search_device(const msvs_index_t * index, const float * d_queries, size_t nq, int k,
                                      int nprobe, const uint64_t * d_alive_bits, size_t nbits, int64_t * d_ids,
                                      float * d_dis, void * hip_stream);

/* VIWithMeta state of a cached index (src/VectorIndex/Cache/VICacheObject.h:40-117), resident in HBM:
 *  - set_delete_bitmap: the lightweight-delete bitmap (1 = alive, over the index labels), swapped atomically like
 *    VIWithMeta::setDeleteBitmap; every search ANDs it into its filter on the device (VIWithDataPart.cpp:903-908) instead
 *    of the host intersecting and re-uploading bitmaps per query.  NULL clears it.
 *  - set_merged_maps: the row-id maps of a DECOUPLED part (a merged part still served by its source parts' indexes,
 *    SegmentId::getMergedMaps): row_ids_map[label] = row of the merged part; inverted_row_ids_map /
 *    inverted_row_sources_map [merged row] = (label, source part).  Once set, a search takes its filter in the merged
 *    part's row space (getRealBitmap, src/VectorIndex/Utils/VIUtils.cpp:479-498) and reports merged-part rows
 *    (transferToNewRowIds, VIWithDataPart.cpp:56-67). */
MSVS_API int msvs_index_set_delete_bitmap(msvs_index_t * index, const uint64_t * alive_bits, size_t nbits);
MSVS_API int msvs_index_set_merged_maps(msvs_index_t * index, const uint64_t * row_ids_map, size_t n_old,
                                        const uint64_t * inverted_row_ids_map, const uint8_t * inverted_row_sources_map,
                                        size_t n_new, uint32_t own_id);

/* Export the index structure (for parity checks against the oracle and for serialisation):
 * any output may be NULL; sizes: centroids nlist*dim, list_off nlist+1, vecs num_data*dim, ids num_data.
 * Cosine indexes export the stored (normalised) rows. */
MSVS_API int msvs_index_export(const msvs_index_t * index, float * centroids, int64_t * list_off, float * vecs,
                               int64_t * ids);

/* One inverted list of a built IVFFLAT index: its rows (len * dim f32, as stored: cosine indexes hold normalised rows) and
 * labels (len i64), len = list_off[list + 1] - list_off[list] of msvs_index_export.  Lets a checker redo what a few queries
 * touch on an index too large to export whole (12.5M x 1536 = 77 GB).  Either output may be NULL. */
MSVS_API int msvs_index_export_list(const msvs_index_t * index, size_t list, float * vecs, int64_t * ids);

/* Balance of the inverted lists of a built IVFFLAT index: list count, shortest / longest list, and Faiss' imbalance factor
 * nlist * sum(len^2) / n^2 (1 = uniform; the expected number of rows a probe meets is imbalance * n / nlist); train_empty =
 * empty clusters the last k-means iteration had to re-seed (0 for loaded indexes / caller-given centroids).  Any out may be NULL. */
MSVS_API int msvs_index_list_stats(const msvs_index_t * index, size_t * nlist, size_t * min_len, size_t * max_len, double * imbalance,
                                   size_t * train_empty);

/* Serialisation through caller-supplied streams -- the form the reference's library uses: Search::VectorIndex::serialize
 * (IndexDataFileWriter<OS>*) / saveDataID / load(IndexDataFileReader<IS>*) / loadDataID open every file of the index
 * through the host's opener, a VectorIndexWriter / VectorIndexReader over IDisk (local disk or S3 alike):
 * src/VectorIndex/Common/VectorIndexIO.h:25-166, VIWithDataPart.cpp:461-473 and :688-700.
 * The index is a set of NAMED files ("data_bin": header, centroids, list offsets, rows; "id_list": the row ids); the
 * host shim maps NAME to <index_name>-NAME.vidx3 inside the part directory (VICommon.h:55, SegmentId).
 *   open(ctx, name, write) -> stream handle or NULL;  write / read -> bytes moved (a short count is an error / EOF);
 *   close -> 0 on success.  All four are called from the thread that called serialize / load.
 * load validates everything it reads (MSVS_ERR_IO on a corrupt or truncated file; nothing is searched before). */
typedef struct msvs_io
{
    void * ctx;
    void * (*open)(void * ctx, const char * name, int write);
    int64_t (*write)(void * ctx, void * stream, const void * buf, size_t n);
    int64_t (*read)(void * ctx, void * stream, void * buf, size_t n);
    int (*close)(void * ctx, void * stream);
} msvs_io_t;
MSVS_API int msvs_index_serialize_io(const msvs_index_t * index, const msvs_io_t * io);
MSVS_API int msvs_index_load_io(const msvs_io_t * io, msvs_index_t ** out);
/* Convenience over stdio: the file set <path_prefix>-data_bin.vidx3 and <path_prefix>-id_list.vidx3. */
MSVS_API int msvs_index_serialize(const msvs_index_t * index, const char * path_prefix);
MSVS_API int msvs_index_load(const char * path_prefix, msvs_index_t ** out);
/* getVersion().toString() and getResourceUsage() of the reference's index object (VIWithDataPart.cpp:368-385,
 * :485-488): the host records them in <index>-vector_index_description.vidx3 (VIMetadata.cpp:115-187). */
MSVS_API const char * msvs_index_version(void);
MSVS_API int msvs_index_resource_usage(const msvs_index_t * index, size_t * memory_usage_bytes,
                                       size_t * disk_usage_bytes, size_t * build_memory_usage_bytes);

/* Measurement support (bench.py): rows the list scan of msvs_index_search(queries, nprobe) has to look at.
 *   *rows          = sum over (query, probed list) of the list length (the per-query model of SURVEY.md 8d);
 *   *rows_streamed = rows the launch actually streams from HBM given its work decomposition: with query tiles
 *                    of T queries per list pass this is sum over lists of ceil(pairs_on_list / T) * list length
 *                    (== *rows when T == 1).  FLAT: nq * num_data / ceil(nq / T) * num_data;
 *   *rows_unique   = rows probed by at least one query of the batch (the union): what MUST come from HBM at least
 *                    once for this batch, i.e. the batch-level algorithmic bytes / (4d + 4).  Queries are HOST. */
MSVS_API int msvs_index_scanned_rows(const msvs_index_t * index, const float * queries, size_t nq, int nprobe,
                                     uint64_t * rows, uint64_t * rows_streamed, uint64_t * rows_unique);
/* Kernel timing with HIP events recorded on the launch stream around every kernel of the scan path.
 * enable(1) starts collecting, get() synchronises and returns call count and summed milliseconds of the kernel
 * family `name` ("flat_scan", "ivf_scan", "merge", "bm25_score"), reset() drops the samples. */
MSVS_API int msvs_profile_enable(int on);
MSVS_API int msvs_profile_get(const char * name, uint64_t * calls, double * total_ms);
MSVS_API int msvs_profile_reset(void);
/* Counters of the matrix-core candidate pass of batched IVFFLAT searches (many queries per list): `queries` that
 * went through it on the current device since process start and how many of them had no exactness certificate and
 * were re-run through the canonical scan (`fallbacks`).  Results are identical either way; this is a speed metric.
 * Synchronises the device. */
MSVS_API int msvs_prefilter_stats(uint64_t * queries, uint64_t * fallbacks);
/* ... and of the coarse quantiser's candidate passes (whose output is the probe lists). */
MSVS_API int msvs_coarse_stats(uint64_t * queries, uint64_t * fallbacks);
/* msvs_index_search from many host threads (the reference: one query per thread, up to ScanThreadLimiter of them,
 * MergeTreeVSManager.cpp:973): up to 8 unfiltered calls of <= 4 queries run directly; callers beyond that queue, and the next call
 * to finish hands its slot to the first waiter together with every waiter that asks for the same k and parameters -- one batched
 * search serves them all (same bits: every path is exact).  Counters: calls through this front end, batches of several callers,
 * queries served by such batches.  Option "combine" = the number of direct calls (0: off). */
MSVS_API int msvs_combine_stats(uint64_t * calls, uint64_t * batches, uint64_t * batched_queries);
/* Scratch memory is kept per (host thread, stream) and reused call after call; it shrinks by itself when a thread's
 * requests stay small for a window of 64 calls.  msvs_release_scratch frees the calling thread's arenas now (device synchronised). */
MSVS_API int msvs_release_scratch(size_t * freed_bytes);
/* Experiment / test knobs (DESIGN.md section 6b lists them; none is needed in production).  They are read ONCE per
 * process from the MSVS_<NAME> environment variables; afterwards only this call changes them -- a search never calls
 * getenv.  `name` without the MSVS_ prefix, any case; value NULL or "" restores the default.
 * MSVS_ERR_INVALID_ARGUMENT for an unknown name. */
MSVS_API int msvs_set_option(const char * name, const char * value);

/* Multi-GPU: one process per GPU, every process holds ONE shard of the index (msvs_index_create params
 * shard_rank / shard_world: IVF lists list_id % world, FLAT row ranges; centroids replicated) and a communicator.
 * msvs_shard_search_device is the whole sharded search as ONE stream-ordered call: the coarse quantiser sharded by
 * query + one all-gather of the probe lists, the scan of the local lists, one all-gather of the packed partial top-k
 * (nq * k * 12 B per rank over xGMI) and the canonical W-way merge -- the device-side analogue of the per-part searches
 * + getTotalTopSearchResultImpl (MergeTreeBaseSearchManager.cpp:207-299).  Every rank calls it with the SAME queries
 * and gets the SAME result (ids bit-exact with the unsharded index).
 * The communicator is RCCL (librccl resolved at first use, straight rccl.h calls): rank 0 creates the id
 * (msvs_comm_unique_id), the host distributes its MSVS_COMM_ID_BYTES bytes by any means it has (ClickHouse: its own
 * TCP / Keeper), every rank calls msvs_comm_init on its device.  msvs_comm_init_custom plugs a caller-supplied
 * all-gather instead (tests run two ranks on one GPU over gloo): it must gather `bytes` from d_send of every rank into
 * d_recv[rank * bytes ...] of every rank, ordered after the work already enqueued on `hip_stream`, and return 0. */
#define MSVS_COMM_ID_BYTES 128
typedef struct msvs_comm msvs_comm_t;
typedef int (*msvs_allgather_fn)(void * ctx, const void * d_send, void * d_recv, size_t bytes, void * hip_stream);
MSVS_API int msvs_comm_unique_id(void * id_out /* MSVS_COMM_ID_BYTES */);
MSVS_API int msvs_comm_init(const void * id, int nranks, int rank, msvs_comm_t ** out);
MSVS_API int msvs_comm_init_custom(int nranks, int rank, msvs_allgather_fn all_gather, void * ctx, msvs_comm_t ** out);
MSVS_API void msvs_comm_free(msvs_comm_t * comm);
/* Sum of n u64 counters over the ranks, in place in HOST memory, every rank gets the sums -- the one exchange step of a sharded
 * BM25 search: (total documents, total tokens per column, document frequency per query term), what BM25InfoInDataParts /
 * the ftsIndex table function add up over parts and shards (BM25InfoInDataParts.cpp:40-93, StorageFtsIndex.cpp:150-213).
 * Runs on the communicator's all-gather (RCCL or the caller's), ordered on hip_stream, and returns when the sums are there. */
MSVS_API int msvs_comm_all_reduce_u64(const msvs_comm_t * comm, uint64_t * values, size_t n, void * hip_stream);
MSVS_API int msvs_comm_rank(const msvs_comm_t * comm);
MSVS_API int msvs_comm_size(const msvs_comm_t * comm);
MSVS_API int msvs_shard_search_device(const msvs_index_t * shard, const msvs_comm_t * comm, const float * d_queries,
                                      size_t nq, int k, int nprobe, const uint64_t * d_alive_bits, size_t nbits,
                                      int64_t * d_ids, float * d_dis, void * hip_stream);
/* The ROUTED form: every rank brings its OWN batch (nq may differ per rank, 0 included) and gets the results of its own queries -- the
 * queries that arrived at this server, StorageDistributed.cpp:1213-1255 -- instead of every rank working through the same batch.
 * FRONT phase: a query's coarse quantiser and the pre-pruning by the list radius (over the lists of the whole index) run on its home
 * rank; one all-gather of the rank x rank count matrix (+ a header per rank: status, filter / pruning flags, shard instance, k, nprobe)
 * is read by every host.  BACK phase: the query visits only the ranks that own lists it still needs (ncclSend / ncclRecv of the exact
 * sizes, grouped), each of which searches it under ITS OWN filter -- d_alive_bits (nullable; 1 = row may be returned, over the row ids
 * this rank holds, like msvs_index_search_device) AND the shard's resident delete bitmap (VIWithDataPart.cpp:903-908) -- and returns
 * the exact top-k over its lists; the home rank merges (MergeTreeBaseSearchManager.cpp:207-299) and applies its row-id map.  ids and
 * distances == the unsharded index's under the union of the filters, bit for bit.  IVFFLAT shards, <= 32 ranks.
 * COLLECTIVE: every rank of the communicator calls it for every step with the same k and nprobe.  What goes wrong on ONE rank before
 * the exchange (bad argument, unsupported k, allocation failure, index not ready, mismatching k / nprobe) is published in the matrix:
 * EVERY rank returns an error for that step and none blocks.  A shard object replaced on one rank (reloaded part) is noticed through
 * its instance word and the index-wide list statistics are gathered again by every rank in the same step.
 * routed_pairs (nullable): the (query, rank) pairs this rank served -- its share of the step's list-scan work.  A caller-supplied
 * transport (msvs_comm_init_custom) emulates the point-to-point exchange with its all-gather (tests). */
MSVS_API int msvs_shard_search_routed_device(const msvs_index_t * shard, const msvs_comm_t * comm, const float * d_queries, size_t nq,
                                             int k, int nprobe, int64_t * d_ids, float * d_dis, void * hip_stream, uint64_t * routed_pairs);
MSVS_API int msvs_shard_search_routed_filtered_device(const msvs_index_t * shard, const msvs_comm_t * comm, const float * d_queries,
                                                      size_t nq, int k, int nprobe, const uint64_t * d_alive_bits, size_t nbits,
                                                      int64_t * d_ids, float * d_dis, void * hip_stream, uint64_t * routed_pairs);
/* The routed search with TWO STEPS IN FLIGHT: call i enqueues the FRONT phase of its batch and then the BACK phase of batch i - 1,
 * whose count matrix was gathered one call ago -- the host does not wait for the device while the list scan of step i - 1 and the
 * coarse stage of step i are still to run.  Both phases run in `hip_stream`'s order (option route_streams = 2: on the communicator's
 * own compute + exchange streams, `hip_stream` ordering the inputs only); a call on another stream than the previous one continues
 * behind it.  The results (and *routed_pairs) of batch i are complete when call i + 1's work on its stream is -- or when the event
 * handed out by call i + 1 in *prev_done_event has fired (prev_done_event may be NULL: no event is recorded then; nullptr at the
 * first call) -- or after msvs_shard_search_drain, which is COLLECTIVE while a routed step is pending (it runs that step's back
 * phase).  Buffers of a batch stay untouched until then; every rank issues the same sequence of calls. */
MSVS_API int msvs_shard_search_routed_device_async(const msvs_index_t * shard, const msvs_comm_t * comm, const float * d_queries, size_t nq,
                                                   int k, int nprobe, const uint64_t * d_alive_bits, size_t nbits, int64_t * d_ids,
                                                   float * d_dis, void * hip_stream, uint64_t * routed_pairs, void ** prev_done_event);
/* The same search with TWO BATCHES IN FLIGHT: the call returns once batch i is enqueued; `hip_stream` orders its INPUTS only,
 * its results are complete when *done_event -- a hipEvent_t owned by the communicator, valid until the second-next async call on
 * it -- has fired: hipStreamWaitEvent on whatever stream reads d_ids / d_dis, or msvs_shard_search_drain.  The coarse pass and
 * the list scan run on the communicator's compute stream, both all-gathers and the merge on its exchange stream, so the packed
 * top-k exchange + merge of batch i (latency-bound: tens of microseconds over xGMI) hide under the list scan of batch i + 1 --
 * the per-part searches of MergeTreeBaseSearchManager.cpp:207-299 overlapping with the merge of the previous query.  d_queries,
 * d_ids, d_dis of a batch must stay untouched until its event; every rank issues the same sequence of calls. */
MSVS_API int msvs_shard_search_device_async(const msvs_index_t * shard, const msvs_comm_t * comm, const float * d_queries,
                                            size_t nq, int k, int nprobe, const uint64_t * d_alive_bits, size_t nbits,
                                            int64_t * d_ids, float * d_dis, void * hip_stream, void ** done_event);
/* `hip_stream` waits for every batch still in flight on the communicator. */
MSVS_API int msvs_shard_search_drain(const msvs_comm_t * comm, void * hip_stream);

/* Multi-part / multi-GPU merge of partial top-k lists with the canonical total order -- the
 * device-side analogue of MergeTreeBaseSearchManager::getTotalTopSearchResultImpl
 * (src/VectorIndex/Storages/MergeTreeBaseSearchManager.cpp:207-299) for lists that share one id space.
 * ids/dis: [nparts][nq][k] (HOST), out: [nq][k]. */
MSVS_API int msvs_merge_topk(const int64_t * ids, const float * dis, size_t nparts, size_t nq, size_t k, int metric,
                             int64_t * out_ids, float * out_dis);
MSVS_API int msvs_merge_topk_device(const int64_t * d_ids, const float * d_dis, size_t nparts, size_t nq, size_t k,
                                    int metric, int64_t * d_out_ids, float * d_out_dis, void * hip_stream);
/* Same, but shard p's [nq][k] lists start at d_ids + p * ids_part_stride / d_dis + p * dis_part_stride (strides in
 * ELEMENTS): lets each rank all-gather ONE packed buffer {ids[nq*k] i64 | dis[nq*k] f32} and merge it in place. */
MSVS_API int msvs_merge_topk_device_strided(const int64_t * d_ids, size_t ids_part_stride, const float * d_dis,
                                            size_t dis_part_stride, size_t nparts, size_t nq, size_t k, int metric,
                                            int64_t * d_out_ids, float * d_out_dis, void * hip_stream);

/* Fusion step of a hybrid search for a batch, on the device: RankFusion / RelativeScoreFusion / computeNormalizedScore
 * (src/VectorIndex/Utils/HybridSearchUtils.cpp:164-300) as MergeTreeHybridSearchManager::hybridSearch applies them to ONE
 * part's vector and text results.  Inputs are the device searches' output arrays: query q's vector rows d_vec_dis /
 * d_vec_ids[q * kv ...] (best first; the first id < 0 ends the list), its text rows d_txt_scores / d_txt_ids[q * kt ...];
 * kv, kt <= 256.  fusion_type: 0 = RRF (score = sum of 1 / (fusion_k + rank), fusion_k 0 = 60), 1 = RSF (fusion_weight *
 * normalised text score + (1 - fusion_weight) * normalised vector score, vector_scan_direction -1: larger is better).
 * Outputs [nq][topk]: fused scores descending, ties by ascending label; label -1 / score 0 past d_n_out[q] rows.  Same
 * arithmetic, same order, same bits as msvs_host_hybrid_search_batch (include/msvs_host.h).  Stream-ordered, no host sync. */
MSVS_API int msvs_hybrid_fuse_device(int fusion_type, const float * d_vec_dis, const int64_t * d_vec_ids, size_t kv,
                                     const float * d_txt_scores, const int64_t * d_txt_ids, size_t kt, size_t nq,
                                     uint64_t fusion_k, float fusion_weight, int vector_scan_direction, size_t topk,
                                     float * d_out_scores, int64_t * d_out_labels, uint32_t * d_n_out, void * hip_stream);

/* ---------------------------------------------------------------------------------------------
 * Seam B -- BM25 posting-list scorer.  Replaces the scoring inside
 *   TANTIVY::ffi_bm25_search(index_path, sentence, column_names, topk, alive_bitmap, use_filter, enable_nlq,
 *                            operator_or, statistics)   src/Storages/MergeTree/TantivyIndexStore.cpp:908-917,939-948
 * Tokenisation / query parsing stay on the host; the postings of a part are exported once into flat arrays:
 * term t owns postings [post_off[t], post_off[t+1]) of (doc_ids ascending, tfs); fieldnorm_ids[doc] is
 * tantivy's 1-byte quantised field length.
 */
typedef struct msvs_postings msvs_postings_t;
MSVS_API int msvs_postings_create(const int64_t * post_off, size_t num_terms, const uint32_t * doc_ids,
                                  const uint32_t * tfs, const uint8_t * fieldnorm_ids, size_t num_docs,
                                  msvs_postings_t ** out);
/* Several text columns in one index (fts index on (doc, doc2): TantivyIndexStore passes column_names): every
 * (field, token) pair is a term of its own with term_field[t] = its field; fieldnorm_ids is [num_fields][num_docs].
 * num_fields <= 4. */
MSVS_API int msvs_postings_create_fields(const int64_t * post_off, size_t num_terms, const uint8_t * term_field,
                                         const uint32_t * doc_ids, const uint32_t * tfs, const uint8_t * fieldnorm_ids,
                                         size_t num_fields, size_t num_docs, msvs_postings_t ** out);
MSVS_API void msvs_postings_free(msvs_postings_t * postings);
/* The part's lightweight-delete bitmap (1 = alive; what MergeTreeTextSearchManager.cpp:199-255 hands to tantivy as
 * u8_alive_bitmap on every call), resident in HBM and swapped atomically; NULL clears.  A per-call filter is ANDed in. */
MSVS_API int msvs_postings_set_alive(msvs_postings_t * postings, const uint64_t * alive_bits, size_t nbits);
/* qterms/df: the query's term ids and their TABLE-level document frequencies (TANTIVY::Statistics.docs_freq,
 * src/VectorIndex/Processors/ReadWithHybridSearch.cpp:89-209); total_docs / total_tokens likewise.
 * Output: up to k (row id, score) best-first (score desc, row asc); *n_out = number written. */
MSVS_API int msvs_bm25_search(const msvs_postings_t * postings, const uint32_t * qterms, const uint64_t * df,
                              size_t num_qterms, uint64_t total_docs, uint64_t total_tokens,
                              const uint64_t * alive_bits, size_t nbits, size_t k, uint64_t * row_ids, float * scores,
                              size_t * n_out);
/* A BATCH of nq queries in one pass (one query is far too little work for the chip: a few MB of postings).
 * Query q owns the flat terms [qoff[q], qoff[q + 1]) of qterms / df (table-level document frequency of each
 * (field, token) term) / qgroups (nullable: the index of the query TOKEN a term stands for -- a token searched in two
 * fields is two terms of one group; NULL = every term its own group).  total_tokens: [num_fields].
 * operator_or != 0: any term matches (tantivy's default, ffi_bm25_search(..., operator_or = true)); 0: every token
 * group must match (at most 16 tokens).  Outputs [nq][k] best-first (score desc, row asc), n_out[q] = hits of query q.
 * The _device form leaves int64 ids (-1 = no hit) and scores in device buffers, stream-ordered, no host sync; its filter
 * is a DEVICE pointer. */
MSVS_API int msvs_bm25_search_batch(const msvs_postings_t * postings, size_t nq, const uint32_t * qoff,
                                    const uint32_t * qterms, const uint32_t * qgroups, const uint64_t * df,
                                    uint64_t total_docs, const uint64_t * total_tokens, int operator_or,
                                    const uint64_t * alive_bits, size_t nbits, size_t k, uint64_t * row_ids, float * scores,
                                    uint32_t * n_out);
/* ---------------------------------------------------------------------------------------------- filters (PREWHERE)
 * The reference evaluates PREWHERE with a CPU pipeline per query and part, collects the passing `_part_offset`s and
 * sets one bit per row (performPrefilter / getFilterFromPipeline,
 * src/VectorIndex/Storages/MergeTreeSelectWithHybridSearchProcessor.cpp:905-1112); the bitmap then rides on the search
 * (executeSearchWithFilter).  msvs_filter_t is that bitmap, resident on the device, with its population count:
 *   - built from host / device words, from a list of passing row offsets (exactly getFilterFromPipeline's loop), or from a
 *     simple predicate `column OP constant` evaluated on the device (row i of the column = `_part_offset` i);
 *   - combined with AND / OR / AND NOT;
 *   - handed to msvs_index_search_filter*, which picks the strategy: bit test inside the scan, or -- when few rows pass
 *     (below 3 % .. 40 % depending on how many queries share a list pass) -- the scan of a compacted view holding only the
 *     passing rows (same results, a fraction of the bytes).  Results are identical to msvs_index_search with the same bitmap. */
typedef struct msvs_filter msvs_filter_t;
typedef struct { int64_t i; double f; } msvs_scalar_t; /* .i for integer columns, .f for float columns */
enum msvs_dtype { MSVS_DT_UINT8 = 0, MSVS_DT_UINT16, MSVS_DT_UINT32, MSVS_DT_UINT64, MSVS_DT_INT8, MSVS_DT_INT16, MSVS_DT_INT32,
                  MSVS_DT_INT64, MSVS_DT_FLOAT32, MSVS_DT_FLOAT64 };
enum msvs_cmp_op { MSVS_OP_EQ = 0, MSVS_OP_NE, MSVS_OP_LT, MSVS_OP_LE, MSVS_OP_GT, MSVS_OP_GE, MSVS_OP_BETWEEN /* lo <= x <= hi */ };
enum msvs_filter_mode { MSVS_FILTER_AND = 0, MSVS_FILTER_OR = 1, MSVS_FILTER_AND_NOT = 2 };
MSVS_API int msvs_filter_from_bits(const uint64_t * bits, size_t nbits, msvs_filter_t ** out);
MSVS_API int msvs_filter_from_offsets(const uint64_t * part_offsets, size_t n, size_t nbits, int mem, msvs_filter_t ** out);
MSVS_API int msvs_filter_from_predicate(const void * column, int dtype, size_t nrows, int mem, int op, msvs_scalar_t lo,
                                        msvs_scalar_t hi, msvs_filter_t ** out);
MSVS_API int msvs_filter_combine(msvs_filter_t * a, const msvs_filter_t * b, int mode); /* a = a MODE b */
MSVS_API int msvs_filter_count(const msvs_filter_t * f, uint64_t * alive, size_t * nbits);
MSVS_API int msvs_filter_to_bits(const msvs_filter_t * f, uint64_t * bits_out);
MSVS_API void msvs_filter_free(msvs_filter_t * f);
MSVS_API int msvs_index_search_filter(const msvs_index_t * index, const float * queries, size_t nq, int k, const char * params,
                                      const msvs_filter_t * filter, int64_t * ids, float * dis);
MSVS_API int msvs_index_search_filter_device(const msvs_index_t * index, const float * d_queries, size_t nq, int k, int nprobe,
                                             const msvs_filter_t * filter, int64_t * d_ids, float * d_dis, void * hip_stream);

/* Monitoring: queries scored through the sample / cut / emit path of long corpora, and how many of them needed the
 * exact fallback (speed only; results never differ). */
MSVS_API int msvs_bm25_stats(uint64_t * queries, uint64_t * fallbacks);
MSVS_API int msvs_bm25_search_batch_device(const msvs_postings_t * postings, size_t nq, const uint32_t * qoff,
                                           const uint32_t * qterms, const uint32_t * qgroups, const uint64_t * df,
                                           uint64_t total_docs, const uint64_t * total_tokens, int operator_or,
                                           const uint64_t * d_alive_bits, size_t nbits, size_t k, int64_t * d_row_ids,
                                           float * d_scores, void * hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* MSVS_H */
