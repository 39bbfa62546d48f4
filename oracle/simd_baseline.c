/*
 * simd_baseline.c -- CPU BASELINE for bench.py's `cpu_baseline` leg.  TEST / MEASUREMENT INFRASTRUCTURE, NOT PRODUCT
 * CODE (same rule as msvs_oracle.c: only tests/, smoke() and bench.py may load it).
 *
 * What it is: the reference's SIMD CPU path for IVFFLAT / FLAT search restated the way its library (Faiss inside
 * contrib/search-index, absent from the tree) does it -- fvec_L2sqr / fvec_inner_product as fused multiply-add SIMD
 * loops (the compiler vectorises them for the build machine: -O3 -march=native, AVX-512 where the host has it), one
 * query per thread with OpenMP across queries (IndexIVF::search parallelises over queries; the brute-force path of
 * MyScaleDB pins OpenMP to 1 thread per search and runs searches on up to 2 x cores threads, ScanThreadLimiter.h), a
 * binary heap for the top-k.  It is NOT the parity oracle: fma changes the last bits of a distance, so ids are compared
 * with the GPU result as recall, not bit for bit.  Built on the machine that runs the bench (oracle/Makefile target
 * `native`) so that the timing uses that machine's SIMD width.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define API __attribute__((visibility("default")))

static inline float l2sqr(const float *restrict x, const float *restrict y, size_t d)
{
    float s = 0.f;
#pragma omp simd reduction(+ : s)
    for (size_t i = 0; i < d; i++) {
        const float t = x[i] - y[i];
        s += t * t;
    }
    return s;
}

static inline float ip(const float *restrict x, const float *restrict y, size_t d)
{
    float s = 0.f;
#pragma omp simd reduction(+ : s)
    for (size_t i = 0; i < d; i++) s += x[i] * y[i];
    return s;
}

typedef struct {
    float d;
    int64_t id;
} ent_t;

/* max-heap on "worse first": for L2 the largest distance at the root, for IP the smallest */
static inline int worse(int ipm, ent_t a, ent_t b) { return ipm ? (a.d < b.d || (a.d == b.d && a.id > b.id)) : (a.d > b.d || (a.d == b.d && a.id > b.id)); }

static void heap_push(int ipm, ent_t *h, size_t k, size_t *n, ent_t e)
{
    if (*n < k) {
        size_t i = (*n)++;
        h[i] = e;
        while (i && worse(ipm, h[i], h[(i - 1) / 2])) {
            ent_t t = h[i];
            h[i] = h[(i - 1) / 2];
            h[(i - 1) / 2] = t;
            i = (i - 1) / 2;
        }
        return;
    }
    if (!worse(ipm, h[0], e)) return;
    h[0] = e;
    size_t i = 0;
    for (;;) {
        size_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < k && worse(ipm, h[l], h[m])) m = l;
        if (r < k && worse(ipm, h[r], h[m])) m = r;
        if (m == i) break;
        ent_t t = h[i];
        h[i] = h[m];
        h[m] = t;
        i = m;
    }
}

static int ent_cmp_l2(const void *a, const void *b)
{
    const ent_t *x = a, *y = b;
    return x->d < y->d ? -1 : x->d > y->d ? 1 : (x->id < y->id ? -1 : x->id > y->id);
}
static int ent_cmp_ip(const void *a, const void *b)
{
    const ent_t *x = a, *y = b;
    return x->d > y->d ? -1 : x->d < y->d ? 1 : (x->id < y->id ? -1 : x->id > y->id);
}

/* IVFFLAT search of nq queries, OpenMP over queries.  metric 0 = L2 (squared), 1 = IP. */
API int simd_ivf_search(const float *centroids, size_t nlist, const int64_t *list_off, const float *vecs, const int64_t *ids,
                        const float *queries, size_t nq, size_t d, size_t nprobe, size_t k, int metric, int64_t *out_ids,
                        float *out_dis, int threads)
{
    if (nprobe > nlist) nprobe = nlist;
#pragma omp parallel num_threads(threads)
    {
        ent_t *ph = (ent_t *)malloc(sizeof(ent_t) * nprobe);
        ent_t *h = (ent_t *)malloc(sizeof(ent_t) * (k ? k : 1));
#pragma omp for schedule(dynamic, 4)
        for (long q = 0; q < (long)nq; q++) {
            const float *xq = queries + (size_t)q * d;
            size_t pn = 0;
            for (size_t l = 0; l < nlist; l++) {
                ent_t e = {metric ? ip(xq, centroids + l * d, d) : l2sqr(xq, centroids + l * d, d), (int64_t)l};
                heap_push(metric, ph, nprobe, &pn, e);
            }
            size_t n = 0;
            for (size_t p = 0; p < pn; p++) {
                const int64_t l = ph[p].id;
                for (int64_t r = list_off[l]; r < list_off[l + 1]; r++) {
                    ent_t e = {metric ? ip(xq, vecs + (size_t)r * d, d) : l2sqr(xq, vecs + (size_t)r * d, d), ids[r]};
                    heap_push(metric, h, k, &n, e);
                }
            }
            qsort(h, n, sizeof(ent_t), metric ? ent_cmp_ip : ent_cmp_l2);
            for (size_t j = 0; j < k; j++) {
                out_ids[(size_t)q * k + j] = j < n ? h[j].id : -1;
                out_dis[(size_t)q * k + j] = j < n ? h[j].d : (metric ? -FLT_MAX : FLT_MAX);
            }
        }
        free(ph);
        free(h);
    }
    return 0;
}

/* exhaustive k-NN (FLAT), OpenMP over queries */
API int simd_knn(const float *x, const float *y, size_t d, size_t k, size_t nx, size_t ny, int metric, int64_t *out_ids,
                 float *out_dis, int threads)
{
#pragma omp parallel num_threads(threads)
    {
        ent_t *h = (ent_t *)malloc(sizeof(ent_t) * (k ? k : 1));
#pragma omp for schedule(dynamic, 1)
        for (long q = 0; q < (long)nx; q++) {
            const float *xq = x + (size_t)q * d;
            size_t n = 0;
            for (size_t r = 0; r < ny; r++) {
                ent_t e = {metric ? ip(xq, y + r * d, d) : l2sqr(xq, y + r * d, d), (int64_t)r};
                heap_push(metric, h, k, &n, e);
            }
            qsort(h, n, sizeof(ent_t), metric ? ent_cmp_ip : ent_cmp_l2);
            for (size_t j = 0; j < k; j++) {
                out_ids[(size_t)q * k + j] = j < n ? h[j].id : -1;
                out_dis[(size_t)q * k + j] = j < n ? h[j].d : (metric ? -FLT_MAX : FLT_MAX);
            }
        }
        free(h);
    }
    return 0;
}

/* reports what the compiler targeted: 16 = AVX-512, 8 = AVX2, 4 = SSE */
API int simd_lanes(void)
{
#if defined(__AVX512F__)
    return 16;
#elif defined(__AVX2__)
    return 8;
#else
    return 4;
#endif
}
