// STUB: the four Faiss entry points tryBruteForceSearch calls (BruteForceSearch.h:80-104), declarations only.
#pragma once

#include <cstddef>
#include <cstdint>

namespace faiss
{

struct IDSelector;

/// {nh queries, k slots each, ids, values}; "minheap" (root = worst = smallest) keeps the LARGEST values: inner product;
/// "maxheap" keeps the smallest: L2.  Results are returned sorted best first, unfilled slots id -1.
struct float_minheap_array_t
{
    size_t nh, k;
    int64_t * ids;
    float * val;
};
struct float_maxheap_array_t
{
    size_t nh, k;
    int64_t * ids;
    float * val;
};

void knn_inner_product(const float * x, const float * y, size_t d, size_t nx, size_t ny, float_minheap_array_t * res,
                       const IDSelector * sel = nullptr);
void knn_L2sqr(const float * x, const float * y, size_t d, size_t nx, size_t ny, float_maxheap_array_t * res,
               const IDSelector * sel = nullptr);
/// a, b: nx / ny codes of ncodes bytes; distances as int32 Hamming counts
void hammings_knn_mc(const uint8_t * a, const uint8_t * b, size_t na, size_t nb, size_t k, size_t ncodes, int32_t * distances,
                     int64_t * labels, const IDSelector * sel = nullptr);

}

/// global in the reference's fork (BruteForceSearch.h:104)
void jaccard_knn(const uint8_t * a, const uint8_t * b, size_t na, size_t nb, size_t k, size_t ncodes, float * distances,
                 int64_t * labels, const faiss::IDSelector * sel = nullptr);
