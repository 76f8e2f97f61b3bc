/* The two CUB entry points the reference calls (rasterizer_impl.cu:166,187-190,278,304-309), with CUB's documented
 * semantics: inclusive prefix sum; STABLE least-significant-digit radix sort of (key, value) pairs over key bits
 * [begin_bit, end_bit).  d_temp_storage == nullptr is the size query. */
#pragma once
#include <algorithm>
#include <numeric>
#include <vector>
#include "cuda_runtime.h"
namespace cub {
struct DeviceScan {
    template <class In, class Out>
    static cudaError_t InclusiveSum(void* tmp, size_t& tmp_bytes, In in, Out out, int n)
    {
        if (!tmp) { tmp_bytes = 16; return cudaSuccess; }
        typename std::remove_reference<decltype(*out)>::type acc = 0;
        for (int i = 0; i < n; i++) { acc += in[i]; out[i] = acc; }
        return cudaSuccess;
    }
};
struct DeviceRadixSort {
    template <class K, class V>
    static cudaError_t SortPairs(void* tmp, size_t& tmp_bytes, const K* kin, K* kout, const V* vin, V* vout, int n,
                                 int begin_bit = 0, int end_bit = sizeof(K) * 8)
    {
        if (!tmp) { tmp_bytes = 16; return cudaSuccess; }
        const K mask = (end_bit - begin_bit >= (int)sizeof(K) * 8) ? ~K(0) : (((K(1) << (end_bit - begin_bit)) - 1) << begin_bit);
        std::vector<int> idx(n);
        std::iota(idx.begin(), idx.end(), 0);
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return (kin[a] & mask) < (kin[b] & mask); });
        for (int i = 0; i < n; i++) { kout[i] = kin[idx[i]]; vout[i] = vin[idx[i]]; }
        return cudaSuccess;
    }
};
}  // namespace cub
