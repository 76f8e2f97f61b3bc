/* cooperative_groups subset used by the reference rasterizer: this_grid().thread_rank(), this_thread_block(). */
#pragma once
#include "cuda_runtime.h"
namespace cooperative_groups {
struct grid_group {
    unsigned long long thread_rank() const
    {
        const unsigned long long bsz = (unsigned long long)blockDim.x * blockDim.y * blockDim.z;
        const unsigned long long bid = ((unsigned long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        const unsigned long long tid = ((unsigned long long)threadIdx.z * blockDim.y + threadIdx.y) * blockDim.x + threadIdx.x;
        return bid * bsz + tid;
    }
};
struct thread_block {
    dim3 group_index() const { return dim3(blockIdx.x, blockIdx.y, blockIdx.z); }
    dim3 thread_index() const { return dim3(threadIdx.x, threadIdx.y, threadIdx.z); }
    unsigned int thread_rank() const { return (threadIdx.z * blockDim.y + threadIdx.y) * blockDim.x + threadIdx.x; }
    void sync() const { __syncthreads(); }
};
inline grid_group this_grid() { return grid_group(); }
inline thread_block this_thread_block() { return thread_block(); }
}  // namespace cooperative_groups
