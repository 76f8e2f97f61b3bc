/*
 * cuda_runtime.h — a minimal "CUDA on the CPU" execution shim.  TEST INFRASTRUCTURE ONLY (oracle/_ref build).
 *
 * Purpose: compile the reference's own rasterizer sources (cuda_rasterizer/{forward,backward,rasterizer_impl}.cu,
 * read in place from /root/reference by oracle/ref_build.py) with g++ and run them on the host, so that the CPU
 * restatement in oracle/raster_oracle.c can be pinned against the reference CODE ITSELF rather than against a
 * reading of it.  Nothing here is shipped or used by the product path.
 *
 * Model: a kernel launch runs its blocks one after another on the calling thread.  The threads of a block are
 * ucontext fibers scheduled round-robin; __syncthreads()/__syncthreads_count()/thread_block::sync() park a fiber
 * until every live fiber of the block has arrived.  Kernels that never hit a barrier in their first block run the
 * remaining blocks as plain loops.  Device memory is host memory; atomics are plain read-modify-writes (execution
 * is sequential), so float atomicAdd order = block order, thread order.
 */
#pragma once

#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include <utility>
#include <vector>

using std::ceil;
using std::exp;
using std::sqrt;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct uint2 { unsigned int x, y; };
struct uint3 { unsigned int x, y, z; };
struct int2 { int x, y; };
struct dim3 {
    unsigned int x, y, z;
    dim3(unsigned int x_ = 1, unsigned int y_ = 1, unsigned int z_ = 1) : x(x_), y(y_), z(z_) {}
};

/* CUDA's integer/float min/max overload set (math_functions.hpp) */
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline unsigned int min(unsigned int a, unsigned int b) { return a < b ? a : b; }
inline unsigned int max(unsigned int a, unsigned int b) { return a > b ? a : b; }
inline unsigned int min(unsigned int a, int b) { return min(a, (unsigned int)b); }
inline unsigned int min(int a, unsigned int b) { return min((unsigned int)a, b); }
inline unsigned int max(unsigned int a, int b) { return max(a, (unsigned int)b); }
inline unsigned int max(int a, unsigned int b) { return max((unsigned int)a, b); }
inline float min(float a, float b) { return fminf(a, b); }
inline float max(float a, float b) { return fmaxf(a, b); }

typedef int cudaError_t;
enum { cudaSuccess = 0 };
enum cudaMemcpyKind { cudaMemcpyHostToHost, cudaMemcpyHostToDevice, cudaMemcpyDeviceToHost, cudaMemcpyDeviceToDevice };
inline cudaError_t cudaMemcpy(void* d, const void* s, size_t n, cudaMemcpyKind) { memcpy(d, s, n); return cudaSuccess; }
inline cudaError_t cudaMemset(void* d, int v, size_t n) { memset(d, v, n); return cudaSuccess; }
inline cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
inline const char* cudaGetErrorString(cudaError_t) { return "cuda_cpu shim"; }

namespace cuemu {

struct ThreadCtx { uint3 tIdx, bIdx; dim3 bDim, gDim; };
extern ThreadCtx g_ctx;          /* coordinates of the fiber currently running */
void barrier_wait(int pred, int* count_out);

void run_grid(dim3 grid, dim3 block, void (*thunk)(void*), void* closure, const void* kernel_key);

template <class K, class... A>
void launch(dim3 grid, dim3 block, K kernel, A... args)
{
    auto body = [&]() { kernel(args...); };
    using B = decltype(body);
    run_grid(grid, block, [](void* p) { (*static_cast<B*>(p))(); }, &body, reinterpret_cast<const void*>(kernel));
}

}  // namespace cuemu

#define threadIdx (cuemu::g_ctx.tIdx)
#define blockIdx (cuemu::g_ctx.bIdx)
#define blockDim (cuemu::g_ctx.bDim)
#define gridDim (cuemu::g_ctx.gDim)

inline void __syncthreads() { cuemu::barrier_wait(0, nullptr); }
inline int __syncthreads_count(int pred) { int c = 0; cuemu::barrier_wait(pred ? 1 : 0, &c); return c; }
inline void __trap() { fprintf(stderr, "__trap()\n"); abort(); }

template <class T> inline T atomicAdd(T* addr, T v) { T old = *addr; *addr = old + v; return old; }
