// Microbenchmark (round 6): throughput of LDS atomics on gfx950 by address pattern -- decides whether a design that sums per-splat terms
// with ds_add_f32 from lanes holding DIFFERENT splats is viable (profiles/r06_lds_atomic_rate.txt).
//   hipcc --offload-arch=gfx950 -O3 -o lds_atomic_rate lds_atomic_rate.hip && ./lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int MODE, int PATTERN>
__global__ void __launch_bounds__(256) k(float* out, int iters)
{
    __shared__ float s[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) s[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t h = (blockIdx.x * 256 + threadIdx.x) * 0x9E3779B9u + 12345u;
    float v = 1.0f + lane;
    for (int it = 0; it < iters; it++) {
        int a;
        if (PATTERN == 0) a = wave * 64 + lane;                         // distinct, conflict-free
        else if (PATTERN == 1) a = wave * 64;                           // all lanes one address
        else if (PATTERN == 2) { h = h * 1664525u + 1013904223u; a = ((h >> 8) % 128) * 11 + (it % 10); }   // random slot, stride 11
        else if (PATTERN == 3) a = wave * 64 + (lane >> 1);             // pairs of lanes share an address
        else if (PATTERN == 4) a = wave * 64 + (lane >> 2);             // quads share
        else a = wave * 64 + (lane >> 3);                               // 8 lanes share
        if (MODE == 0) atomicAdd(&s[a], v);
        else if (MODE == 1) atomicAdd(reinterpret_cast<uint32_t*>(&s[a]), (uint32_t)lane);
        else if (MODE == 2) s[a] = v;
        else if (MODE == 3) atomicAdd(reinterpret_cast<unsigned long long*>(&s[(a & ~1) % 4094]), (unsigned long long)lane);
        else if (MODE == 4) v += atomicAdd(&s[a], v) * 1e-30f;       // returning
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = s[threadIdx.x] + v;
}

template <int MODE, int PATTERN>
void run(const char* name, float* out)
{
    const int blocks = 256 * 4, iters = 4096;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, PATTERN>), dim3(blocks), dim3(256), 0, 0, out, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, PATTERN>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)blocks * 4 * iters;
    // per CU: blocks / 256 CUs workgroups in sequence (4 per CU resident at once)
    const double cyc_per_instr_per_cu = ms * 1e-3 * 2.4e9 / (wave_instr / 256.0);
    printf("%-40s %8.3f ms  %6.2f cycles (at 2.4 GHz) per wave instruction per CU\n", name, ms, cyc_per_instr_per_cu);
}

int main()
{
    float* out; hipMalloc(&out, 256 * 4 * 256 * sizeof(float));
#define R(M, P, N) run<M, P>(N, out)
    R(2, 0, "ds_write_b32 distinct");
    R(0, 0, "ds_add_f32 distinct");
    R(0, 1, "ds_add_f32 one address");
    R(0, 3, "ds_add_f32 pairs share");
    R(0, 4, "ds_add_f32 quads share");
    R(0, 5, "ds_add_f32 octets share");
    R(0, 2, "ds_add_f32 random slot*11+k");
    R(1, 0, "ds_add_u32 distinct");
    R(1, 1, "ds_add_u32 one address");
    R(1, 4, "ds_add_u32 quads share");
    R(1, 2, "ds_add_u32 random slot*11+k");
    R(3, 0, "ds_add_u64 distinct");
    R(3, 1, "ds_add_u64 one address");
    R(4, 0, "ds_add_rtn_f32 distinct");
    R(4, 1, "ds_add_rtn_f32 one address");
    return 0;
}
