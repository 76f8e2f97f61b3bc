// Vector-memory instruction throughput per CU for the access shapes the convolution loaders can use (L2-resident data):
//   0  global_load_dword,   lanes contiguous (256 B per wave instruction)
//   1  global_load_dwordx4, lanes contiguous, 16-byte aligned (1 KB per wave instruction)
//   2  global_load_dwordx4, lanes contiguous, shifted by 4 bytes (tap dx = +-1 of a 3 x 3 convolution)
//   3  global_load_dwordx4, 16 lanes on 16 different planes (channel-fastest patch loader), 4 lane groups contiguous
//   4  global_load_dwordx2, lanes contiguous, shifted by 4 bytes
// build: hipcc --offload-arch=gfx950 -O3 -o load_rate load_rate.hip ; run: ./load_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(512, 4) load_rate_kernel(const float* __restrict__ src, float* out, int iters, int span_floats)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // every workgroup walks its own window of the buffer; windows overlap across workgroups (weights-like reuse)
    size_t base = ((size_t)blockIdx.x * 8191 + wave * 1031) % (size_t)(span_floats - (1 << 16));
    base &= ~size_t(3);
    float acc = 0.f;
    for (int it = 0; it < iters; it++) {
        const size_t o = base + (size_t)(it & 31) * 1024;
        if constexpr (MODE == 0) {
            acc += src[o + lane];
        } else if constexpr (MODE == 1) {
            f32x4 v = *reinterpret_cast<const f32x4*>(src + o + lane * 4);
            acc += v[0] + v[1] + v[2] + v[3];
        } else if constexpr (MODE == 2) {
            f32x4 v = *reinterpret_cast<const f32x4*>(src + o + lane * 4 + 1);
            acc += v[0] + v[1] + v[2] + v[3];
        } else if constexpr (MODE == 3) {
            f32x4 v = *reinterpret_cast<const f32x4*>(src + o + (size_t)(lane & 15) * 4096 + (lane >> 4) * 4);
            acc += v[0] + v[1] + v[2] + v[3];
        } else {
            f32x2 v = *reinterpret_cast<const f32x2*>(src + o + lane * 2 + 1);
            acc += v[0] + v[1];
        }
    }
    if (acc == 1234.5f) out[0] = acc;
}

template <int MODE>
static void run(const float* src, float* out, int span, const char* name, int bytes_per_instr)
{
    const int blocks = 512, iters = 4096;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(load_rate_kernel<MODE>, dim3(blocks), dim3(512), 0, 0, src, out, 64, span);
    hipEventRecord(a);
    hipLaunchKernelGGL(load_rate_kernel<MODE>, dim3(blocks), dim3(512), 0, 0, src, out, iters, span);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double instr = (double)blocks * 8 * iters;              // wave instructions
    const double per_cu_ns = ms * 1e6 / (instr / 256.0);
    printf("%-46s %7.1f us  %6.2f ns per wave-instruction per CU (%5.1f cycles @2.1 GHz)  %6.2f TB/s\n", name, ms * 1e3, per_cu_ns,
           per_cu_ns * 2.1, instr * bytes_per_instr / (ms * 1e-3) / 1e12);
}

int main()
{
    const int span = 4 << 20;       // 16 MB of floats: L2 + MALL resident
    float *src, *out;
    hipMalloc(&src, (size_t)span * 4 + 65536);
    hipMalloc(&out, 64);
    hipMemset(src, 0, (size_t)span * 4 + 65536);
    run<0>(src, out, span, "dword   contiguous", 256);
    run<1>(src, out, span, "dwordx4 contiguous aligned", 1024);
    run<2>(src, out, span, "dwordx4 contiguous +4 B", 1024);
    run<3>(src, out, span, "dwordx4 16 planes x 4 quads", 1024);
    run<4>(src, out, span, "dwordx2 contiguous +4 B", 512);
    return 0;
}
