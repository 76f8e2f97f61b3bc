// Internal declarations shared by the HIP translation units of libag_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/ag_raster.h"

namespace ag {

constexpr int kTileX = AG_TILE_X;
constexpr int kTileY = AG_TILE_Y;
constexpr int kWave = 64;  // CDNA4 wavefront

// Per-Gaussian record produced by the forward preprocess and consumed (gathered through the sorted tile lists) by
// the blend kernels: three 16-byte loads per instance instead of four separate gathers.
struct __attribute__((aligned(16))) GaussRec {
    float x, y;        // pixel-space mean (forward.cu:232 point_image)
    float ca, cb;      // conic a, b
    float cc, op;      // conic c, opacity
    float r, g;        // colour
    float b, depth;    // colour, view-space depth
    float r2cut;       // conservative squared pixel distance beyond which alpha < 1/255 (wave-level cull)
    float qcut;        // the same bound on q = a dx^2 + 2 b dx dy + c dy^2 = -2 power itself: 2 ln(255 op) with the same slack (exact cull of the blend backward)
};
static_assert(sizeof(GaussRec) == 48, "GaussRec must be 48 bytes");

// Per-Gaussian accumulator row of the blend backward: one 64-byte line.  With q = G * dL/dalpha of a (pixel, entry) pair and
// (dx, dy) = mean2D - pixel, the blend backward sums the six MOMENTS q, q dx, q dy, q dx^2, q dx dy, q dy^2 over the pixels; every
// geometric gradient of backward.cu:560-600 is a per-Gaussian linear map of those (opacity, conic and the viewport scale are constants
// of the Gaussian), applied once per Gaussian by the preprocess backward (accum_to_gradients) instead of once per (pixel, entry):
//   dL/dopacity = Q;  dL/dconic = -0.5 op (QXX, QXY, QYY);  dL/dmean2D = -0.5 op (W (ca QDX + cb QDY), H (cc QDY + cb QDX))
constexpr int kAccumFloats = 16;
enum AccumSlot { A_QDX = 0, A_QDY, A_QXX, A_QXY, A_QYY, A_Q, A_COLR, A_COLG, A_COLB, A_DEPTH };

// Exact cull of a splat against a rectangle of pixel centres (round 5 in the blend backward, round 6 in the forward too): the smallest value of
// q(u, v) = a u^2 + 2 b u v + c v^2 (= -2 power) over [U0, U1] x [V0, V1] (the rectangle relative to the splat) against the splat's own threshold
// qcut = 2 ln(255 op) (+ slack, ag_preprocess.hip).  q is convex with its minimum 0 at the splat, so over the rectangle it is smallest on an edge
// that faces the splat: the lines u = ue and v = ve through the rectangle's point nearest to the splat (ue = med3(0, U0, U1)); on u = ue,
// c q = (c v + b ue)^2 + det ue^2 with w = c v + b ue in [c V0 + b ue, c V1 + b ue], so min c q = med3(0, W0, W1)^2 + det ue^2 -- no division.
// A conic that is not positive definite carries qcut = 3e38 (never culled).
__device__ __forceinline__ bool rect_reaches(float U0, float U1, float V0, float V1, float ca, float cb, float cc, float qc)
{
    const float ue = __builtin_amdgcn_fmed3f(0.f, U0, U1), ve = __builtin_amdgcn_fmed3f(0.f, V0, V1);
    const float det = fmaf(ca, cc, -cb * cb);
    const float bue = cb * ue, bve = cb * ve;
    const float w1 = __builtin_amdgcn_fmed3f(0.f, fmaf(cc, V0, bue), fmaf(cc, V1, bue));
    const float w2 = __builtin_amdgcn_fmed3f(0.f, fmaf(ca, U0, bve), fmaf(ca, U1, bve));
    const float l1 = fmaf(w1, w1, det * ue * ue), l2 = fmaf(w2, w2, det * ve * ve);
    return (l1 <= cc * qc) | (l2 <= ca * qc);
}

// Blend kernels: a workgroup of 8 waves owns an 8x4 pixel region (8 regions per 16x16 tile); each wave blends
// 4 pixels (one per 16-lane DPP row) x 16 list entries per step.
constexpr int kRegW = 8, kRegH = 4, kRegionsPerTile = (kTileX / kRegW) * (kTileY / kRegH);
constexpr int kBlendThreads = 512;
constexpr int kChunk = 512;   // tile-list entries culled cooperatively per pass (one per thread)
constexpr int kQueues = 8;    // XCDs: the static work assignment keeps all regions of a tile on one of them

// Work distribution of the persistent blend kernels: STATIC.  (A global work queue was measured first: returning
// atomics on a few hot words sustain only ~5-10 pops/us on this part, 100+ us per frame for ~5000 items.)
// Tiles are ranked longest-list-first by tile_scan_kernel.  Workgroup b sits on XCD b % 8 (observed dispatcher
// placement, used for speed only) and is the (b / 8)-th workgroup of that XCD.  XCD x owns tile ranks x, x+8, x+16, ...
// and ALL regions of those tiles, so the workgroups that gather one tile's list share an L2; inside the XCD the
// (tile, region) items are dealt round-robin, which with the longest-first order is the usual LPT balance.
//
// Balance: the cost of an item depends on how many of the tile's entries survive the region's cull and on where its pixels
// saturate, which no length-based order predicts -- with exactly as many workgroups as resident slots (512 in the backward)
// their end times spread over 82-154 us of a 154-us kernel (busy fraction 0.63-0.76, profiles/bwd_wg_times.py).  A software
// queue costs more than it gains (32 counters, tickets drawn two items ahead: backward 166 -> 146 us but forward 67 -> 75 us
// and the whole step 10 % SLOWER: device-scope returning atomics take microseconds under the flush traffic).  The hardware
// dispatcher does it for free: with kBlendGrid = 2048 workgroups each takes 2-3 items by the same static rule and late
// workgroups start on whichever CU frees a slot first: forward 67 -> 59 us, backward 166 -> 152 us, step +14-20 %.
// (Rounds 1-2: 4096-8192 workgroups gave the same kernel times and two concurrent streams then starved each other.  Round 3, with the
// backward on 16384 single-wave workgroups and the leaner forward: 2048 / 4096 / 6144 / 8192 / 16384 workgroups = 50.2 / 47.1 / 46.1 /
// 46.3 / 48.4 us forward, 7423 / 7501 / 7534 / 7475 / 7375 views/s with three views in flight, 1793 -> 1844 views/s at 2048^2 --
// profiles/ab_kernels.sh.  One item per workgroup up to 768 active tiles.)
#ifndef AG_BLEND_GRID
#define AG_BLEND_GRID 6144
#endif
constexpr int kBlendGrid = AG_BLEND_GRID;

struct ItemIter {
    uint32_t i, stride, x, n_active;
    __device__ __forceinline__ ItemIter(uint32_t block, uint32_t grid, uint32_t n_active_)
        : i(block / kQueues), stride((grid + kQueues - 1 - (block % kQueues)) / kQueues), x(block % kQueues), n_active(n_active_) {}
    // next (tile rank, region) of this workgroup; false when exhausted
    __device__ __forceinline__ bool next(uint32_t& tile_rank, uint32_t& region)
    {
        tile_rank = (i / kRegionsPerTile) * kQueues + x;
        region = i % kRegionsPerTile;
        i += stride;
        return tile_rank < n_active;
    }
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope release fence, which on
// gfx950 makes every wave wait for the acknowledgement of its outstanding global stores/atomics (vmcnt(0)) -- and, because
// vmcnt retires in order, for any prefetch issued before them.  The blend kernels exchange data between waves through
// LDS only (global memory is written with fire-and-forget stores/atomics), so they use this instead.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Bounding window of the tile rects of a workgroup's Gaussians, for the window histograms of the preprocess and scatter kernels.
// One LDS atomic per wave and bound after a wave reduction: 256 threads hammering the same four LDS words directly serialise lane
// by lane (1024 same-address atomics per workgroup).  Call from ALL lanes; lanes without a rect pass has = false.
__device__ __forceinline__ void window_accumulate(int* s_win, bool has, int x0, int y0, int x1, int y1)
{
    int a = has ? x0 : 0x7fffffff, b = has ? y0 : 0x7fffffff, c = has ? x1 : 0, d = has ? y1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a = min(a, __shfl_xor(a, o, 64));
        b = min(b, __shfl_xor(b, o, 64));
        c = max(c, __shfl_xor(c, o, 64));
        d = max(d, __shfl_xor(d, o, 64));
    }
    if ((threadIdx.x & 63) == 0 && c > 0) {
        atomicMin(&s_win[0], a); atomicMin(&s_win[1], b);
        atomicMax(&s_win[2], c); atomicMax(&s_win[3], d);
    }
}

inline __host__ __device__ size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct GeomLayout {
    size_t rec, cov3d, tiles_touched, clamped, total;
    __host__ __device__ explicit GeomLayout(size_t P)
    {
        size_t o = 0;
        rec = o;            o = align_up(o + P * sizeof(GaussRec), 256);
        cov3d = o;          o = align_up(o + P * 6 * sizeof(float), 256);
        tiles_touched = o;  o = align_up(o + P * sizeof(uint32_t), 256);
        clamped = o;        o = align_up(o + P * 3, 256);
        total = o + 256;
    }
};

struct ImageLayout {
    size_t ranges, tile_count, cursor, n_contrib, num_rendered, tile_order, total;
    __host__ __device__ ImageLayout(size_t W, size_t H)
    {
        const size_t T = ((W + kTileX - 1) / kTileX) * ((H + kTileY - 1) / kTileY);
        size_t o = 0;
        ranges = o;        o = align_up(o + T * 2 * sizeof(uint32_t), 256);
        tile_count = o;    o = align_up(o + T * sizeof(uint32_t), 256);
        cursor = o;        o = align_up(o + T * sizeof(uint32_t), 256);
        n_contrib = o;     o = align_up(o + W * H * sizeof(uint32_t), 256);
        num_rendered = o;  o = align_up(o + 16, 256);
        tile_order = o;    o = align_up(o + T * 4 * sizeof(uint32_t), 256);   // uint4 {tile, begin, end, 0}, longest list first
        total = o + 256;
    }
};

struct BinLayout {
    size_t keys, point_list, total;
    __host__ __device__ explicit BinLayout(size_t R)
    {
        size_t o = 0;
        keys = o;        o = align_up(o + R * sizeof(uint64_t), 256);
        point_list = o;  o = align_up(o + R * sizeof(uint32_t), 256);
        total = o + 256;
    }
};

// Scratch base pointers may be arbitrarily aligned (torch uint8 tensors are 512-B aligned in practice, but the ABI
// does not require it): all sub-arrays are addressed from the base rounded up to 256 B.
inline __host__ __device__ char* aligned_base(const void* p)
{
    return reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p) + 255) & ~uintptr_t(255));
}

void set_error(const char* fmt, ...);

// Optional event bracketing of kernel launches (ag_prof_* in the ABI).  Usage: { ProfScope ps(AG_K_X, stream); launch; }
// `work` = what the launch computes in the kernel's roofline unit (FLOPs for the convolutions), summed by ag_prof_collect.
int prof_begin(int kernel_id, hipStream_t s, double work, const char* tag = nullptr);
void prof_end(int handle, hipStream_t s);
struct ProfScope {
    int handle; hipStream_t s;
    ProfScope(int id, hipStream_t s_, double work = 0.0, const char* tag = nullptr) : handle(prof_begin(id, s_, work, tag)), s(s_) {}
    ~ProfScope() { prof_end(handle, s); }
    ProfScope(const ProfScope&) = delete;
    ProfScope& operator=(const ProfScope&) = delete;
};
int check_hip(hipError_t e, const char* what);

// launchers (one per translation unit)
int launch_preprocess(const AgRasterForwardArgs& a, hipStream_t s);
int launch_tile_scan(const AgRasterForwardArgs& a, hipStream_t s, uint32_t capacity = 0xffffffffu, bool skip_large = false);
int launch_bin_sort(const AgRasterForwardArgs& a, int R, hipStream_t s, bool skip_large = false);
int launch_blend_forward(const AgRasterForwardArgs& a, int R, hipStream_t s);
int launch_blend_backward(const AgRasterBackwardArgs& a, hipStream_t s);
int launch_preprocess_backward(const AgRasterBackwardArgs& a, hipStream_t s);
int launch_debug_atomic_rate(float* accum, int lines, int blocks, int iters, int comps, hipStream_t s);
int launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t s);

}  // namespace ag
