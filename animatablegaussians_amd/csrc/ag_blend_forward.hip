// Forward per-tile alpha blend with colour + depth + alpha outputs (gfx950).
//
// Replaces renderCUDA<3> forward (reference cuda_rasterizer/forward.cu:261-381).  Same per-pixel recurrence:
//   power = -0.5 (a dx^2 + c dy^2) - b dx dy ; skip if power > 0
//   alpha = min(0.99, o exp(power))          ; skip if alpha < 1/255
//   stop BEFORE accumulating when T (1 - alpha) < 1e-4
//   C += c alpha T ; D += depth alpha T ; A += alpha T ; T *= (1 - alpha)
//   n_contrib = 1-based list position of the last accumulated entry
//   out = C + T bg ; out_alpha = A (accumulated weight, not 1 - T) ; out_depth = D
//
// The reference walks a tile's list serially per pixel (one 256-thread block per tile, one Gaussian at a time).  With
// an avatar only 300-650 of 4096 tiles are non-empty and side views put 6000 entries in one tile, so that serial chain
// -- not throughput -- sets the kernel time.  This kernel breaks the chain in two places:
//   * PIXELS: a tile is split into 8 regions of 8x4 pixels, each owned by one 8-wave workgroup; persistent workgroups
//     take (tile, region) items from the longest-list-first order of tile_scan_kernel by a STATIC rule (ItemIter in
//     ag_common.h: all regions of a tile on one XCD; a global work queue was measured first and cost 100+ us per frame).
//   * LIST ENTRIES: inside a wave, each 16-lane DPP row is ONE pixel and its 16 lanes are 16 CONSECUTIVE list entries.
//     The transmittance is a prefix product: T_before(e) = T * prod_{i<e} (1 - alpha_i) over the contributing entries,
//     computed with a 4-step DPP row scan; the reference's early stop is "the first entry with T_before*(1-alpha) <
//     1e-4", found with a ballot + count-trailing-zeros, everything behind it is masked off.  Per 16 entries a wave
//     issues ~50 VALU ops instead of ~45 per entry.
// Second cull, per wave: of the entries that survive the region's cull each wave keeps (as 16-bit indices in LDS, list order
// preserved) those whose cut-off disc touches ITS 2x2 pixels -- 40 % on average -- and blends only them: 61 -> 54 us, side
// views 77 -> 58 us.  (The same idea in the backward costs more than it saves: its waves are coupled by the flush barrier
// every 64 entries, so the steps a wave skips are not on the critical path.)
// Staging: the workgroup culls the tile list 512 entries at a time against its region with the record's conservative
// cut-off radius (r2cut, ag_preprocess.hip), compacts the survivors IN LIST ORDER into LDS (wave ballots + an 8-entry
// prefix), and prefetches the next 512 while blending.  Culled entries still count in n_contrib: the list position
// travels with the record.  Results differ from a serial evaluation only by the association of the transmittance
// product (a few ulp).
// Bound: VALU + LDS; HBM traffic is the compulsory 52 B/instance (x8 regions, served by L2) + 24 B/pixel.
#include <cstdlib>
#include "ag_common.h"

namespace ag {

struct BlendFwdParams {
    int W, H, gx, T;
    const uint2* __restrict__ ranges;
    const uint32_t* __restrict__ point_list;
    const GaussRec* __restrict__ rec;
    const uint4* __restrict__ tile_order;
    const uint32_t* __restrict__ counts;   // [0] instances, [1] non-empty tiles
    const float* __restrict__ bg;
    float* __restrict__ out_color;
    float* __restrict__ out_depth;
    float* __restrict__ out_alpha;
    uint32_t* __restrict__ n_contrib;
};

#define AG_ROW_SHR(n) (0x110 + (n))

template <int N>
__device__ __forceinline__ float row_shr(float v, float fill)
{
    // lane l of each 16-lane row receives lane l-N of the same row; the first N lanes receive `fill`
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), AG_ROW_SHR(N), 0xf, 0xf, false));
}

__device__ __forceinline__ float row_lane15(float v)   // row_newbcast:15: every lane receives lane 15 of its 16-lane row
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x15f, 0xf, 0xf, true));
}

__device__ __forceinline__ float row_inclusive_product(float x)
{
    // x <- row_shr(x) * x with the shift as the DPP operand of the multiply: without bound_ctrl the lanes whose source falls
    // outside the row are not written and keep x (x 1).  The compiler's own code for `x *= row_shr<N>(x, 1.0f)` is a
    // v_mov_b32 of the fill, a v_mov_b32_dpp and the v_mul.  s_nop 1: two wait states between a VALU write and a DPP read.
    asm volatile("s_nop 1\n\t"
                 "v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf"
                 : "+v"(x));
    return x;
}

__device__ __forceinline__ float row_total(float x)   // sum over the row, valid in lane 15 of the row
{
    x += row_shr<1>(x, 0.0f);
    x += row_shr<2>(x, 0.0f);
    x += row_shr<4>(x, 0.0f);
    x += row_shr<8>(x, 0.0f);
    return x;
}

__device__ __forceinline__ float row_read(float v, int lane, int k)   // value of lane k of this lane's row
{
    return __shfl(v, (lane & 48) + k, 64);
}

// AG_FWD_STATS (diagnostic build only: profiles/ub/build_variant.sh fstats ag_blend_forward -DAG_FWD_STATS, profiles/fwd_step_stats.py): lane 0 of
// every wave adds its work figures to global counters.  Not compiled into the product.
#ifdef AG_FWD_STATS
enum { FS_ITEMS = 0, FS_CHUNKS, FS_LIST, FS_REGION_SURV, FS_SUBCULL_PASSES, FS_WAVE_SURV, FS_STEPS, FS_FAST_STEPS, FS_ACTIVE_PAIRS, FS_BREAKS, FS_N = 12 };
__device__ unsigned long long g_fwd_stats[FS_N];
#define FST(k, v) do { if (lane == 0) atomicAdd(&g_fwd_stats[k], (unsigned long long)(v)); } while (0)
extern "C" int ag_debug_fwd_stats(unsigned long long* out)
{
    if (hipDeviceSynchronize() != hipSuccess) return AG_ERR_HIP;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fwd_stats), sizeof(g_fwd_stats)) != hipSuccess) return AG_ERR_HIP;
    unsigned long long zero[FS_N] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_fwd_stats), zero, sizeof(zero)) != hipSuccess) return AG_ERR_HIP;
    return AG_OK;
}
#else
#define FST(k, v) do { } while (0)
#endif

#ifndef AG_FWD_PRIO
#define AG_FWD_PRIO 0
#endif
#ifndef AG_FWD_PRIO_T1
#define AG_FWD_PRIO_T1 1024u
#define AG_FWD_PRIO_T2 2048u
#define AG_FWD_PRIO_T3 4096u
#endif
#ifndef AG_FWD_LDS_PAD
#define AG_FWD_LDS_PAD 0
#endif
#ifndef AG_FWD_WAVES_PER_SIMD
#define AG_FWD_WAVES_PER_SIMD 8      // <= 64 VGPRs: four resident workgroups per CU (71 VGPRs without the bound: 58.8 us against 52.2, profiles/ab_fwd.sh)
#endif
#if AG_FWD_WAVES_PER_SIMD
__global__ void __launch_bounds__(kBlendThreads, AG_FWD_WAVES_PER_SIMD) blend_forward_kernel(BlendFwdParams p)
#else
__global__ void __launch_bounds__(kBlendThreads) blend_forward_kernel(BlendFwdParams p)
#endif
{
    constexpr int NW = kBlendThreads / 64;
    __shared__ float4 s_rec[kChunk * 3];
    __shared__ int s_wave_cnt[2][NW];
    __shared__ int s_wave_done[NW];
    __shared__ uint16_t s_widx[NW][kChunk];   // per wave: the compacted entries that can reach ITS 2x2 pixels, in list order
#if AG_FWD_LDS_PAD          /* diagnostic: fewer resident workgroups per CU (profiles/r03_bwd_step_stats.txt: the forward's time against occupancy) */
    __shared__ uint32_t s_pad[AG_FWD_LDS_PAD / 4];
    if (threadIdx.x == 0 && blockIdx.x == 0xffffffffu) s_pad[0] = 1u;
    asm volatile("" :: "v"(s_pad[threadIdx.x & 7]));
#endif

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int row = lane >> 4, e = lane & 15;
    const uint32_t n_active = p.counts[1];

    // Empty tiles (the tail of the work order) only need their background: static share.
    for (uint32_t t = n_active + blockIdx.x; t < (uint32_t)p.T; t += gridDim.x) {
        const int tile = (int)p.tile_order[t].x;
        const int px = (tile % p.gx) * kTileX + (tid & 15), py = (tile / p.gx) * kTileY + (tid >> 4);
        if (tid < kTileX * kTileY && px < p.W && py < p.H) {
            const int pix = p.W * py + px;
            const size_t HW = (size_t)p.W * p.H;
            p.n_contrib[pix] = 0;
            p.out_color[pix] = p.bg[0];
            p.out_color[HW + pix] = p.bg[1];
            p.out_color[2 * HW + pix] = p.bg[2];
            p.out_alpha[pix] = 0.f;
            p.out_depth[pix] = 0.f;
        }
    }

    // Static work assignment (see ItemIter); the descriptor of item i+1 is fetched while item i is blended.
    ItemIter it(blockIdx.x, gridDim.x, n_active);
    uint32_t tr, rg;
    bool have = it.next(tr, rg);
    uint4 hdr = make_uint4(0u, 0u, 0u, 0u);
    if (have) { hdr = p.tile_order[tr]; hdr.w = rg; }

    while (have) {
        uint32_t tr_n, rg_n;
        const bool have_n = it.next(tr_n, rg_n);
        uint4 hdr_n = make_uint4(0u, 0u, 0u, 0u);
        if (have_n) { hdr_n = p.tile_order[tr_n]; hdr_n.w = rg_n; }

        const int tile = (int)hdr.x, reg = (int)hdr.w;
        const uint2 range = make_uint2(hdr.y, hdr.z);
#if AG_FWD_PRIO
        {   // issue priority by list length (the regions of the longest tiles are the kernel's critical path)
            const uint32_t len = range.y - range.x;
            if (len >= AG_FWD_PRIO_T3) __builtin_amdgcn_s_setprio(3);
            else if (len >= AG_FWD_PRIO_T2) __builtin_amdgcn_s_setprio(2);
            else if (len >= AG_FWD_PRIO_T1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
#endif
        const int tile_x = tile % p.gx, tile_y = tile / p.gx;
        const int rx0 = tile_x * kTileX + (reg & 1) * kRegW, ry0 = tile_y * kTileY + (reg >> 1) * kRegH;
        // the wave's 4 pixels form a 2x2 block (better coherence of the per-wave early-outs than a 4x1 strip)
        const int px = rx0 + (wave & 3) * 2 + (row & 1), py = ry0 + (wave >> 2) * 2 + (row >> 1);
        const bool inside = px < p.W && py < p.H;
        const float pxf = (float)px, pyf = (float)py;
        const float qx0f = (float)rx0, qy0f = (float)ry0, qx1f = (float)(rx0 + kRegW - 1), qy1f = (float)(ry0 + kRegH - 1);

        float T = 1.0f, Cr = 0.f, Cg = 0.f, Cb = 0.f, Dd = 0.f, Ws = 0.f;   // T: row-uniform; sums: per-lane partials
        uint32_t last = 0;
        bool done = !inside;                                                // row-uniform

        // staging pipeline: records one chunk ahead, list indices two chunks ahead
        // Every load of the pipeline is UNCONDITIONAL (lanes past the end of the list read entry range.x / its record and are masked at
        // the cull): a conditional load makes the compiler merge old and new registers right behind the load -- a wait for the data on
        // the spot, i.e. no prefetch at all (round 3, found in the ISA of the wave backward kernel; this kernel had the same stall once
        // per 512-entry chunk).
        uint32_t id_next;
        float4 r0, r1, r2;
        {
            const uint32_t k0 = range.x + tid, k1 = k0 + kChunk;
            const uint32_t id0 = p.point_list[k0 < range.y ? k0 : range.x];
            id_next = p.point_list[k1 < range.y ? k1 : range.x];
            const float4* src = reinterpret_cast<const float4*>(p.rec + id0);
            r0 = src[0]; r1 = src[1]; r2 = src[2];
        }
        // cull of chunk 0
        bool keep;
#ifndef AG_FWD_TIGHT_CULL
#define AG_FWD_TIGHT_CULL 0      /* round 6, measured and NOT kept: the exact quadratic-form cull of the backward (rect_reaches) instead of the cut-off
                                    disc keeps 172 instead of 196 entries per region and 58 instead of 72 per wave (2.9 instead of 3.4 steps), all raster
                                    tests pass -- and the kernel takes 62.3 us instead of 47.2 (74.7 with the short-circuit form): this kernel is bound by
                                    its per-item critical path, and the test's ~15 dependent instructions per cull pass sit on it while the steps it
                                    saves do not pay them back (profiles/r06_fwd_exact_cull.txt).  -DAG_FWD_TIGHT_CULL=1 builds it. */
#endif
#if AG_FWD_TIGHT_CULL
        keep = (int)(range.x + tid < range.y) & (int)rect_reaches(qx0f - r0.x, qx1f - r0.x, qy0f - r0.y, qy1f - r0.y, r0.z, r0.w, r1.x, r2.w);     // no short circuit: a branch here waits for the prefetched records on the spot
#else
        {
            const float ddx = fmaxf(fmaxf(qx0f - r0.x, r0.x - qx1f), 0.f);
            const float ddy = fmaxf(fmaxf(qy0f - r0.y, r0.y - qy1f), 0.f);
            keep = (range.x + tid < range.y) && ((ddx * ddx + ddy * ddy) <= r2.z);
        }
#endif
        unsigned long long mask = __ballot(keep);
        if (lane == 0) { s_wave_cnt[0][wave] = __popcll(mask); s_wave_done[wave] = 0; }
        lds_barrier();

        if (wave == 0) { FST(FS_ITEMS, 1); FST(FS_LIST, range.y - range.x); }
        int cpar = 0;
        for (uint32_t base = range.x; base < range.y; base += kChunk, cpar ^= 1) {
            if (wave == 0) FST(FS_CHUNKS, 1);
            // ---- ordered compaction of the survivors into LDS ----
            const uint32_t k = base + tid;
            const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            // offsets of the waves' survivors: lane w reads wave w's count, a three-step DPP prefix sum over the 8 counts, two lane reads
            // (one LDS read and 7 instructions instead of 8 reads and a 16-instruction scalar loop)
            int off, K;
            {
                static_assert(NW == 8, "the prefix below covers 8 waves");
                const int c = s_wave_cnt[cpar][lane & 7];
                int x = c;
                x += __builtin_amdgcn_update_dpp(0, x, AG_ROW_SHR(1), 0xf, 0xf, true);
                x += __builtin_amdgcn_update_dpp(0, x, AG_ROW_SHR(2), 0xf, 0xf, true);
                x += __builtin_amdgcn_update_dpp(0, x, AG_ROW_SHR(4), 0xf, 0xf, true);      // lanes 0..7: inclusive prefix
                const int wv = __builtin_amdgcn_readfirstlane(wave);
                off = __builtin_amdgcn_readlane(x, wv) - __builtin_amdgcn_readlane(c, wv);
                K = __builtin_amdgcn_readlane(x, 7);
            }
            if (keep) {
                const int slot = off + rank;
                s_rec[slot * 3 + 0] = r0;
                s_rec[slot * 3 + 1] = r1;
                s_rec[slot * 3 + 2] = make_float4(r2.x, r2.y, __uint_as_float(k - range.x + 1u), AG_FWD_TIGHT_CULL ? r2.w : r2.z);  // b, depth, 1-based position, qcut (r2cut)
            }
            // issue the gathers of chunk c+1 and the index loads of chunk c+2; both are consumed after the blend
            const uint32_t kn = k + kChunk, knn = kn + kChunk;
            {
                const float4* src = reinterpret_cast<const float4*>(p.rec + id_next);
                r0 = src[0]; r1 = src[1]; r2 = src[2];
            }
            id_next = p.point_list[knn < range.y ? knn : range.x];
            lds_barrier();

            // ---- second cull, per wave: a splat that reaches the 8x4 region reaches on average 40 % of its eight 2x2 blocks.
            // Each wave keeps the indices of the entries whose cut-off disc touches ITS block (same conservative test, so the
            // dropped entries contribute exactly nothing) and blends only those: half the steps of walking the region's list.
            int cntw = 0;
#if defined(AG_FWD_KO) && AG_FWD_KO == 1      /* timing probe (same results, more steps): no second cull -- every wave walks all K region survivors */
            if (!__all(done)) {
                for (int i0 = 0; i0 < K; i0 += 64) if (i0 + lane < K) s_widx[wave][i0 + lane] = (uint16_t)(i0 + lane);
                cntw = K;
            }
            if (false) {
#else
            if (!__all(done)) {     // a wave whose four pixels are finished has nothing to pick from this chunk
#endif
                const float bx0 = (float)(rx0 + (wave & 3) * 2), by0 = (float)(ry0 + (wave >> 2) * 2);
                for (int i0 = 0; i0 < K; i0 += 64) {
                    const int i = i0 + lane;
                    bool mine = false;
                    if (i < K) {
                        const float4 a0 = s_rec[i * 3 + 0];
                        const float r2c = s_rec[i * 3 + 2].w;
#if AG_FWD_TIGHT_CULL
                        const float cc0 = s_rec[i * 3 + 1].x;
                        mine = rect_reaches(bx0 - a0.x, bx0 + 1.0f - a0.x, by0 - a0.y, by0 + 1.0f - a0.y, a0.z, a0.w, cc0, r2c);
#else
                        const float ddx = fmaxf(fmaxf(bx0 - a0.x, a0.x - (bx0 + 1.0f)), 0.f);
                        const float ddy = fmaxf(fmaxf(by0 - a0.y, a0.y - (by0 + 1.0f)), 0.f);
                        mine = (ddx * ddx + ddy * ddy) <= r2c;
#endif
                    }
                    const unsigned long long m = __ballot(mine);
                    if (mine) s_widx[wave][cntw + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)i;
                    cntw += __popcll(m);
                }
            }

            if (wave == 0) FST(FS_REGION_SURV, K);
            FST(FS_SUBCULL_PASSES, (K + 63) / 64); FST(FS_WAVE_SURV, cntw);
            // ---- blend: 16 entries per step per pixel row ----
#if defined(AG_FWD_KO) && AG_FWD_KO == 2      /* timing probe (wrong results): the culls alone, no blend steps */
            cntw = 0;
#endif
            for (int s0 = 0; s0 < cntw; s0 += 16) {
                if (__all(done)) { FST(FS_BREAKS, 1); break; }
                FST(FS_STEPS, 1);
                const int li = s0 + e;
                const bool ev = li < cntw;
                const int ci = s_widx[wave][ev ? li : (cntw - 1)];
                const float4 a = s_rec[ci * 3 + 0];   // x, y, conic a, conic b
                const float4 b = s_rec[ci * 3 + 1];   // conic c, opacity, r, g
                const float4 c = s_rec[ci * 3 + 2];   // b, depth, position
                const float dx = a.x - pxf, dy = a.y - pyf;
                const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
                const float alpha = fminf(0.99f, b.y * __builtin_amdgcn_exp2f(power * 1.4426950408889634f));
                const bool valid = ev && !done && (power <= 0.0f) && (alpha >= 1.0f / 255.0f);
                const float fac = valid ? (1.0f - alpha) : 1.0f;
                const float pin = row_inclusive_product(fac);
                const float Tb = T * row_shr<1>(pin, 1.0f);          // transmittance in front of entry e
                const float testT = T * pin;                         // ... and behind it
                // A pixel stops once in its life, a step happens dozens of times: when no entry of the wave's four rows stops (the vote reads
                // the compare's own lane mask), every valid entry blends and the row's next T is lane 15's -- one row_newbcast instead of
                // the ballot / first-stop / ds_bpermute ladder below.
                const float stop_t = valid ? testT : 1.0f;
                FST(FS_ACTIVE_PAIRS, __popcll(__ballot(valid)));
                if (__ballot(stop_t < 0.0001f) == 0ull) {
                    FST(FS_FAST_STEPS, 1);
                    const float w = valid ? alpha * Tb : 0.f;
                    Cr += b.z * w;
                    Cg += b.w * w;
                    Cb += c.x * w;
                    Dd += c.y * w;
                    Ws += w;
                    last = valid ? __float_as_uint(c.z) : last;
                    T = row_lane15(testT);
                    continue;
                }
                const bool stop = valid && (testT < 0.0001f);
                const uint32_t rowbits = (uint32_t)(__ballot(stop) >> (lane & 48)) & 0xffffu;
                const int f = rowbits ? __builtin_ctz(rowbits) : 16; // first stopping entry of this pixel
                const bool act = valid && (e < f);
                const float w = act ? alpha * Tb : 0.f;
                Cr += b.z * w;
                Cg += b.w * w;
                Cb += c.x * w;
                Dd += c.y * w;
                Ws += w;
                last = act ? __float_as_uint(c.z) : last;
                // row state for the next step: T in front of the stopping entry, or behind entry 15
                const bool stopped = f < 16;
                T = row_read(stopped ? Tb : testT, lane, stopped ? f : 15);
                done = done || stopped;
            }

            // ---- cull of the next chunk (its records have landed by now) + completion vote ----
#if AG_FWD_TIGHT_CULL
            keep = (int)(kn < range.y) & (int)rect_reaches(qx0f - r0.x, qx1f - r0.x, qy0f - r0.y, qy1f - r0.y, r0.z, r0.w, r1.x, r2.w);
#else
            {
                const float ddx = fmaxf(fmaxf(qx0f - r0.x, r0.x - qx1f), 0.f);
                const float ddy = fmaxf(fmaxf(qy0f - r0.y, r0.y - qy1f), 0.f);
                keep = (kn < range.y) && ((ddx * ddx + ddy * ddy) <= r2.z);
            }
#endif
            mask = __ballot(keep);
            const int wdone = __all(done);
            if (lane == 0) { s_wave_cnt[cpar ^ 1][wave] = __popcll(mask); s_wave_done[wave] = wdone; }
            lds_barrier();   // counts + votes visible, s_rec free for the next compaction
            int all_done = 1;
#pragma unroll
            for (int w = 0; w < NW; w++) all_done &= s_wave_done[w];
            if (all_done) break;
        }

        // ---- per-pixel totals (lane 15 of each row) and output ----
        // row sums (valid in lane 15) with the shift as the DPP operand of the addition: 4 instructions per value (the compiler's form of
        // `x += row_shr(x)` is a v_mov_b32_dpp and the add); positions grow with e and with the step, so the row maximum is the last contributor
        uint32_t lm = last;
#define AG_ROW_SUM_STEP(N)                                                                                                        \
        asm volatile("s_nop 1\n\t"                                                                                               \
                     "v_add_f32_dpp %0, %0, %0 row_shr:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                        \
                     "v_add_f32_dpp %1, %1, %1 row_shr:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                        \
                     "v_add_f32_dpp %2, %2, %2 row_shr:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                        \
                     "v_add_f32_dpp %3, %3, %3 row_shr:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                        \
                     "v_add_f32_dpp %4, %4, %4 row_shr:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                        \
                     "v_max_u32_dpp %5, %5, %5 row_shr:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1"                              \
                     : "+v"(Cr), "+v"(Cg), "+v"(Cb), "+v"(Dd), "+v"(Ws), "+v"(lm));
        AG_ROW_SUM_STEP(1) AG_ROW_SUM_STEP(2) AG_ROW_SUM_STEP(4) AG_ROW_SUM_STEP(8)
#undef AG_ROW_SUM_STEP
        if (inside && e == 15) {
            const int pix = p.W * py + px;
            const size_t HW = (size_t)p.W * p.H;
            p.n_contrib[pix] = lm;
            p.out_color[pix] = Cr + T * p.bg[0];
            p.out_color[HW + pix] = Cg + T * p.bg[1];
            p.out_color[2 * HW + pix] = Cb + T * p.bg[2];
            p.out_alpha[pix] = Ws;
            p.out_depth[pix] = Dd;
        }
        lds_barrier();   // s_rec / counters reusable
        hdr = hdr_n;
        have = have_n;
    }
}

int launch_blend_forward(const AgRasterForwardArgs& a, int R, hipStream_t s)
{
    BlendFwdParams p;
    p.W = a.W; p.H = a.H;
    p.gx = (a.W + kTileX - 1) / kTileX;
    const int gy = (a.H + kTileY - 1) / kTileY;
    p.T = p.gx * gy;
    char* gb = aligned_base(a.geom_buffer);
    char* ib = aligned_base(a.image_buffer);
    GeomLayout gl((size_t)a.P);
    ImageLayout il((size_t)a.W, (size_t)a.H);
    p.ranges = reinterpret_cast<const uint2*>(ib + il.ranges);
    p.rec = reinterpret_cast<const GaussRec*>(gb + gl.rec);
    if (R > 0) {
        BinLayout bl((size_t)R);
        p.point_list = reinterpret_cast<const uint32_t*>(aligned_base(a.binning_buffer) + bl.point_list);
    } else {
        p.point_list = nullptr;  // every range is (0,0): never dereferenced
    }
    p.tile_order = reinterpret_cast<const uint4*>(ib + il.tile_order);
    p.counts = reinterpret_cast<const uint32_t*>(ib + il.num_rendered);
    p.bg = a.bg;
    p.out_color = a.out_color; p.out_depth = a.out_depth; p.out_alpha = a.out_alpha;
    p.n_contrib = reinterpret_cast<uint32_t*>(ib + il.n_contrib);
    // up to kBlendGrid (6144) workgroups, of which 1024 (4 per CU) are resident: one item each on avatar views (~5000 region items), the
    // longest lists first, and the hardware dispatcher starts the rest wherever a slot frees up first -- load balancing without atomics
    // (see kBlendGrid in ag_common.h for the measurements); beyond that the static rule deals several items per workgroup.
    const long long items = (long long)p.T * kRegionsPerTile;
    const int grid = (int)(items < kBlendGrid ? items : kBlendGrid);
    { ProfScope ps(AG_K_BLEND_FORWARD, s); hipLaunchKernelGGL(blend_forward_kernel, dim3(grid), dim3(kBlendThreads), 0, s, p); }
    return check_hip(hipGetLastError(), "blend_forward_kernel");
}

}  // namespace ag
