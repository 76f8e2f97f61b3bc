// Backward per-tile alpha blend: gradients w.r.t. 2D mean, conic, opacity, colour and depth (gfx950).
//
// Replaces renderCUDA<3> backward (reference cuda_rasterizer/backward.cu:415-601): back-to-front replay from
// T_final = 1 - alpha_out with T recovered by division, the depth and alpha-output gradient terms, the background
// term, no gradient gate at the 0.99 alpha clamp, dL_dmean2D in NDC-scaled units (x 0.5 W, x 0.5 H).
//
// One independent WAVE per (tile, 4 x 4-pixel block): the wave culls the tile's depth-sorted list against its block (rearmost first,
// only up to the block's largest n_contrib), compacts the survivors in processing order into a wave-private LDS ring and blends four
// of them per step with lanes = 16 pixels x 4 entries.  The reference's serial recurrences become scans over the 4 entries of a step
// (two DPP row shifts) with a carry in bank 0:
//   T_e      = T / prod_{i<=e} (1 - alpha_i)                          inclusive product scan (the divisions of :534)
//   behind_e = m_{e-1}( ... m_0(behind_0)),  m_i(S) = alpha_i w_i + (1 - alpha_i) S
//                                                                      scan of affine maps over ONE scalar: the colour / depth / alpha
//                                                                      "accum_rec" of :541-565 enter dL/dalpha only through their dot
//                                                                      product with the pixel's gradient
// after which the terms of the 4 entries are independent.  Per (pixel, entry) the kernel forms the six moments of q = G dL/dalpha
// (ag_common.h AccumSlot) and the four colour / depth terms, sums them over the 16 pixels (v_permlane32_swap, v_permlane16_swap, two quad
// butterflies: 10 -> 5 -> 3 registers) into a 1-KB window in LDS and flushes 16 entries at a time with one global atomic per (block,
// splat, component), laid out so that 16 adjacent lanes hit the 16 slots of ONE 64-byte accumulator line (the atomic units work per
// line request).  The reference issues ten atomics per (pixel, splat).  The round-2 design (8-wave workgroups on 8 x 4 regions, 16
// entries per DPP row, per-wave partial slabs; 125 us on the bench view against 88) is commit 7b5ac22's version of this file;
// profiles/r03_bwd_timeline.txt is its per-item timeline, which motivated this one.
#include <cstdlib>
#include "ag_common.h"

namespace ag {

struct BlendBwdParams {
    int W, H, gx, T;
    const uint4* __restrict__ tile_order;
    const uint32_t* __restrict__ counts;
    const uint32_t* __restrict__ point_list;
    const GaussRec* __restrict__ rec;
    const float* __restrict__ bg;
    const float* __restrict__ alphas;
    const uint32_t* __restrict__ n_contrib;
    const float* __restrict__ dL_dpix;
    const float* __restrict__ dL_ddepth;
    const float* __restrict__ dL_dalpha;
    float* __restrict__ accum;  // [P, 16]
};

#define AG_DPP_QUAD_PERM(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
#define AG_DPP_ROW_SHL(n) (0x100 + (n))
#define AG_DPP_ROW_SHR(n) (0x110 + (n))
#define AG_DPP_ROW_ROR(n) (0x120 + (n))

template <int CTRL>
__device__ __forceinline__ float dpp(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

template <int N>
__device__ __forceinline__ float row_shr(float v, float fill)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), AG_DPP_ROW_SHR(N), 0xf, 0xf, false));
}

template <int N>
__device__ __forceinline__ float row_shr0(float v)   // zero fill: foldable into the DPP operand of an FMA
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), AG_DPP_ROW_SHR(N), 0xf, 0xf, true));
}

__device__ __forceinline__ float row_read(float v, int lane, int k)   // value of lane k of this lane's 16-lane row
{
    return __shfl(v, (lane & 48) + k, 64);
}

__device__ __forceinline__ void swap32(float& a, float& b)
{
    // lanes 32-63 of a <-> lanes 0-31 of b
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}

__device__ __forceinline__ void swap16(float& a, float& b)
{
    // odd 16-lane rows of a <-> even rows of b
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}

// What the per-item timeline of the round-2 region kernel showed (profiles/r03_bwd_timeline.txt, bench view): its body runs at the rate six
// waves sharing a SIMD's VALU allow (1.8 us per 32-entry sub-chunk = the ~180 VALU instructions of each of the six), so the kernel is
// bound by instruction count, and 84 % of its (pixel, entry) lane slots are dead: a wave walks ALL survivors of the 8x4 region cull
// (134 per item) for its 2x2 pixels although only 44 reach them, more than half of its 16-entry steps are skipped after paying
// for their loads / exp / tests, and the executed ones have 35 % active lanes; 26 % of an item's time is its three-round-trip start-up
// and the eight waves of a workgroup meet at a barrier every 32 entries.  Here:
//   * lanes = 16 pixels (4 x 4 block) x 4 list entries; lane = y * 16 + e * 4 + x, so a pixel's 4 entries sit in the 4 banks of a DPP
//     row: the affine-map scan is TWO row_shr steps (4, 8) instead of four, the carry to the next step is a bank-masked row_shl pair,
//     and the sum over the 16 pixels is the same permlane32/16 swap pair (over y) plus two quad butterflies (over x);
//   * the wave culls the tile list against ITS 4 x 4 pixels (108 survivors of 475 walked per block on bench view 0 -- profiles/
//     r03_bwd_step_stats.txt -- against 134 of 446 for the 8x4 region) and blends exactly those: 27 dense steps per block, 27 % of their
//     (pixel, entry) lanes active;
//   * sums of 16 blended entries collect in a 1-KB wave-private LDS slab and leave as line-coalesced atomics (16 adjacent lanes = one
//     64-byte accumulator line, as before); line requests: one per (block, survivor) = 0.85 M against 0.68 M -- still far inside
//     the 20 lines / ns the memory side retires;
//   * waves never wait for each other: the start-up chain of one item hides behind the other five waves of its SIMD.
constexpr int kBlk = 4;                                  // 4 x 4 pixels per wave
constexpr int kBlocksPerTile = (kTileX / kBlk) * (kTileY / kBlk);
#ifndef AG_BWD_WAVE_GRID
#define AG_BWD_WAVE_GRID 16384   /* one item per workgroup up to 1024 tiles x 16 blocks: 106.7 us against 108.3 with 8192 (profiles/ab_bwd.sh) */
#endif
#ifndef AG_BWD_WIN
#define AG_BWD_WIN 16
#endif
constexpr int kWaveGrid = AG_BWD_WAVE_GRID;              // single-wave workgroups; the hardware dispatcher balances them
#ifndef AG_BWD_RING
#define AG_BWD_RING 128
#endif
constexpr int kRing = AG_BWD_RING;                       // compacted records waiting to be blended (< 4 left over + <= 64 new); power of two: slot = position & 127
constexpr int kWin = AG_BWD_WIN;                         // blended entries per atomic flush

struct WaveItemIter {
    uint32_t i, stride, x, n_active;
    __device__ __forceinline__ WaveItemIter(uint32_t block, uint32_t grid, uint32_t n_active_)
        : i(block / kQueues), stride((grid + kQueues - 1 - (block % kQueues)) / kQueues), x(block % kQueues), n_active(n_active_) {}
    __device__ __forceinline__ bool next(uint32_t& tile_rank, uint32_t& blk)
    {
        tile_rank = (i / kBlocksPerTile) * kQueues + x;      // all 16 blocks of a tile on one XCD (they gather the same list)
        blk = i % kBlocksPerTile;
        i += stride;
        return tile_rank < n_active;
    }
};

// maximum over the 64 lanes as a wave-uniform value (DPP row reduction, two cross-row broadcasts, one v_readlane; no LDS round trips)
__device__ __forceinline__ uint32_t wave_umax(uint32_t v)
{
#define AG_UMAX_DPP(CTRL, ROWMASK) v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, false))
    AG_UMAX_DPP(AG_DPP_ROW_SHR(1), 0xf); AG_UMAX_DPP(AG_DPP_ROW_SHR(2), 0xf); AG_UMAX_DPP(AG_DPP_ROW_SHR(4), 0xf); AG_UMAX_DPP(AG_DPP_ROW_SHR(8), 0xf);
    AG_UMAX_DPP(0x142, 0xa);      // row_bcast:15 -> rows 1, 3
    AG_UMAX_DPP(0x143, 0xc);      // row_bcast:31 -> rows 2, 3
#undef AG_UMAX_DPP
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

template <int CTRL, int BANK_MASK>
__device__ __forceinline__ float dpp_banks(float old, float v)   // lanes of the banks in BANK_MASK take the DPP source, the others keep `old`
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xf, BANK_MASK, false));
}
// value of bank 3 (entry 3) of this lane's row, same x: bank 2 <- bank 3, then banks 0, 1 <- banks 2, 3
__device__ __forceinline__ float from_entry3(float v)
{
    v = dpp_banks<AG_DPP_ROW_SHL(4), 0x4>(v, v);
    return dpp_banks<AG_DPP_ROW_SHL(8), 0x3>(v, v);
}

// AG_BWD_STATS (diagnostic build only: profiles/ub/build_variant.sh stats ag_blend_backward -DAG_BWD_STATS, profiles/bwd_step_stats.py): lane 0
// of every wave adds its work figures to global counters.  Not compiled into the product.
#ifdef AG_BWD_STATS
enum { ST_ITEMS = 0, ST_WALKED, ST_SURVIVORS, ST_STEPS, ST_SKIPPED, ST_ACTIVE_PAIRS, ST_ACTIVE_ENTRIES, ST_STEPS_BY_ACTIVE_ENTRIES /* 5 slots */, ST_N = 12 };
__device__ unsigned long long g_bwd_stats[ST_N];
#define ST(k, v) do { if (lane == 0) atomicAdd(&g_bwd_stats[k], (unsigned long long)(v)); } while (0)
extern "C" int ag_debug_bwd_stats(unsigned long long* out)
{
    if (hipDeviceSynchronize() != hipSuccess) return AG_ERR_HIP;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bwd_stats), sizeof(g_bwd_stats)) != hipSuccess) return AG_ERR_HIP;
    unsigned long long zero[ST_N] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_stats), zero, sizeof(zero)) != hipSuccess) return AG_ERR_HIP;
    return AG_OK;
}
#else
#define ST(k, v) do { } while (0)
#endif

#ifndef AG_BWD_TIGHT_CULL
#define AG_BWD_TIGHT_CULL 1
#endif
#ifndef AG_BWD_PRIO
#define AG_BWD_PRIO 0
#endif
#ifndef AG_BWD_PRIO_T1
#define AG_BWD_PRIO_T1 512u
#define AG_BWD_PRIO_T2 1024u
#define AG_BWD_PRIO_T3 2048u
#endif
#ifndef AG_BWD_WAVE_OCC
#define AG_BWD_WAVE_OCC 6       // waves per SIMD the register budget is cut for
#endif
__global__ void __launch_bounds__(64, AG_BWD_WAVE_OCC) blend_backward_wave_kernel(BlendBwdParams p)
{
    __shared__ float4 s_ring[kRing * 3];          // [x, y, ca, cb] [cc, op, r, g] [b, depth, position, gaussian id]
    __shared__ float s_out[kWin * 16];            // [entry of the window][accumulator slot]
    __shared__ uint32_t s_wgid[kWin];

    const int lane = threadIdx.x;
    const int ry = lane >> 4, e = (lane >> 2) & 3, qx = lane & 3;
    const uint32_t n_active = p.counts[1];
    const size_t HW = (size_t)p.W * p.H;
    const float bg0 = p.bg[0], bg1 = p.bg[1], bg2 = p.bg[2];
    // flush lane (entry, component): component -> window slot [row][4]: rows hold the values (0, 1, 2) (3, 4, -) (5, 6, 7) (8, 9, -)
    const int out_comp = lane & 15;
    const int out_slot = out_comp < 3 ? out_comp : out_comp < 5 ? out_comp + 1 : out_comp < 8 ? out_comp + 3 : out_comp < 10 ? out_comp + 4 : 15;
    const float bank0 = (e == 0) ? 1.0f : 0.0f;

    WaveItemIter it(blockIdx.x, gridDim.x, n_active);
    uint32_t tr, bk;
    while (it.next(tr, bk)) {
        const uint4 hdr = p.tile_order[tr];
        const int tile = (int)hdr.x;
        const uint32_t rbeg = hdr.y;
        const int tile_x = tile % p.gx, tile_y = tile / p.gx;
        const int bx0 = tile_x * kTileX + (int)(bk & 3) * kBlk, by0 = tile_y * kTileY + (int)(bk >> 2) * kBlk;
        const int px = bx0 + qx, py = by0 + ry;
        const bool inside = px < p.W && py < p.H;
        const float pxf = (float)px, pyf = (float)py;
        const float qx0f = (float)bx0, qy0f = (float)by0, qx1f = (float)(bx0 + kBlk - 1), qy1f = (float)(by0 + kBlk - 1);
        const int pix = p.W * py + px;
        uint32_t last_contributor = 0;
        float T_final = 0.f, gr = 0.f, gg = 0.f, gb = 0.f, gd = 0.f, ga = 0.f;
        if (inside) {
            last_contributor = p.n_contrib[pix];
            T_final = 1.0f - p.alphas[pix];
            gr = p.dL_dpix[pix];
            gg = p.dL_dpix[HW + pix];
            gb = p.dL_dpix[2 * HW + pix];
            gd = p.dL_ddepth[pix];
            ga = p.dL_dalpha[pix];
        }
        const float ntf_bg = -T_final * (bg0 * gr + bg1 * gg + bg2 * gb);     // background term of dL/dalpha_e, times 1 / (1 - alpha_e)
        const uint32_t wmax = wave_umax(last_contributor);
        if (wmax == 0) continue;
        const uint32_t rend = rbeg + wmax;                   // walk [rbeg, rend) from the back
#if AG_BWD_PRIO
        // issue priority by list length: the blocks of the longest tiles are the kernel's critical path (they start first and are still
        // walking when the short items of the tail share their SIMD)
        if (wmax >= AG_BWD_PRIO_T3) __builtin_amdgcn_s_setprio(3);
        else if (wmax >= AG_BWD_PRIO_T2) __builtin_amdgcn_s_setprio(2);
        else if (wmax >= AG_BWD_PRIO_T1) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
#endif
        ST(ST_ITEMS, 1); ST(ST_WALKED, wmax);

        float P = 1.0f;              // prod (1 - alpha) over the entries done so far (all four lanes of the pixel)
        float S = 0.f;               // g . (blended state behind the entries done so far), bank 0 only
        // staging pipeline: records one pass (64 entries) ahead, indices two passes ahead.  Every load is unconditional (lanes past the
        // end read entry rbeg / record 0 and are masked at the cull): a conditional load makes the compiler merge old and new
        // registers right behind the load, i.e. wait for it on the spot.
        uint32_t id_cur = p.point_list[(uint32_t)lane < wmax ? rend - 1u - (uint32_t)lane : rbeg];
        uint32_t id_next = p.point_list[(uint32_t)lane + 64u < wmax ? rend - 1u - ((uint32_t)lane + 64u) : rbeg];
        float4 r0, r1, r2;
        float r_cc;      // conic c once more, by a load of its own: read out of r1 the compiler parks r1.yzw in scratch memory until the ring write
        {
            const float4* src = reinterpret_cast<const float4*>(p.rec + id_cur);
            r0 = src[0]; r1 = src[1]; r2 = src[2];
            r_cc = p.rec[id_cur].cc;
        }

        int head = 0, cnt = 0;       // ring positions (wave-uniform), monotonically increasing; slot = position % kRing
        int win = 0;                 // entries in the output window
        auto flush = [&]() {
            // all LDS reads first (unconditional: inside the arrays), then the atomics: one wave instruction per 4 entries, the 16 lanes
            // of a row on the 16 slots of one accumulator line
            float val[kWin / 4];
            uint32_t gid[kWin / 4];
#pragma unroll
            for (int pass = 0; pass < kWin / 4; pass++) {
                val[pass] = s_out[(pass * 4 + (lane >> 4)) * 16 + out_slot];
                gid[pass] = s_wgid[pass * 4 + (lane >> 4)];
            }
#pragma unroll
            for (int pass = 0; pass < kWin / 4; pass++) {
                const int ent = pass * 4 + (lane >> 4), comp = lane & 15;
#ifdef AG_BWD_NO_ATOMICS      /* timing probe only (wrong sums): plain stores in place of the atomics -- what do the line-coalesced atomics cost? */
                if (ent < win && comp < 10 && val[pass] != 0.f) p.accum[(size_t)gid[pass] * kAccumFloats + comp] = val[pass];
#else
                if (ent < win && comp < 10 && val[pass] != 0.f) atomicAdd(p.accum + (size_t)gid[pass] * kAccumFloats + comp, val[pass]);
#endif
            }
            win = 0;
        };

        for (uint32_t done = 0; done < wmax; done += 64u) {
            const uint32_t o = done + (uint32_t)lane;        // offset from the back
            // ---- cull against this wave's 4 x 4 pixels, ordered compaction into the ring ----
#if AG_BWD_TIGHT_CULL
            // Exact cull (round 5): the smallest value of q(u, v) = a u^2 + 2 b u v + c v^2 (= -2 power) over the rectangle [U0, U1] x [V0, V1]
            // of the block's pixel centres relative to the splat, against the splat's own threshold qcut = 2 ln(255 op) (+ slack,
            // ag_preprocess.hip).  q is convex with its minimum 0 at the splat, so over the rectangle it is smallest on an edge that
            // faces the splat: the vertical line u = ue and the horizontal line v = ve through the rectangle's point nearest to the
            // splat (ue = med3(0, U0, U1); a zero means the splat lies inside that range and the line's minimum is a feasible point of
            // the other one's).  On u = ue:  c q = (c v + b ue)^2 + det ue^2  with  w = c v + b ue  in [c V0 + b ue, c V1 + b ue], so
            // min c q = med3(0, W0, W1)^2 + det ue^2 -- no division; the same on v = ve with a.  The disc test this replaces (isotropic
            // radius from the larger eigenvalue) kept 108 entries per block of which 24 % were active on no pixel and the others on 4.3 of 16.
            const float U0 = qx0f - r0.x, U1 = qx1f - r0.x, V0 = qy0f - r0.y, V1 = qy1f - r0.y;
            const float ue = __builtin_amdgcn_fmed3f(0.f, U0, U1), ve = __builtin_amdgcn_fmed3f(0.f, V0, V1);
            const float ca = r0.z, cb = r0.w, cc = r_cc, qc = r2.w;
            const float det = fmaf(ca, cc, -cb * cb);
            const float bue = cb * ue, bve = cb * ve;
            const float w1 = __builtin_amdgcn_fmed3f(0.f, fmaf(cc, V0, bue), fmaf(cc, V1, bue));
            const float w2 = __builtin_amdgcn_fmed3f(0.f, fmaf(ca, U0, bve), fmaf(ca, U1, bve));
            const float l1 = fmaf(w1, w1, det * ue * ue), l2 = fmaf(w2, w2, det * ve * ve);
            const bool keep = (o < wmax) && ((l1 <= cc * qc) | (l2 <= ca * qc));
#else
            const float ddx = fmaxf(fmaxf(qx0f - r0.x, r0.x - qx1f), 0.f);
            const float ddy = fmaxf(fmaxf(qy0f - r0.y, r0.y - qy1f), 0.f);
            const bool keep = (o < wmax) && ((ddx * ddx + ddy * ddy) <= r2.z);
#endif
            const unsigned long long mask = __ballot(keep);
            if (keep) {
                const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                const int slot = (kRing & (kRing - 1)) ? (cnt + rank) % kRing : (int)((uint32_t)(cnt + rank) & (uint32_t)(kRing - 1));
                s_ring[slot * 3 + 0] = r0;
                s_ring[slot * 3 + 1] = r1;
                s_ring[slot * 3 + 2] = make_float4(r2.x, r2.y, __uint_as_float(wmax - o), __uint_as_float(id_cur));
            }
            cnt += __popcll(mask);
            ST(ST_SURVIVORS, __popcll(mask));
            // ---- next pass's records, the indices of the one after ----
            const uint32_t on = o + 64u, onn = on + 64u;
            id_cur = id_next;
            {
                const float4* src = reinterpret_cast<const float4*>(p.rec + id_next);
                r0 = src[0]; r1 = src[1]; r2 = src[2];
                r_cc = p.rec[id_next].cc;
            }
            id_next = p.point_list[onn < wmax ? rend - 1u - onn : rbeg];
            (void)on;
            const bool last_pass = done + 64u >= wmax;
            // The atomics share vmcnt with the loads above and gfx950 may retire the two kinds out of order, so the wait for the next
            // pass's records drains every atomic in flight: issue the window left over from the previous pass NOW, a whole blend phase
            // before that wait, instead of at the end of its own phase.
            if (win) flush();

            // ---- blend 4 ring entries per step ----
#if defined(AG_BWD_KO) && AG_BWD_KO == 1      /* timing probe (wrong results): the walk alone -- loads, cull, ring writes; no blend steps */
            head = cnt;
#endif
            while (cnt - head >= 4 || (last_pass && cnt > head)) {
                const int idx = head + e;
                const bool ev = idx < cnt;
                const int slot = (kRing & (kRing - 1)) ? (ev ? idx : cnt - 1) % kRing : (int)((uint32_t)(ev ? idx : cnt - 1) & (uint32_t)(kRing - 1));
                head += 4;
                const float4 a = s_ring[slot * 3 + 0];   // x, y, conic a, conic b
                const float4 b = s_ring[slot * 3 + 1];   // conic c, opacity, r, g
                const float4 c = s_ring[slot * 3 + 2];   // b, depth, position, id
                const float dx = a.x - pxf, dy = a.y - pyf;
                const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
                const float G = __builtin_amdgcn_exp2f(power * 1.4426950408889634f);
                // (the alpha threshold is the LAST test so that the wave vote below reads the compare's own lane mask)
                const bool pre = (int)ev & (int)(__float_as_uint(c.z) <= last_contributor) & (int)(power <= 0.0f);   // no short circuit: a branch here drags an LDS read behind the exp
                const float al0 = pre ? fminf(0.99f, b.y * G) : 0.f;
                const bool act = al0 >= 1.0f / 255.0f;
#ifdef AG_BWD_STATS
                {
                    const unsigned long long am = __ballot(act);
                    unsigned long long em = am | (am >> 32); em |= em >> 16;            // fold the 4 rows: bit (e * 4 + x)
                    int ne = 0;
                    for (int q = 0; q < 4; q++) ne += ((em >> (4 * q)) & 0xfull) ? 1 : 0;
                    ST(ST_STEPS, 1); ST(ST_SKIPPED, am == 0ull); ST(ST_ACTIVE_PAIRS, __popcll(am)); ST(ST_ACTIVE_ENTRIES, ne);
                    ST(ST_STEPS_BY_ACTIVE_ENTRIES + ne, 1);
                }
#endif
                if (__ballot(act) == 0ull) continue;
                const float al = act ? al0 : 0.f;
                const float fac = 1.0f - al;
                // dL/dalpha_e needs the blended state behind entry e only through its dot product with the pixel's gradient
                // g = (gr, gg, gb, gd, ga) (backward.cu:560-585 keeps five accum_rec channels and multiplies each by its dL_dchannel; g is
                // the same for every entry of a pixel, so the five recurrences are one): affine maps m_e(S) = fac_e S + al_e w_e with
                // w_e = g . (r, g, b, depth, 1)_e over the scalar S = g . state.  S lives in BANK 0 only (0 in the other banks) and
                // is folded into entry 0's map before the scan, so the inclusive scan yields the state behind each entry directly.
                const float w = fmaf(b.z, gr, fmaf(b.w, gg, fmaf(c.x, gb, fmaf(c.y, gd, ga))));
                float A = fac;
                float B = fmaf(fac, S, al * w);
#define AG_SCAN_STEP(N)                                                                                                        \
                asm volatile("s_nop 1\n\t"                                                                                     \
                             "v_fmac_f32_dpp %0, %0, %1 row_shr:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"             \
                             "v_mul_f32_dpp %1, %1, %1 row_shr:" #N " row_mask:0xf bank_mask:0xf"                               \
                             : "+v"(B), "+v"(A));
                AG_SCAN_STEP(4) AG_SCAN_STEP(8)
#undef AG_SCAN_STEP
                // A = prod_{i<=e} fac_i over the step; P = the same product over all earlier steps (carried in every lane of the pixel);
                // T in front of entry e = T_final / (P A).  The reference divides T by (1 - alpha) once per entry (backward.cu:534); here
                // the chain carried from step to step is the PRODUCT (multiplications only, as in the forward) and every T is one
                // v_rcp_f32 (1 ulp) + one multiplication away from it, so reciprocal errors do not accumulate along the list.
                const float Pe = P * A;
                const float Tin = T_final * __builtin_amdgcn_rcpf(Pe);
                const float beh = dpp_banks<AG_DPP_ROW_SHR(4), 0xf>(S, B);

                float v[10];
                {
                    float dL_dopa = w - beh;
                    dL_dopa = fmaf(dL_dopa, Tin, __builtin_amdgcn_rcpf(fac) * ntf_bg);
                    // q = G dL/dalpha; the geometric gradients of the entry are q times monomials of (dx, dy) times constants of
                    // the Gaussian (opacity, conic, viewport scale), which the preprocess backward applies once per Gaussian
                    const float q = act ? G * dL_dopa : 0.f;
                    const float wgt = al * Tin;
                    const float qdx = q * dx, qdy = q * dy;
                    v[A_QDX] = qdx;
                    v[A_QDY] = qdy;
                    v[A_QXX] = qdx * dx;
                    v[A_QXY] = qdx * dy;
                    v[A_QYY] = qdy * dy;
                    v[A_Q] = q;
                    v[A_COLR] = wgt * gr;
                    v[A_COLG] = wgt * gg;
                    v[A_COLB] = wgt * gb;
                    v[A_DEPTH] = wgt * gd;
                }
                // carry: the product up to entry 3 to all four lanes of the pixel; the state behind entry 3 to bank 0 (zero elsewhere)
                P = from_entry3(Pe);
                // (one v_mul_f32_dpp: bank 3's value rotated into bank 0, times the lane constant 1 in bank 0 / 0 elsewhere)
                // s_nop 1: B was written by the inline-asm scan above, which the compiler's hazard recogniser does not see -- a VALU write needs two
                // wait states before a DPP read of the same register (as the other DPP blocks of this kernel do)
                asm volatile("s_nop 1\n\t"
                             "v_mul_f32_dpp %0, %1, %2 row_ror:4 row_mask:0xf bank_mask:0xf" : "=&v"(S) : "v"(B), "v"(bank0));

#if defined(AG_BWD_KO) && AG_BWD_KO == 2      /* timing probe (wrong results): the step without the sums over the pixels and without the window / flush */
                if (v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7] + v[8] + v[9] == 12345.678f) s_out[lane] = P + S;
                continue;
#endif
                // sum over the block's 16 pixels per entry: rows (y) by register exchange 10 -> 5 -> 3, then x inside the quads
                float s[6];
#pragma unroll
                for (int i = 0; i < 5; i++) {
                    swap32(v[i], v[i + 5]);
                    s[i] = v[i] + v[i + 5];
                }
                s[5] = 0.f;
                float u[3];
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    swap16(s[i], s[i + 3]);
                    u[i] = s[i] + s[i + 3];      // row 0: value i, row 1: value i+3, row 2: value i+5, row 3: value i+8 (odd rows: u[2] = 0)
                }
                // x inside the quads: two butterflies as DPP operands of the additions (spelled out: the compiler sinks the second
                // addition into the store's exec region and leaves a v_mov_b32_dpp per value behind)
                asm volatile("s_nop 1\n\t"
                             "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                             "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                             "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                             : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]));
                // window: [entry][row][4]; lanes of entries past the end of the ring hold exact zeros (act is false there)
                if (qx == 0) {
                    float* dst = s_out + (win + e) * 16 + ry * 4;
                    dst[0] = u[0]; dst[1] = u[1]; dst[2] = u[2];
                    if (ry == 0) s_wgid[win + e] = __float_as_uint(c.w);
                }
                win += 4;
                if (win == kWin) flush();
            }
        }
        if (win) flush();
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Round 6 EXPERIMENT, not in the product build (bash profiles/ub/build_variant.sh pixel ag_blend_backward -DAG_BWD_PIXEL_KERNEL, then
// AG_LIB_PATH=.../libag_pixel.so): the pixel-lane kernel.  MEASURED (profiles/r06_bwd_pixel.md): correct (all raster GPU tests pass), 45 % of its
// consume lanes active against the wave kernel's 31 %, 16.2 M VALU instructions against 21.3 M -- and 356 us against 77.8 us, for two reasons
// that are properties of the hardware, not of this code:
//   * ds_add_f32 is a SERIAL unit on gfx950: 194 cycles per wave instruction whatever the address pattern (3 cycles per lane;
//     profiles/ub/lds_atomic_rate.hip, profiles/r06_lds_atomic_rate.txt: ds_add_u32 5.9 cycles) -- 39 M lane-adds per view = 457 k LDS cycles per CU.
//     Any design whose lanes hold DIFFERENT splats needs such adds (or ten global atomics per pair); that includes the tile-level LDS merge
//     of the wave kernel's block sums (3 ds_add_f32 x 16 lanes per step = 144 LDS cycles per step, 57 us per view of LDS time alone);
//   * with plain stores in place of the adds (wrong sums, timing only) it still takes 169 us: 2221 regions are 2.2 waves per SIMD, each a
//     dependent chain LDS read -> exp -> rcp -> ... per trip, so the SIMDs idle (VALU 30 % busy) -- the bench view has 142 k covered pixels,
//     one lane per pixel is 2.2 k waves; the wave kernel spends 4 lanes per pixel on the scan and gets 8.3 k.
// Kept as the record of why the sums over the pixels stay register reductions over entry-aligned lanes.
#ifdef AG_BWD_PIXEL_KERNEL
// The pixel-lane kernel.  The wave kernel above keeps the 4 entries of a step aligned across its 16 pixels so that the sums over
// the pixels are register exchanges -- and pays for it with 31 % active (pixel, entry) lanes: a splat of the bench scene covers ~5 of a
// block's 16 pixels (profiles/r05b_bwd_step_stats_tight.txt).  Here a wave owns an 8 x 8-pixel REGION (a tile quadrant), lane = pixel, and every
// lane walks ITS OWN list of covering splats at its own pace, so a lane slot is wasted only while its pixel has less to do than the busiest
// pixel of the region (55 % active on the bench view, CPU simulation of the schedule: profiles/r06_bwd_pixel_sim.txt), and a pair costs the plain
// serial recurrences of the reference (no scans, no cross-lane sums: ~50 VALU per 64 lanes against 83).  The price: different lanes hold
// different splats, so the sums over the pixels are LDS float atomics (ds_add_f32, 10 per pair) into a per-slot accumulator that leaves
// as line-coalesced global atomics once per (region, splat) -- 4 blocks' worth of the wave kernel's line requests in one.
//
//   production (all 64 lanes = list entries): the region's exact quadratic-form cull, 64 entries per pass, records one pass ahead in
//     registers; survivors are committed in walking order to a ring of kPxSlots records as space frees up (a culled pass waits in registers);
//   sealing (per batch of 32 ring slots; lanes = 32 splats x 2 region halves): the splat's pixel coverage, row by row, from the roots of
//     q(x) = qcut on that row (a superset of alpha >= 1/255: qcut carries the preprocess's slack), then a 32 x 32 bit-matrix transpose inside
//     each half-wave (5 butterfly steps) turns "pixels of a splat" into "splats of a pixel": one 32-bit word per lane and batch;
//   consumption (lane = pixel): take the lowest set bit of the word, read that record, do the reference's serial step (backward.cu:494-600,
//     T by the running product as in the wave kernel), add the ten terms to the slot's accumulators;
//   retirement: a batch all lanes are through is flushed (16 lanes = one 64-byte accumulator line) and its 32 slots return to the ring.
// Nothing waits: single-wave workgroups, LDS traffic of one wave is in order.
constexpr int kReg = 8;                                   // 8 x 8 pixels per wave
constexpr int kRegsPerTile = (kTileX / kReg) * (kTileY / kReg);
constexpr int kPxBatch = 32;
#ifndef AG_BWD_PX_NB
#define AG_BWD_PX_NB 4
#endif
constexpr int kPxNB = AG_BWD_PX_NB;                       // batches in the ring
constexpr int kPxSlots = kPxBatch * kPxNB;
#ifndef AG_BWD_PX_GRID
#define AG_BWD_PX_GRID 16384
#endif
constexpr int kPxGrid = AG_BWD_PX_GRID;
constexpr int kPxAccStride = 11;                          // odd: a slot's ten sums start on any of the 32 banks
static_assert((kPxSlots & (kPxSlots - 1)) == 0, "ring size must be a power of two");
#ifndef AG_BWD_PX_OCC
#define AG_BWD_PX_OCC 4
#endif

struct RegionItemIter {
    uint32_t i, stride, x, n_active;
    __device__ __forceinline__ RegionItemIter(uint32_t block, uint32_t grid, uint32_t n_active_)
        : i(block / kQueues), stride((grid + kQueues - 1 - (block % kQueues)) / kQueues), x(block % kQueues), n_active(n_active_) {}
    __device__ __forceinline__ bool next(uint32_t& tile_rank, uint32_t& rg)
    {
        tile_rank = (i / kRegsPerTile) * kQueues + x;        // the 4 regions of a tile on one XCD (they gather the same list)
        rg = i % kRegsPerTile;
        i += stride;
        return tile_rank < n_active;
    }
};

#ifdef AG_BWD_STATS
enum { PX_ITEMS = 0, PX_WALKED, PX_SURVIVORS, PX_ITERS, PX_ACTIVE_PAIRS, PX_MASK_BITS, PX_TRIPS, PX_BATCHES, PX_N = 8 };
__device__ unsigned long long g_bwd_px_stats[PX_N];
#define PST(k, v) do { if (lane == 0) atomicAdd(&g_bwd_px_stats[k], (unsigned long long)(v)); } while (0)
extern "C" int ag_debug_bwd_px_stats(unsigned long long* out)
{
    if (hipDeviceSynchronize() != hipSuccess) return AG_ERR_HIP;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bwd_px_stats), sizeof(g_bwd_px_stats)) != hipSuccess) return AG_ERR_HIP;
    unsigned long long zero[PX_N] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_px_stats), zero, sizeof(zero)) != hipSuccess) return AG_ERR_HIP;
    return AG_OK;
}
#else
#define PST(k, v) do { } while (0)
#endif

__global__ void __launch_bounds__(64, AG_BWD_PX_OCC) blend_backward_pixel_kernel(BlendBwdParams p)
{
    __shared__ float4 s_rec[kPxSlots * 3];        // [x, y, ca, cb] [cc, op, r, g] [b, depth, position, gaussian id]
    __shared__ float s_qc[kPxSlots];              // the splat's threshold on q (GaussRec::qcut)
    __shared__ float s_acc[kPxSlots * kPxAccStride];
    __shared__ uint32_t s_mask[kPxNB * 64];       // [batch of the ring][pixel]: which of the batch's 32 splats may reach the pixel

    const int lane = threadIdx.x;
    const int rx = lane & 7, ry = lane >> 3;
    const int mj = lane & 31, mh = lane >> 5;     // sealing: splat of the batch, half of the region (rows 4 mh .. 4 mh + 3)
    const int f_ent = lane >> 4, f_comp = lane & 15;
    const uint32_t n_active = p.counts[1];
    const size_t HW = (size_t)p.W * p.H;
    const float bg0 = p.bg[0], bg1 = p.bg[1], bg2 = p.bg[2];
    for (int i = lane; i < kPxSlots * kPxAccStride; i += 64) s_acc[i] = 0.f;

    RegionItemIter it(blockIdx.x, gridDim.x, n_active);
    uint32_t tr, rg;
    while (it.next(tr, rg)) {
        const uint4 hdr = p.tile_order[tr];
        const int tile = (int)hdr.x;
        const uint32_t rbeg = hdr.y;
        const int tile_x = tile % p.gx, tile_y = tile / p.gx;
        const int bx0 = tile_x * kTileX + (int)(rg & 1) * kReg, by0 = tile_y * kTileY + (int)(rg >> 1) * kReg;
        const int px = bx0 + rx, py = by0 + ry;
        const bool inside = px < p.W && py < p.H;
        const float pxf = (float)px, pyf = (float)py;
        const float qx0f = (float)bx0, qy0f = (float)by0, qx1f = (float)(bx0 + kReg - 1), qy1f = (float)(by0 + kReg - 1);
        const int pix = p.W * py + px;
        uint32_t last_contributor = 0;
        float T_final = 0.f, gr = 0.f, gg = 0.f, gb = 0.f, gd = 0.f, ga = 0.f;
        if (inside) {
            last_contributor = p.n_contrib[pix];
            T_final = 1.0f - p.alphas[pix];
            gr = p.dL_dpix[pix];
            gg = p.dL_dpix[HW + pix];
            gb = p.dL_dpix[2 * HW + pix];
            gd = p.dL_ddepth[pix];
            ga = p.dL_dalpha[pix];
        }
        const float ntf_bg = -T_final * (bg0 * gr + bg1 * gg + bg2 * gb);
        const uint32_t wmax = wave_umax(last_contributor);
        if (wmax == 0) continue;
        const uint32_t rend = rbeg + wmax;                   // walk [rbeg, rend) from the back
        PST(PX_ITEMS, 1); PST(PX_WALKED, wmax);

        // ---- production state.  n*: records of the next pass (loads in flight), r*: the culled pass waiting to be committed ----
        uint32_t id_n = p.point_list[(uint32_t)lane < wmax ? rend - 1u - (uint32_t)lane : rbeg];
        uint32_t id_nn = p.point_list[(uint32_t)lane + 64u < wmax ? rend - 1u - ((uint32_t)lane + 64u) : rbeg];
        // (named scalars, not float4 variables: carried around the loop as vectors the compiler parks single components in scratch memory)
#define AG_PX_LOAD_NEXT(ID)                                                                 \
        {                                                                                   \
            const float4* src = reinterpret_cast<const float4*>(p.rec + (ID));              \
            const float4 t0 = src[0], t1 = src[1], t2 = src[2];                             \
            n_x = t0.x; n_y = t0.y; n_ca = t0.z; n_cb = t0.w;                               \
            n_cc = t1.x; n_op = t1.y; n_r = t1.z; n_g = t1.w;                               \
            n_b = t2.x; n_d = t2.y; n_qc = t2.w;                                            \
        }
        float n_x, n_y, n_ca, n_cb, n_cc, n_op, n_r, n_g, n_b, n_d, n_qc;
        AG_PX_LOAD_NEXT(id_n)
        float r_x = 0.f, r_y = 0.f, r_ca = 0.f, r_cb = 0.f, r_cc = 0.f, r_op = 0.f, r_r = 0.f, r_g = 0.f, r_b = 0.f, r_d = 0.f, r_qc = 0.f;
        uint32_t id_r = 0, pos_r = 0;
        bool keep = false;
        uint32_t rank = 0;
        uint32_t started = 0;        // list entries whose pass has been culled (multiple of 64)
        uint32_t ptotal = 0, pc = 0; // survivors of the pending pass, committed so far
        uint32_t cnt = 0;            // ring positions committed
        uint32_t sealed = 0;         // batches with their pixel words built
        uint32_t tail = 0;           // batches retired
        // ---- consumption state (per pixel) ----
        float P = 1.0f;              // prod (1 - alpha) over the pixel's entries done so far
        float S = 0.f;               // g . (blended state behind the entries done so far)
        int cur = -1;                // batch of `word`
        uint32_t word = 0;           // splats of batch `cur` still to do for this pixel

        for (;;) {
            // ================= production =================
            if (pc == ptotal && started < wmax) {
                // cull the next pass against the region (the exact rectangle minimum of the wave kernel, on 8 x 8 pixels)
                r_x = n_x; r_y = n_y; r_ca = n_ca; r_cb = n_cb; r_cc = n_cc; r_op = n_op; r_r = n_r; r_g = n_g; r_b = n_b; r_d = n_d; r_qc = n_qc;
                id_r = id_n;
                const uint32_t o = started + (uint32_t)lane;     // offset from the back
                pos_r = wmax - o;
                const float U0 = qx0f - r_x, U1 = qx1f - r_x, V0 = qy0f - r_y, V1 = qy1f - r_y;
                const float ue = __builtin_amdgcn_fmed3f(0.f, U0, U1), ve = __builtin_amdgcn_fmed3f(0.f, V0, V1);
                const float ca = r_ca, cb = r_cb, cc = r_cc, qc = r_qc;
                const float det = fmaf(ca, cc, -cb * cb);
                const float bue = cb * ue, bve = cb * ve;
                const float w1 = __builtin_amdgcn_fmed3f(0.f, fmaf(cc, V0, bue), fmaf(cc, V1, bue));
                const float w2 = __builtin_amdgcn_fmed3f(0.f, fmaf(ca, U0, bve), fmaf(ca, U1, bve));
                const float l1 = fmaf(w1, w1, det * ue * ue), l2 = fmaf(w2, w2, det * ve * ve);
                keep = (o < wmax) && ((l1 <= cc * qc) | (l2 <= ca * qc));
                const unsigned long long km = __ballot(keep);
                rank = (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
                ptotal = (uint32_t)__popcll(km);
                pc = 0;
                started += 64u;
                // the pass after: records now, the indices of the one after that
                id_n = id_nn;
                AG_PX_LOAD_NEXT(id_nn)
                const uint32_t onn = started + 64u + (uint32_t)lane;
                id_nn = p.point_list[onn < wmax ? rend - 1u - onn : rbeg];
            }
            if (pc < ptotal) {
                const uint32_t room = (uint32_t)kPxSlots - (cnt - tail * (uint32_t)kPxBatch);
                const uint32_t n = min(room, ptotal - pc);
                if (n) {
                    const uint32_t rel = rank - pc;          // wraps for survivors committed earlier
                    if (keep && rel < n) {
                        const uint32_t slot = (cnt + rel) & (uint32_t)(kPxSlots - 1);
                        s_rec[slot * 3 + 0] = make_float4(r_x, r_y, r_ca, r_cb);
                        s_rec[slot * 3 + 1] = make_float4(r_cc, r_op, r_r, r_g);
                        s_rec[slot * 3 + 2] = make_float4(r_b, r_d, __uint_as_float(pos_r), __uint_as_float(id_r));
                        s_qc[slot] = r_qc;
                    }
                    cnt += n; pc += n;
                    PST(PX_SURVIVORS, n);
                }
            }
            const bool fed = (started >= wmax) && (pc == ptotal);        // nothing more will enter the ring
            while ((sealed + 1u) * (uint32_t)kPxBatch <= cnt || (fed && sealed * (uint32_t)kPxBatch < cnt)) {
                // ---- seal batch `sealed`: coverage of splat mj on rows 4 mh .. 4 mh + 3 of the region, transposed to splats per pixel ----
                const uint32_t nb = min((uint32_t)kPxBatch, cnt - sealed * (uint32_t)kPxBatch);
                const uint32_t slot = (sealed * (uint32_t)kPxBatch + (uint32_t)mj) & (uint32_t)(kPxSlots - 1);
                const float4 a = s_rec[slot * 3 + 0];
                const float2 co = *reinterpret_cast<const float2*>(&s_rec[slot * 3 + 1]);
                const float qc = s_qc[slot];
                const float ca = a.z, cb = a.w, cc = co.x;
                const float det = fmaf(ca, cc, -cb * cb);
                const float inva = __builtin_amdgcn_rcpf(ca);
                const float aq = ca * qc;
                const float cxr = a.x - qx0f;                // splat centre relative to the region's first column
                const bool pd = (ca > 0.f) & (det > 0.f) & (qc < 1.0e37f);
                uint32_t w = 0;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float dy = a.y - (qy0f + (float)(mh * 4 + r));
                    const float t = det * dy;
                    const float disc = fmaf(-t, dy, aq);     // a qcut - det dy^2: a times the discriminant quarter of q(dx) = qcut on this row
                    const float sq = __builtin_amdgcn_sqrtf(fmaxf(disc, 0.f));
                    const float bdy = cb * dy;
                    const float lo = fmaf(bdy - sq, inva, cxr) - 2e-3f, hi = fmaf(bdy + sq, inva, cxr) + 2e-3f;
                    // columns k with lo < k <= hi (a superset of the exact set by the 2e-3 px margins)
                    int xl = (int)floorf(lo) + 1, xh = (int)floorf(hi);
                    xl = max(xl, 0); xh = min(xh, kReg - 1);
                    uint32_t bits = (disc >= 0.f && xl <= xh) ? ((2u << xh) - (1u << xl)) : 0u;
                    if (!(lo <= hi) || !pd) bits = 0xffu;     // NaN / not positive definite / never-culled splat: whole row
                    w |= bits << (8 * r);
                }
                if ((uint32_t)mj >= nb) w = 0;
                // 32 x 32 bit-matrix transpose inside each half-wave: lane (splat j) bit (pixel i)  ->  lane (pixel i) bit (splat j)
#define AG_TR_STEP(SH, MASK)                                                                               \
                {                                                                                          \
                    const uint32_t o_ = (uint32_t)__shfl_xor((int)w, SH, 64);                              \
                    w = (mj & SH) ? ((w & MASK) | ((o_ & MASK) >> SH)) : ((w & ~MASK) | ((o_ & ~MASK) << SH)); \
                }
                AG_TR_STEP(16, 0xffff0000u) AG_TR_STEP(8, 0xff00ff00u) AG_TR_STEP(4, 0xf0f0f0f0u) AG_TR_STEP(2, 0xccccccccu) AG_TR_STEP(1, 0xaaaaaaaau)
#undef AG_TR_STEP
                // the batch's frontmost splat is its last: a pixel whose list ends in front of it... (position > n_contrib) has nothing here
                const uint32_t fslot = (sealed * (uint32_t)kPxBatch + nb - 1u) & (uint32_t)(kPxSlots - 1);
                const uint32_t fpos = __float_as_uint(s_rec[fslot * 3 + 2].z);
                if (fpos > last_contributor) w = 0;
                s_mask[(sealed % (uint32_t)kPxNB) * 64u + (uint32_t)lane] = w;
                PST(PX_BATCHES, 1);
#ifdef AG_BWD_STATS
                { unsigned long long tot = 0; for (int l = 0; l < 64; l++) tot += __popc(__shfl((int)w, l, 64)); PST(PX_MASK_BITS, tot); }
#endif
                sealed++;
            }

            // ================= consumption: one entry per pixel =================
            while (word == 0u && cur + 1 < (int)sealed) {
                cur++;
                word = s_mask[((uint32_t)cur % (uint32_t)kPxNB) * 64u + (uint32_t)lane];
            }
            const bool has = word != 0u;
            const unsigned long long hm = __ballot(has);
            PST(PX_TRIPS, 1);
            if (hm != 0ull) {
                const uint32_t j = has ? (uint32_t)__builtin_ctz(word) : 0u;
                word &= word - 1u;
                const uint32_t slot = has ? (((uint32_t)cur * (uint32_t)kPxBatch + j) & (uint32_t)(kPxSlots - 1)) : 0u;
                const float4 a = s_rec[slot * 3 + 0];   // x, y, conic a, conic b
                const float4 b = s_rec[slot * 3 + 1];   // conic c, opacity, r, g
                const float4 c = s_rec[slot * 3 + 2];   // b, depth, position, id
                const float dx = a.x - pxf, dy = a.y - pyf;
                const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
                const float G = __builtin_amdgcn_exp2f(power * 1.4426950408889634f);
                const bool pre = (int)has & (int)(__float_as_uint(c.z) <= last_contributor) & (int)(power <= 0.0f);
                const float al0 = pre ? fminf(0.99f, b.y * G) : 0.f;
                const bool act = al0 >= 1.0f / 255.0f;
                const float al = act ? al0 : 0.f;
                const float fac = 1.0f - al;
                const float w = fmaf(b.z, gr, fmaf(b.w, gg, fmaf(c.x, gb, fmaf(c.y, gd, ga))));
                P *= fac;                                                   // prod (1 - alpha) up to and including this entry
                const float Tin = T_final * __builtin_amdgcn_rcpf(P);       // transmittance in front of the entry
                float dL_dopa = w - S;
                dL_dopa = fmaf(dL_dopa, Tin, __builtin_amdgcn_rcpf(fac) * ntf_bg);
                S = fmaf(fac, S, al * w);
                const float q = G * dL_dopa;
                const float wgt = al * Tin;
                const float qdx = q * dx, qdy = q * dy;
                { const unsigned long long am_ = __ballot(act); PST(PX_ITERS, 1); PST(PX_ACTIVE_PAIRS, __popcll(am_)); }
#ifndef AG_PX_ATOMIC
#define AG_PX_ATOMIC 1
#endif
                if (act) {
                    float* dst = s_acc + slot * kPxAccStride;
#if AG_PX_ATOMIC == 1
#define AG_PX_ADD(K, V) atomicAdd(dst + (K), (V))
#elif AG_PX_ATOMIC == 2      /* timing probe only: integer atomics (wrong sums) */
#define AG_PX_ADD(K, V) atomicAdd(reinterpret_cast<uint32_t*>(dst + (K)), __float_as_uint(V))
#else                        /* timing probe only: plain stores (wrong sums) */
#define AG_PX_ADD(K, V) dst[K] = (V)
#endif
                    AG_PX_ADD(A_QDX, qdx);
                    AG_PX_ADD(A_QDY, qdy);
                    AG_PX_ADD(A_QXX, qdx * dx);
                    AG_PX_ADD(A_QXY, qdx * dy);
                    AG_PX_ADD(A_QYY, qdy * dy);
                    AG_PX_ADD(A_Q, q);
                    AG_PX_ADD(A_COLR, wgt * gr);
                    AG_PX_ADD(A_COLG, wgt * gg);
                    AG_PX_ADD(A_COLB, wgt * gb);
                    AG_PX_ADD(A_DEPTH, wgt * gd);
#undef AG_PX_ADD
                }
            }

            // ================= retirement =================
            if (tail < sealed) {
                const bool busy = (cur < (int)tail) | ((cur == (int)tail) & (word != 0u));
                if (__ballot(busy) == 0ull) {
                    const uint32_t nb = min((uint32_t)kPxBatch, cnt - tail * (uint32_t)kPxBatch);
                    const uint32_t base = (tail * (uint32_t)kPxBatch) & (uint32_t)(kPxSlots - 1);
                    float val[kPxBatch / 4];
                    uint32_t gid[kPxBatch / 4];
#pragma unroll
                    for (int pass = 0; pass < kPxBatch / 4; pass++) {
                        const uint32_t slot = base + (uint32_t)(pass * 4 + f_ent);
                        val[pass] = s_acc[slot * kPxAccStride + (f_comp < 10 ? f_comp : 10)];
                        gid[pass] = __float_as_uint(s_rec[slot * 3 + 2].w);
                    }
#pragma unroll
                    for (int pass = 0; pass < kPxBatch / 4; pass++) {
                        const uint32_t ent = (uint32_t)(pass * 4 + f_ent);
                        if (ent < nb && f_comp < 10 && val[pass] != 0.f) atomicAdd(p.accum + (size_t)gid[pass] * kAccumFloats + f_comp, val[pass]);
                    }
                    for (int i = lane; i < kPxBatch * kPxAccStride; i += 64) s_acc[base * kPxAccStride + i] = 0.f;
                    tail++;
                }
            } else if (fed && hm == 0ull) {
                break;        // everything walked, committed, sealed, consumed and retired
            }
        }
    }
}

#endif  // AG_BWD_PIXEL_KERNEL

// Calibration (profiles/atomic_rate.py): how many line-coalesced float atomics per second the memory side sustains -- the
// ceiling of every design that flushes the backward's sums with less pre-reduction.  Each wave instruction adds to the first
// `comps` slots of 4 pseudo-random 64-byte accumulator lines (16 adjacent lanes per line), exactly the flush's access shape.
__global__ void __launch_bounds__(512) debug_atomic_rate_kernel(float* __restrict__ accum, uint32_t lines, int iters, int comps)
{
    const uint32_t lane = threadIdx.x & 63, wave_global = (blockIdx.x * 512 + threadIdx.x) >> 6;
    uint32_t h = wave_global * 0x9E3779B9u + 12345u;
    for (int i = 0; i < iters; i++) {
        h = h * 1664525u + 1013904223u;
        const uint32_t line = ((h >> 8) + (lane >> 4) * 977u) % lines;
        if ((int)(lane & 15) < comps) atomicAdd(accum + (size_t)line * kAccumFloats + (lane & 15), 1.0f);
    }
}

int launch_debug_atomic_rate(float* accum, int lines, int blocks, int iters, int comps, hipStream_t s)
{
    hipLaunchKernelGGL(debug_atomic_rate_kernel, dim3(blocks), dim3(512), 0, s, accum, (uint32_t)lines, iters, comps);
    return check_hip(hipGetLastError(), "debug_atomic_rate_kernel");
}

int launch_blend_backward(const AgRasterBackwardArgs& a, hipStream_t s)
{
    BlendBwdParams p;
    p.W = a.W; p.H = a.H;
    p.gx = (a.W + kTileX - 1) / kTileX;
    const int gy = (a.H + kTileY - 1) / kTileY;
    p.T = p.gx * gy;
    const char* gb = aligned_base(a.geom_buffer);
    const char* ib = aligned_base(a.image_buffer);
    GeomLayout gl((size_t)a.P);
    ImageLayout il((size_t)a.W, (size_t)a.H);
    BinLayout bl((size_t)a.num_rendered);
    p.tile_order = reinterpret_cast<const uint4*>(ib + il.tile_order);
    p.counts = reinterpret_cast<const uint32_t*>(ib + il.num_rendered);
    p.rec = reinterpret_cast<const GaussRec*>(gb + gl.rec);
    p.point_list = a.num_rendered > 0
        ? reinterpret_cast<const uint32_t*>(aligned_base(a.binning_buffer) + bl.point_list) : nullptr;
    p.bg = a.bg;
    p.alphas = a.alphas;
    p.n_contrib = reinterpret_cast<const uint32_t*>(ib + il.n_contrib);
    p.dL_dpix = a.dL_dout_color; p.dL_ddepth = a.dL_dout_depth; p.dL_dalpha = a.dL_dout_alpha;
    p.accum = reinterpret_cast<float*>(aligned_base(a.accum_buffer));
    if (check_hip(hipMemsetAsync(p.accum, 0, (size_t)a.P * kAccumFloats * sizeof(float), s), "memset accum")) return AG_ERR_HIP;
    if (a.num_rendered <= 0) return AG_OK;
    const long long items = (long long)p.T * kBlocksPerTile;
    const int grid = (int)(items < kWaveGrid ? items : kWaveGrid);
#ifdef AG_BWD_PIXEL_KERNEL
    static const bool use_wave = [] { const char* e = getenv("AG_BWD_KERNEL"); return e && e[0] == 'w'; }();
    if (!use_wave) {
        const long long ritems = (long long)p.T * kRegsPerTile;
        const int rgrid = (int)(ritems < kPxGrid ? ritems : kPxGrid);
        { ProfScope ps(AG_K_BLEND_BACKWARD, s); hipLaunchKernelGGL(blend_backward_pixel_kernel, dim3(rgrid), dim3(64), 0, s, p); }
        return check_hip(hipGetLastError(), "blend_backward_pixel_kernel");
    }
#endif
    { ProfScope ps(AG_K_BLEND_BACKWARD, s); hipLaunchKernelGGL(blend_backward_wave_kernel, dim3(grid), dim3(64), 0, s, p); }
    return check_hip(hipGetLastError(), "blend_backward_wave_kernel");
}

}  // namespace ag
