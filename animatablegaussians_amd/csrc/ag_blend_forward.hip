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
// MI355X mapping (not the reference's 256-thread lock-step block):
//   * one workgroup per 16x16 tile, but its four wave64s are INDEPENDENT: wave w owns the 8x8 pixel quad
//     (w&1, w>>1) and walks the tile's sorted list on its own, 64 entries at a time (lane = list entry);
//   * while staging, each lane tests its entry against the quad with the record's conservative cut-off radius
//     (r2cut, see ag_preprocess.hip) and the survivors are ballot-compacted into a wave-private LDS slab, so the
//     inner loop only visits splats that can reach alpha >= 1/255 somewhere in the quad (a radius-4 splat touches
//     ~1.7 of a tile's 4 quads).  Culled entries still count in n_contrib because the list position travels with
//     the record;
//   * the inner loop reads one record per iteration as three uniform-address ds_read_b128 (LDS broadcast), no
//     workgroup barrier anywhere; a wave leaves as soon as all of its 64 pixels are saturated.
// Bound: VALU (about 30 ops per (splat, quad) pair) + LDS broadcast; HBM traffic is the compulsory
// 52 B/instance + 24 B/pixel.
#include "ag_common.h"

namespace ag {

struct BlendFwdParams {
    int W, H, gx, T;
    const uint2* __restrict__ ranges;
    const uint32_t* __restrict__ point_list;
    const GaussRec* __restrict__ rec;
    const float* __restrict__ bg;
    float* __restrict__ out_color;
    float* __restrict__ out_depth;
    float* __restrict__ out_alpha;
    uint32_t* __restrict__ n_contrib;
};

__global__ void __launch_bounds__(256) blend_forward_kernel(BlendFwdParams p)
{
    __shared__ float4 slab[4][68 * 3];

    const int tile = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile_x = tile % p.gx, tile_y = tile / p.gx;
    const int qx0 = tile_x * kTileX + (wave & 1) * 8, qy0 = tile_y * kTileY + (wave >> 1) * 8;
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool inside = px < p.W && py < p.H;
    const float pxf = (float)px, pyf = (float)py;
    const float qx0f = (float)qx0, qy0f = (float)qy0, qx1f = (float)(qx0 + 7), qy1f = (float)(qy0 + 7);
    const uint2 range = p.ranges[tile];
    float4* my = slab[wave];

    float T = 1.0f, Cr = 0.f, Cg = 0.f, Cb = 0.f, Dd = 0.f, Wsum = 0.f;
    uint32_t last = 0;
    bool done = !inside;

    // Software pipeline over 64-entry batches: while batch b is culled/compacted and blended, the (dependent)
    // point_list -> record gathers of batch b+1 are already in flight; hipcc waits for them at their first use,
    // i.e. at the top of the next iteration.  With ~2.5 resident waves per SIMD nothing else would hide that latency.
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = make_float4(0.f, 0.f, -1.f, 0.f);
    if (range.x + lane < range.y) {
        const float4* src = reinterpret_cast<const float4*>(p.rec + p.point_list[range.x + lane]);
        r0 = src[0]; r1 = src[1]; r2 = src[2];
    }
    for (uint32_t base = range.x; base < range.y; base += 64) {
        if (__all(done)) break;
        const uint32_t k = base + lane;
        // cull against this wave's 8x8 quad (r2.z = r2cut; lanes past the end carry r2cut = -1)
        const float ddx = fmaxf(fmaxf(qx0f - r0.x, r0.x - qx1f), 0.f);
        const float ddy = fmaxf(fmaxf(qy0f - r0.y, r0.y - qy1f), 0.f);
        const bool keep = (k < range.y) && ((ddx * ddx + ddy * ddy) <= r2.z);
        const unsigned long long mask = __ballot(keep);
        const int slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        if (keep) {
            my[slot * 3 + 0] = r0;
            my[slot * 3 + 1] = r1;
            my[slot * 3 + 2] = make_float4(r2.x, r2.y, __uint_as_float(k - range.x + 1u), 0.f);  // 1-based list position
        }
        // prefetch the next batch
        const uint32_t kn = k + 64;
        if (kn < range.y) {
            const float4* src = reinterpret_cast<const float4*>(p.rec + p.point_list[kn]);
            r0 = src[0]; r1 = src[1]; r2 = src[2];
        }
        const int cnt = __popcll(mask);
        // pad the compacted batch to a multiple of 4 with inert records (opacity 0 -> alpha 0 -> never a candidate)
        if (lane < 4) {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            my[(cnt + lane) * 3 + 0] = z;
            my[(cnt + lane) * 3 + 1] = z;
            my[(cnt + lane) * 3 + 2] = z;
        }
        __builtin_amdgcn_wave_barrier();
        // Four entries per trip: their conic/exp evaluations are independent, which gives a wave that is alone on
        // its SIMD (the long-list tail of the kernel) instruction-level parallelism; only the short T recurrence is
        // serial.
        for (int j = 0; j < cnt; j += 4) {
            float4 a[4], b[4], c[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                a[u] = my[(j + u) * 3 + 0];
                b[u] = my[(j + u) * 3 + 1];
                c[u] = my[(j + u) * 3 + 2];
            }
            float power[4], alpha[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const float dx = a[u].x - pxf, dy = a[u].y - pyf;
                power[u] = -0.5f * (a[u].z * dx * dx + b[u].x * dy * dy) - a[u].w * dx * dy;
                alpha[u] = fminf(0.99f, b[u].y * __builtin_amdgcn_exp2f(power[u] * 1.4426950408889634f));
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const bool cand = !done && (power[u] <= 0.0f) && (alpha[u] >= 1.0f / 255.0f);
                const float test_T = T * (1.0f - alpha[u]);
                const bool stop = cand && (test_T < 0.0001f);
                const bool act = cand && !stop;
                done = done || stop;
                const float w = act ? alpha[u] * T : 0.f;
                Cr += b[u].z * w;
                Cg += b[u].w * w;
                Cb += c[u].x * w;
                Dd += c[u].y * w;
                Wsum += w;
                T = act ? test_T : T;
                last = act ? __float_as_uint(c[u].z) : last;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }

    if (inside) {
        const int pix = p.W * py + px;
        const size_t HW = (size_t)p.W * p.H;
        p.n_contrib[pix] = last;
        p.out_color[pix] = Cr + T * p.bg[0];
        p.out_color[HW + pix] = Cg + T * p.bg[1];
        p.out_color[2 * HW + pix] = Cb + T * p.bg[2];
        p.out_alpha[pix] = Wsum;
        p.out_depth[pix] = Dd;
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
    p.bg = a.bg;
    p.out_color = a.out_color; p.out_depth = a.out_depth; p.out_alpha = a.out_alpha;
    p.n_contrib = reinterpret_cast<uint32_t*>(ib + il.n_contrib);
    { ProfScope ps(AG_K_BLEND_FORWARD, s); hipLaunchKernelGGL(blend_forward_kernel, dim3(p.T), dim3(256), 0, s, p); }
    return check_hip(hipGetLastError(), "blend_forward_kernel");
}

}  // namespace ag
