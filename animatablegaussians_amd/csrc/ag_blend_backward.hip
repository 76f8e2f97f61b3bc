// Backward per-tile alpha blend: gradients w.r.t. 2D mean, conic, opacity, colour and depth (gfx950).
//
// Replaces renderCUDA<3> backward (reference cuda_rasterizer/backward.cu:415-601): back-to-front replay from
// T_final = 1 - alpha_out with T recovered by division, the depth and alpha-output gradient terms, the background
// term, no gradient gate at the 0.99 alpha clamp, dL_dmean2D in NDC-scaled units (x 0.5 W, x 0.5 H).
//
// The reference issues 10 global float atomicAdds per (pixel, Gaussian) pair.  Here (wave64, CDNA4):
//   * four independent waves per 16x16 tile, one 8x8 pixel quad each, same staging/cull/compaction as the forward
//     (ag_blend_forward.hip), but walking the tile list from the back and starting at the quad's largest
//     n_contrib, so the forward's early termination is inherited;
//   * for each surviving splat the 10 per-pixel partial gradients are summed over the 64 lanes by a TRANSPOSED
//     butterfly: v_permlane32_swap / v_permlane16_swap exchange register halves so that each step halves the number
//     of live values while doubling the lanes summed (16 -> 8 -> 4 -> 2 -> 1 registers), then two quad-perm adds.
//     35 VALU ops for all 10 sums instead of 70 for ten independent DPP reductions; afterwards lane l holds the
//     total of value (l >> 2) & 15;
//   * the per-splat totals are parked in the wave's LDS slab and flushed once per 64-entry batch with one lane per
//     splat: ten atomic instructions per batch, all landing in that splat's single 64-byte accumulator line.
// Global atomics drop from 10 per (pixel, splat) to 10 per (quad, splat).
#include "ag_common.h"

namespace ag {

struct BlendBwdParams {
    int W, H, gx, T;
    const uint2* __restrict__ ranges;
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

// Sum each of v[0..15] over the 64 lanes; on return lane l holds the total of v[(l >> 2) & 15].
__device__ __forceinline__ float wave_reduce16_transposed(float (&v)[16], int lane)
{
    float s[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        swap32(v[i], v[i + 8]);
        s[i] = v[i] + v[i + 8];  // lanes <32: value i, lanes >=32: value i+8
    }
    float u[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        swap16(s[i], s[i + 4]);
        u[i] = s[i] + s[i + 4];  // row r: value i + 4*(r&1) + 8*(r>>1)
    }
    float w[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const float lo = u[i] + dpp<AG_DPP_ROW_ROR(8)>(u[i]);
        const float hi = u[i + 2] + dpp<AG_DPP_ROW_ROR(8)>(u[i + 2]);
        w[i] = (lane & 8) ? hi : lo;
    }
    const float lo = w[0] + dpp<AG_DPP_ROW_SHL(4)>(w[0]);
    const float hi = w[1] + dpp<AG_DPP_ROW_SHR(4)>(w[1]);
    float x = (lane & 4) ? hi : lo;
    x += dpp<AG_DPP_QUAD_PERM(1, 0, 3, 2)>(x);
    x += dpp<AG_DPP_QUAD_PERM(2, 3, 0, 1)>(x);
    return x;
}

__global__ void __launch_bounds__(256) blend_backward_kernel(BlendBwdParams p)
{
    __shared__ float4 slab[4][64 * 3];
    __shared__ float gslab[4][16 * 65];

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
    float* myg = gslab[wave];

    const int pix = p.W * py + px;
    const size_t HW = (size_t)p.W * p.H;
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
    const float bg_dot = p.bg[0] * gr + p.bg[1] * gg + p.bg[2] * gb;
    const float ddelx_dx = 0.5f * (float)p.W, ddely_dy = 0.5f * (float)p.H;

    // largest n_contrib of the quad: nothing behind it contributed to any of these pixels
    uint32_t wmax = last_contributor;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, d, 64));
    if (wmax == 0) return;

    float T = T_final;
    float acc_r = 0.f, acc_g = 0.f, acc_b = 0.f, acc_d = 0.f, acc_a = 0.f;
    float last_alpha = 0.f, last_r = 0.f, last_g = 0.f, last_b = 0.f, last_d = 0.f;

    // entries [range.x, range.x + wmax) back to front, 64 per batch; lane 0 takes the rearmost entry of the batch.
    // Same software pipeline as the forward: the gathers of the next batch fly while this one is processed.
    constexpr int GS = 65;   // padded row stride of the per-batch gradient slab: [16 values][64 splats]
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
    uint32_t id = 0;
    {
        const uint32_t take0 = wmax < 64u ? wmax : 64u;
        if ((uint32_t)lane < take0) {
            id = p.point_list[range.x + wmax - 1u - (uint32_t)lane];
            const float4* src = reinterpret_cast<const float4*>(p.rec + id);
            r0 = src[0]; r1 = src[1]; r2 = src[2];
        }
    }
    for (uint32_t remaining = wmax; remaining > 0;) {
        const uint32_t take = remaining < 64u ? remaining : 64u;
        const uint32_t top = range.x + remaining;  // one past the rearmost entry of this batch
        remaining -= take;
        const float ddx = fmaxf(fmaxf(qx0f - r0.x, r0.x - qx1f), 0.f);
        const float ddy = fmaxf(fmaxf(qy0f - r0.y, r0.y - qy1f), 0.f);
        const bool keep = ((uint32_t)lane < take) && ((ddx * ddx + ddy * ddy) <= r2.z);
        const unsigned long long mask = __ballot(keep);
        const int slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        if (keep) {
            my[slot * 3 + 0] = r0;
            my[slot * 3 + 1] = r1;
            // b, depth, 1-based list position, Gaussian index
            my[slot * 3 + 2] = make_float4(r2.x, r2.y, __uint_as_float(top - (uint32_t)lane - range.x), __uint_as_float(id));
        }
        // prefetch the next (nearer) batch
        {
            const uint32_t take_n = remaining < 64u ? remaining : 64u;
            if ((uint32_t)lane < take_n) {
                id = p.point_list[range.x + remaining - 1u - (uint32_t)lane];
                const float4* src = reinterpret_cast<const float4*>(p.rec + id);
                r0 = src[0]; r1 = src[1]; r2 = src[2];
            }
        }
        const int cnt = __popcll(mask);
        __builtin_amdgcn_wave_barrier();
        if (cnt == 0) continue;

        float4 a = my[0], b = my[1], c = my[2];
        for (int j = 0; j < cnt; j++) {
            const int jn = (j + 1 < cnt) ? j + 1 : j;
            float4 na = my[jn * 3 + 0], nb = my[jn * 3 + 1], nc = my[jn * 3 + 2];
            __builtin_amdgcn_sched_barrier(0);
            // a: x, y, ca, cb   b: cc, op, r, g   c: b, depth, pos, id
            const uint32_t pos1 = __float_as_uint(c.z);
            const float dx = a.x - pxf, dy = a.y - pyf;
            const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
            const float G = __builtin_amdgcn_exp2f(power * 1.4426950408889634f);
            const float alpha = fminf(0.99f, b.y * G);
            const bool act = (pos1 <= last_contributor) && (power <= 0.0f) && (alpha >= 1.0f / 255.0f);
            if (__any(act)) {
                float v[16];
#pragma unroll
                for (int i = 0; i < 16; i++) v[i] = 0.f;
                if (act) {
                    const float one_m_alpha = 1.f - alpha;
                    T = T / one_m_alpha;
                    const float wgt = alpha * T;
                    const float one_m_last = 1.f - last_alpha;
                    acc_r = last_alpha * last_r + one_m_last * acc_r;
                    acc_g = last_alpha * last_g + one_m_last * acc_g;
                    acc_b = last_alpha * last_b + one_m_last * acc_b;
                    acc_d = last_alpha * last_d + one_m_last * acc_d;
                    acc_a = last_alpha + one_m_last * acc_a;
                    last_r = b.z; last_g = b.w; last_b = c.x; last_d = c.y;
                    float dL_dopa = (b.z - acc_r) * gr + (b.w - acc_g) * gg + (c.x - acc_b) * gb;
                    dL_dopa += (c.y - acc_d) * gd;
                    dL_dopa += (1.f - acc_a) * ga;
                    dL_dopa *= T;
                    last_alpha = alpha;
                    dL_dopa += (-T_final / one_m_alpha) * bg_dot;
                    const float dL_dG = b.y * dL_dopa;
                    const float gdx = G * dx, gdy = G * dy;
                    const float dG_ddelx = -gdx * a.z - gdy * a.w;
                    const float dG_ddely = -gdy * b.x - gdx * a.w;
                    v[A_M2X] = dL_dG * dG_ddelx * ddelx_dx;
                    v[A_M2Y] = dL_dG * dG_ddely * ddely_dy;
                    v[A_CONX] = -0.5f * gdx * dx * dL_dG;
                    v[A_CONY] = -0.5f * gdx * dy * dL_dG;
                    v[A_CONW] = -0.5f * gdy * dy * dL_dG;
                    v[A_OPAC] = G * dL_dopa;
                    v[A_COLR] = wgt * gr;
                    v[A_COLG] = wgt * gg;
                    v[A_COLB] = wgt * gb;
                    v[A_DEPTH] = wgt * gd;
                }
                const float tot = wave_reduce16_transposed(v, lane);
                // value slot 15 is unused by the gradients: it carries the "this quad touched the splat" flag
                if ((lane & 3) == 0) myg[(lane >> 2) * GS + j] = (lane == 60) ? 1.f : tot;
            } else if (lane == 0) {
                myg[15 * GS + j] = 0.f;
            }
            asm volatile("" : "+v"(na.x), "+v"(na.y), "+v"(na.z), "+v"(na.w), "+v"(nb.x), "+v"(nb.y), "+v"(nb.z),
                         "+v"(nb.w), "+v"(nc.x), "+v"(nc.y), "+v"(nc.z), "+v"(nc.w));
            a = na; b = nb; c = nc;
        }
        __builtin_amdgcn_wave_barrier();

        // flush: lane j owns compacted splat j; slab rows are bank-conflict free for both the 16-lane column write above
        // and this 64-lane row read (stride 65 floats)
        if (lane < cnt && myg[15 * GS + lane] != 0.f) {
            const uint32_t gid = __float_as_uint(my[lane * 3 + 2].w);
            float* dst = p.accum + (size_t)gid * kAccumFloats;
#pragma unroll
            for (int i = 0; i < 10; i++) atomicAdd(dst + i, myg[i * GS + lane]);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ void __launch_bounds__(64) debug_wave_reduce16_kernel(const float* __restrict__ in, float* __restrict__ out)
{
    const int lane = threadIdx.x;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = in[lane * 16 + i];
    out[lane] = wave_reduce16_transposed(v, lane);
}

int launch_debug_wave_reduce16(const float* in, float* out, hipStream_t s)
{
    hipLaunchKernelGGL(debug_wave_reduce16_kernel, dim3(1), dim3(64), 0, s, in, out);
    return check_hip(hipGetLastError(), "debug_wave_reduce16_kernel");
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
    p.ranges = reinterpret_cast<const uint2*>(ib + il.ranges);
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
    { ProfScope ps(AG_K_BLEND_BACKWARD, s); hipLaunchKernelGGL(blend_backward_kernel, dim3(p.T), dim3(256), 0, s, p); }
    return check_hip(hipGetLastError(), "blend_backward_kernel");
}

}  // namespace ag
