// Backward of the per-Gaussian preprocess, fused into one streaming kernel (gfx950).
//
// Replaces computeCov2DCUDA (reference cuda_rasterizer/backward.cu:144-274) + preprocessCUDA<3> backward
// (backward.cu:346-412) + computeCov3D backward (backward.cu:278-341), which the reference runs as two launches
// with dL_dmeans / dL_dcov3D round-tripping through HBM, plus the ten torch::zeros fills of
// rasterize_points.cu:158-167 (every output element is written here, zeros for culled Gaussians).
// One thread per Gaussian: reads its 64-byte accumulator line (blend backward), 12+24+12+16 B of inputs, writes
// the 8 output rows.  HBM-bound: ~128 B in + ~104 B out per Gaussian.
// Round 5: the rows enter and leave through LDS.  A thread's own rows are 3 / 6 / 4 floats at a 12 / 24 / 16-byte lane stride -- 17 load
// and 23 store instructions that each touch 12-24 cache lines a third or a sixth at a time; staged, the workgroup moves every array
// of its 256 Gaussians as one contiguous run of 16-byte accesses, lanes along addresses (8 store instructions per thread), exactly as
// the forward preprocess writes its records.
//
// Semantics kept from the reference: the 1.3*tanfov clamp zeroes dL/dt.x, dL/dt.y outside the frustum guard
// (:168-176), denom2inv = 1/(denom^2 + 1e-7) (:203), p_w = 1/(w + 1e-7) (:376), the depth gradient enters through
// view-matrix row 2 (:391-403), the quaternion gradient is w.r.t. the RAW (un-normalised) quaternion (:340).
#include "ag_common.h"
#include "ag_sh.h"

namespace ag {

struct M3b { float m[3][3]; };  // m[col][row], GLM convention

__device__ __forceinline__ M3b mb_cols(float a, float b, float c, float d, float e, float f, float g, float h, float i)
{
    M3b r;
    r.m[0][0] = a; r.m[0][1] = b; r.m[0][2] = c;
    r.m[1][0] = d; r.m[1][1] = e; r.m[1][2] = f;
    r.m[2][0] = g; r.m[2][1] = h; r.m[2][2] = i;
    return r;
}

__device__ __forceinline__ M3b mb_mul(const M3b& a, const M3b& b)
{
    M3b r;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int q = 0; q < 3; q++)
            r.m[c][q] = a.m[0][q] * b.m[c][0] + a.m[1][q] * b.m[c][1] + a.m[2][q] * b.m[c][2];
    return r;
}

__device__ __forceinline__ M3b mb_t(const M3b& a)
{
    M3b r;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int q = 0; q < 3; q++) r.m[c][q] = a.m[q][c];
    return r;
}

struct PreBwdParams {
    int P;
    float h_x, h_y, tan_fovx, tan_fovy, scale_modifier;
    const float* __restrict__ means3D;
    const int* __restrict__ radii;
    const float* __restrict__ scales;
    const float* __restrict__ rotations;
    const float* __restrict__ cov3Ds;
    const float* __restrict__ view;
    const float* __restrict__ proj;
    const float* __restrict__ accum;
    const GaussRec* __restrict__ rec;       // forward records: conic + opacity of the moment -> gradient map
    float half_w, half_h;                   // d(ndc -> pixel)/d(ndc): 0.5 W, 0.5 H (backward.cu:463-464)
    const float* __restrict__ shs;          // NULL on the colors_precomp path
    const float* __restrict__ campos;
    const uint8_t* __restrict__ clamped;
    int sh_degree, sh_coeffs;
    float* __restrict__ dL_dsh;             // [P][sh_coeffs][3], every element written
    float* __restrict__ dL_dmeans2D;
    float* __restrict__ dL_dcolors;
    float* __restrict__ dL_dopacity;
    float* __restrict__ dL_dmeans3D;
    float* __restrict__ dL_dcov3D;
    float* __restrict__ dL_dscales;
    float* __restrict__ dL_drotations;
    int accumulate;                          // 1: add into the outputs (sum over the views of a step) instead of overwriting them
};

typedef float pb_f4 __attribute__((ext_vector_type(4)));

// rows [first, first + nv) of a [P][K] array <-> LDS, lanes along addresses.  first is a multiple of 256, so the run starts 16-byte
// aligned whenever the array itself is (torch allocations are; an unaligned base takes the scalar path).
template <int K>
__device__ __forceinline__ void stage_in(float* __restrict__ s, const float* __restrict__ src, int first, int nv)
{
    const float* g = src + (size_t)first * K;
    const int nf = nv * K;
    if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
        const int nf4 = nf >> 2;
        for (int i = threadIdx.x; i < nf4; i += 256) reinterpret_cast<pb_f4*>(s)[i] = reinterpret_cast<const pb_f4*>(g)[i];
        if ((int)threadIdx.x < (nf & 3)) s[4 * nf4 + threadIdx.x] = g[4 * nf4 + threadIdx.x];
    } else {
        for (int i = threadIdx.x; i < nf; i += 256) s[i] = g[i];
    }
}

template <int K>
__device__ __forceinline__ void stage_out(float* __restrict__ dst, const float* __restrict__ s, int first, int nv, bool accumulate)
{
    float* g = dst + (size_t)first * K;
    const int nf = nv * K;
    if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
        const int nf4 = nf >> 2;
        for (int i = threadIdx.x; i < nf4; i += 256) {
            pb_f4 v = reinterpret_cast<const pb_f4*>(s)[i];
            if (accumulate) v += reinterpret_cast<const pb_f4*>(g)[i];
            reinterpret_cast<pb_f4*>(g)[i] = v;
        }
        if ((int)threadIdx.x < (nf & 3)) {
            const int i = 4 * nf4 + threadIdx.x;
            g[i] = accumulate ? g[i] + s[i] : s[i];
        }
    } else {
        for (int i = threadIdx.x; i < nf; i += 256) g[i] = accumulate ? g[i] + s[i] : s[i];
    }
}

__global__ void __launch_bounds__(256) preprocess_backward_kernel(PreBwdParams p)
{
    // staging buffer: first the inputs [means3D 3][cov3D 6][scales 3][rotations 4] x 256, then (after a barrier) the outputs
    // [means2D 3][colors 3][means3D 3][scales 3][opacity 1][cov3D 6][rotations 4] x 256
    __shared__ __attribute__((aligned(16))) float s_io[256 * 23];
    const int first = blockIdx.x * 256, nv = min(256, p.P - first);          // Gaussians of this workgroup
    const int tid = threadIdx.x, idx = first + tid;
    const bool valid = tid < nv;
    float* const s_mean = s_io, * const s_c3 = s_io + 256 * 3, * const s_sc = s_io + 256 * 9, * const s_rot = s_io + 256 * 12;
    stage_in<3>(s_mean, p.means3D, first, nv);
    stage_in<6>(s_c3, p.cov3Ds, first, nv);
    if (p.scales) { stage_in<3>(s_sc, p.scales, first, nv); stage_in<4>(s_rot, p.rotations, first, nv); }
    const int my_radius = valid ? p.radii[idx] : 0;
    __syncthreads();

    float g_m2[3] = { 0.f, 0.f, 0.f }, g_col[3] = { 0.f, 0.f, 0.f }, g_op = 0.f;
    float g_mean[3] = { 0.f, 0.f, 0.f }, g_cov[6] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
    float g_scale[3] = { 0.f, 0.f, 0.f }, g_rot[4] = { 0.f, 0.f, 0.f, 0.f };

    if (my_radius > 0) {
        float V[16], Pm[16];
#pragma unroll
        for (int i = 0; i < 16; i++) { V[i] = p.view[i]; Pm[i] = p.proj[i]; }
        const float4* arow = reinterpret_cast<const float4*>(p.accum + (size_t)idx * kAccumFloats);
        const float4 a0 = arow[0], a1 = arow[1], a2 = arow[2];
        // moments of q = G dL/dalpha over the Gaussian's pixels -> the gradients backward.cu:560-600 accumulates per (pixel, entry)
        // (ag_common.h AccumSlot): dG/ddelx = -G (ca dx + cb dy), dL/dG = op dL/dalpha, dL/dconic = -0.5 G d d^T dL/dG
        const GaussRec gr = p.rec[idx];
        const float nhop = -0.5f * gr.op;
        g_m2[0] = (2.0f * nhop * p.half_w) * fmaf(gr.ca, a0.x, gr.cb * a0.y);
        g_m2[1] = (2.0f * nhop * p.half_h) * fmaf(gr.cc, a0.y, gr.cb * a0.x);
        const float dcon_x = nhop * a0.z, dcon_y = nhop * a0.w, dcon_w = nhop * a1.x;
        g_op = a1.y;
        g_col[0] = a1.z; g_col[1] = a1.w; g_col[2] = a2.x;
        const float g_depth = a2.y;

        const float mx = s_mean[3 * tid + 0], my = s_mean[3 * tid + 1], mz = s_mean[3 * tid + 2];
        float c3[6];
#pragma unroll
        for (int k = 0; k < 6; k++) c3[k] = s_c3[6 * tid + k];

        // ---- cov2D backward (backward.cu:144-274) ----
        float tx = V[0] * mx + V[4] * my + V[8] * mz + V[12];
        float ty = V[1] * mx + V[5] * my + V[9] * mz + V[13];
        const float tz = V[2] * mx + V[6] * my + V[10] * mz + V[14];
        const float limx = 1.3f * p.tan_fovx, limy = 1.3f * p.tan_fovy;
        const float txtz = tx / tz, tytz = ty / tz;
        tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
        ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
        const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
        const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;

        const M3b J = mb_cols(p.h_x / tz, 0.0f, -(p.h_x * tx) / (tz * tz),
                              0.0f, p.h_y / tz, -(p.h_y * ty) / (tz * tz),
                              0.f, 0.f, 0.f);
        const M3b Wm = mb_cols(V[0], V[4], V[8], V[1], V[5], V[9], V[2], V[6], V[10]);
        const M3b Vrk = mb_cols(c3[0], c3[1], c3[2], c3[1], c3[3], c3[4], c3[2], c3[4], c3[5]);
        const M3b Tm = mb_mul(Wm, J);
        const M3b cov2D = mb_mul(mb_mul(mb_t(Tm), mb_t(Vrk)), Tm);
        const float a = cov2D.m[0][0] + 0.3f, b = cov2D.m[0][1], c = cov2D.m[1][1] + 0.3f;
        const float denom = a * c - b * b;
        float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
#define TT(i, j) Tm.m[i][j]
#define VV(i, j) Vrk.m[i][j]
        if (denom2inv != 0.f) {
            dL_da = denom2inv * (-c * c * dcon_x + 2 * b * c * dcon_y + (denom - a * c) * dcon_w);
            dL_dc = denom2inv * (-a * a * dcon_w + 2 * a * b * dcon_y + (denom - a * c) * dcon_x);
            dL_db = denom2inv * 2 * (b * c * dcon_x - (denom + 2 * b * b) * dcon_y + a * b * dcon_w);
            g_cov[0] = (TT(0, 0) * TT(0, 0) * dL_da + TT(0, 0) * TT(1, 0) * dL_db + TT(1, 0) * TT(1, 0) * dL_dc);
            g_cov[3] = (TT(0, 1) * TT(0, 1) * dL_da + TT(0, 1) * TT(1, 1) * dL_db + TT(1, 1) * TT(1, 1) * dL_dc);
            g_cov[5] = (TT(0, 2) * TT(0, 2) * dL_da + TT(0, 2) * TT(1, 2) * dL_db + TT(1, 2) * TT(1, 2) * dL_dc);
            g_cov[1] = 2 * TT(0, 0) * TT(0, 1) * dL_da + (TT(0, 0) * TT(1, 1) + TT(0, 1) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 1) * dL_dc;
            g_cov[2] = 2 * TT(0, 0) * TT(0, 2) * dL_da + (TT(0, 0) * TT(1, 2) + TT(0, 2) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 2) * dL_dc;
            g_cov[4] = 2 * TT(0, 2) * TT(0, 1) * dL_da + (TT(0, 1) * TT(1, 2) + TT(0, 2) * TT(1, 1)) * dL_db + 2 * TT(1, 1) * TT(1, 2) * dL_dc;
        }
        const float dL_dT00 = 2 * (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_da +
                              (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_db;
        const float dL_dT01 = 2 * (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_da +
                              (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_db;
        const float dL_dT02 = 2 * (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_da +
                              (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_db;
        const float dL_dT10 = 2 * (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_dc +
                              (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_db;
        const float dL_dT11 = 2 * (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_dc +
                              (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_db;
        const float dL_dT12 = 2 * (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_dc +
                              (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_db;
#undef TT
#undef VV
        const float dL_dJ00 = Wm.m[0][0] * dL_dT00 + Wm.m[0][1] * dL_dT01 + Wm.m[0][2] * dL_dT02;
        const float dL_dJ02 = Wm.m[2][0] * dL_dT00 + Wm.m[2][1] * dL_dT01 + Wm.m[2][2] * dL_dT02;
        const float dL_dJ11 = Wm.m[1][0] * dL_dT10 + Wm.m[1][1] * dL_dT11 + Wm.m[1][2] * dL_dT12;
        const float dL_dJ12 = Wm.m[2][0] * dL_dT10 + Wm.m[2][1] * dL_dT11 + Wm.m[2][2] * dL_dT12;
        const float itz = 1.f / tz;
        const float tz2 = itz * itz;
        const float tz3 = tz2 * itz;
        const float dL_dtx = x_grad_mul * -p.h_x * tz2 * dL_dJ02;
        const float dL_dty = y_grad_mul * -p.h_y * tz2 * dL_dJ12;
        const float dL_dtz = -p.h_x * tz2 * dL_dJ00 - p.h_y * tz2 * dL_dJ11 + (2 * p.h_x * tx) * tz3 * dL_dJ02 +
                             (2 * p.h_y * ty) * tz3 * dL_dJ12;
        g_mean[0] = V[0] * dL_dtx + V[1] * dL_dty + V[2] * dL_dtz;
        g_mean[1] = V[4] * dL_dtx + V[5] * dL_dty + V[6] * dL_dtz;
        g_mean[2] = V[8] * dL_dtx + V[9] * dL_dty + V[10] * dL_dtz;

        // ---- projection + depth path (backward.cu:346-403) ----
        const float hw = Pm[3] * mx + Pm[7] * my + Pm[11] * mz + Pm[15];
        const float m_w = 1.0f / (hw + 0.0000001f);
        const float mul1 = (Pm[0] * mx + Pm[4] * my + Pm[8] * mz + Pm[12]) * m_w * m_w;
        const float mul2 = (Pm[1] * mx + Pm[5] * my + Pm[9] * mz + Pm[13]) * m_w * m_w;
        g_mean[0] += (Pm[0] * m_w - Pm[3] * mul1) * g_m2[0] + (Pm[1] * m_w - Pm[3] * mul2) * g_m2[1];
        g_mean[1] += (Pm[4] * m_w - Pm[7] * mul1) * g_m2[0] + (Pm[5] * m_w - Pm[7] * mul2) * g_m2[1];
        g_mean[2] += (Pm[8] * m_w - Pm[11] * mul1) * g_m2[0] + (Pm[9] * m_w - Pm[11] * mul2) * g_m2[1];
        const float mul3 = V[2] * mx + V[6] * my + V[10] * mz + V[14];
        g_mean[0] += (V[2] - V[3] * mul3) * g_depth;
        g_mean[1] += (V[6] - V[7] * mul3) * g_depth;
        g_mean[2] += (V[10] - V[11] * mul3) * g_depth;

        // ---- spherical-harmonics colours (backward.cu:406-407 -> :20-139), after the projection and depth terms ----
        if (p.shs) {
            const float gc[3] = { g_col[0] * (p.clamped[3 * idx + 0] ? 0.f : 1.f), g_col[1] * (p.clamped[3 * idx + 1] ? 0.f : 1.f),
                                  g_col[2] * (p.clamped[3 * idx + 2] ? 0.f : 1.f) };
            float gm[3];
            sh_backward(p.sh_degree, sh_direction(mx, my, mz, p.campos), p.shs + (size_t)idx * p.sh_coeffs * 3, gc,
                        p.dL_dsh + (size_t)idx * p.sh_coeffs * 3, gm);
            g_mean[0] += gm[0]; g_mean[1] += gm[1]; g_mean[2] += gm[2];
        }

        // ---- cov3D -> scale / raw quaternion (backward.cu:278-341) ----
        if (p.scales) {
            const float r = s_rot[4 * tid + 0], x = s_rot[4 * tid + 1];
            const float y = s_rot[4 * tid + 2], z = s_rot[4 * tid + 3];
            const M3b R = mb_cols(
                1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
            const float s0 = p.scale_modifier * s_sc[3 * tid + 0];
            const float s1 = p.scale_modifier * s_sc[3 * tid + 1];
            const float s2 = p.scale_modifier * s_sc[3 * tid + 2];
            M3b S = mb_cols(1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f);
            S.m[0][0] = s0; S.m[1][1] = s1; S.m[2][2] = s2;
            M3b M = mb_mul(S, R);
            const M3b dSig = mb_cols(g_cov[0], 0.5f * g_cov[1], 0.5f * g_cov[2],
                                     0.5f * g_cov[1], g_cov[3], 0.5f * g_cov[4],
                                     0.5f * g_cov[2], 0.5f * g_cov[4], g_cov[5]);
#pragma unroll
            for (int cc = 0; cc < 3; cc++)
#pragma unroll
                for (int q = 0; q < 3; q++) M.m[cc][q] = 2.0f * M.m[cc][q];
            const M3b dL_dM = mb_mul(M, dSig);
            const M3b Rt = mb_t(R);
            M3b D = mb_t(dL_dM);
            g_scale[0] = Rt.m[0][0] * D.m[0][0] + Rt.m[0][1] * D.m[0][1] + Rt.m[0][2] * D.m[0][2];
            g_scale[1] = Rt.m[1][0] * D.m[1][0] + Rt.m[1][1] * D.m[1][1] + Rt.m[1][2] * D.m[1][2];
            g_scale[2] = Rt.m[2][0] * D.m[2][0] + Rt.m[2][1] * D.m[2][1] + Rt.m[2][2] * D.m[2][2];
#pragma unroll
            for (int q = 0; q < 3; q++) { D.m[0][q] *= s0; D.m[1][q] *= s1; D.m[2][q] *= s2; }
            g_rot[0] = 2 * z * (D.m[0][1] - D.m[1][0]) + 2 * y * (D.m[2][0] - D.m[0][2]) + 2 * x * (D.m[1][2] - D.m[2][1]);
            g_rot[1] = 2 * y * (D.m[1][0] + D.m[0][1]) + 2 * z * (D.m[2][0] + D.m[0][2]) + 2 * r * (D.m[1][2] - D.m[2][1]) - 4 * x * (D.m[2][2] + D.m[1][1]);
            g_rot[2] = 2 * x * (D.m[1][0] + D.m[0][1]) + 2 * r * (D.m[2][0] - D.m[0][2]) + 2 * z * (D.m[1][2] + D.m[2][1]) - 4 * y * (D.m[2][2] + D.m[0][0]);
            g_rot[3] = 2 * r * (D.m[0][1] - D.m[1][0]) + 2 * x * (D.m[2][0] + D.m[0][2]) + 2 * y * (D.m[1][2] + D.m[2][1]) - 4 * z * (D.m[1][1] + D.m[0][0]);
        }
    }

    if (p.shs && valid) {    // coefficients the active degree does not use, and every coefficient of a culled Gaussian: zero
        const int first_unused = (my_radius > 0) ? 3 * (p.sh_degree + 1) * (p.sh_degree + 1) : 0;
        float* row = p.dL_dsh + (size_t)idx * p.sh_coeffs * 3;
        for (int i = first_unused; i < 3 * p.sh_coeffs; i++) row[i] = 0.f;
    }
    __syncthreads();            // every thread has read its inputs: the buffer now takes the outputs
    float* const o_m2 = s_io, * const o_col = s_io + 256 * 3, * const o_mean = s_io + 256 * 6, * const o_sc = s_io + 256 * 9;
    float* const o_op = s_io + 256 * 12, * const o_cov = s_io + 256 * 13, * const o_rot = s_io + 256 * 19;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        o_m2[3 * tid + k] = g_m2[k];
        o_col[3 * tid + k] = g_col[k];
        o_mean[3 * tid + k] = g_mean[k];
        o_sc[3 * tid + k] = g_scale[k];
    }
    o_op[tid] = g_op;
#pragma unroll
    for (int k = 0; k < 6; k++) o_cov[6 * tid + k] = g_cov[k];
#pragma unroll
    for (int k = 0; k < 4; k++) o_rot[4 * tid + k] = g_rot[k];
    __syncthreads();
    // (accumulate: multi-view step -- every Gaussian's rows are owned by this workgroup, so plain read-modify-write sums the views)
    const bool acc = p.accumulate != 0;
    stage_out<3>(p.dL_dmeans2D, o_m2, first, nv, acc);
    stage_out<3>(p.dL_dcolors, o_col, first, nv, acc);
    stage_out<3>(p.dL_dmeans3D, o_mean, first, nv, acc);
    stage_out<3>(p.dL_dscales, o_sc, first, nv, acc);
    stage_out<1>(p.dL_dopacity, o_op, first, nv, acc);
    stage_out<6>(p.dL_dcov3D, o_cov, first, nv, acc);
    stage_out<4>(p.dL_drotations, o_rot, first, nv, acc);
}

int launch_preprocess_backward(const AgRasterBackwardArgs& a, hipStream_t s)
{
    PreBwdParams p;
    p.P = a.P;
    p.h_y = a.H / (2.0f * a.tan_fovy);   // rasterizer_impl.cu:386-387
    p.h_x = a.W / (2.0f * a.tan_fovx);
    p.tan_fovx = a.tan_fovx; p.tan_fovy = a.tan_fovy; p.scale_modifier = a.scale_modifier;
    p.means3D = a.means3D; p.radii = a.radii; p.scales = a.scales; p.rotations = a.rotations;
    const char* gb = aligned_base(a.geom_buffer);
    GeomLayout gl((size_t)a.P);
    p.cov3Ds = a.cov3D_precomp ? a.cov3D_precomp : reinterpret_cast<const float*>(gb + gl.cov3d);
    p.view = a.viewmatrix; p.proj = a.projmatrix;
    p.accum = reinterpret_cast<const float*>(aligned_base(a.accum_buffer));
    p.rec = reinterpret_cast<const GaussRec*>(gb + gl.rec);
    p.half_w = 0.5f * (float)a.W; p.half_h = 0.5f * (float)a.H;
    p.shs = a.colors_precomp ? nullptr : a.shs; p.campos = a.campos; p.sh_degree = a.sh_degree; p.sh_coeffs = a.sh_coeffs;
    p.clamped = reinterpret_cast<const uint8_t*>(gb + gl.clamped); p.dL_dsh = a.dL_dsh;
    p.dL_dmeans2D = a.dL_dmeans2D; p.dL_dcolors = a.dL_dcolors; p.dL_dopacity = a.dL_dopacity;
    p.dL_dmeans3D = a.dL_dmeans3D; p.dL_dcov3D = a.dL_dcov3D; p.dL_dscales = a.dL_dscales;
    p.dL_drotations = a.dL_drotations;
    p.accumulate = a.accumulate;
    { ProfScope ps(AG_K_PREPROCESS_BACKWARD, s); hipLaunchKernelGGL(preprocess_backward_kernel, dim3((a.P + 255) / 256), dim3(256), 0, s, p); }
    return check_hip(hipGetLastError(), "preprocess_backward_kernel");
}

}  // namespace ag
