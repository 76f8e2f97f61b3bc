// Per-Gaussian assembly (map gather + activations) and linear-blend skinning, forward and backward (gfx950).
//
// Replaces the torch-op chains of reference network/avatar.py:84-124 (see include/ag_avatar.h).  All four kernels are
// HBM-bound streaming kernels, one thread per Gaussian:
//   gather fwd : 14 strided-but-coalesced channel reads (consecutive Gaussians are consecutive map pixels) + 44 B of
//                canonical parameters in, 56 B out                                              ~160 B / Gaussian
//   gather bwd : the same reads + 56 B of upstream gradients, 14 scattered channel writes into pre-zeroed maps
//   lbs fwd/bwd: the [N, J] blend-weight rows (220 B / Gaussian at J = 55, the dominant stream) are staged through
//                wave-private LDS with fully coalesced loads and read back conflict-free (row stride J is odd);
//                the J joint matrices are wave-uniform and come in through scalar loads.
#include "ag_common.h"
#include "../../include/ag_avatar.h"

namespace ag {

// ------------------------------------------------------------------------------------------------------------------
// gather + activations
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ size_t map_offset(int pix, int S, int C, int c)
{
    // canvas pixel (v, u) of the [S, 2S] front|back layout -> element of the NCHW [1, 2C, S, S] network output
    const int v = pix / (2 * S), u = pix - v * 2 * S;
    const int back = u >= S;
    return ((size_t)(back * C + c) * S + v) * S + (u - back * S);
}

// Each of the three parts (positions | opacity, scales, rotations | colours) is produced when its output pointer is set;
// a null MAP pointer stands for an all-zero network output (the activations of the canonical parameters alone:
// GaussianModel.get_opacity / get_scaling / get_rotation, gaussians/gaussian_model.py:115-147).  All wave-uniform.
__global__ void __launch_bounds__(256) gather_forward_kernel(AgGatherArgs a)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= a.N) return;
    const int pix = a.pix[n];
    const size_t plane = (size_t)a.S * a.S;
    const size_t o3 = map_offset(pix, a.S, 3, 0), o8 = map_offset(pix, a.S, 8, 0);
    if (a.positions) {
#pragma unroll
        for (int c = 0; c < 3; c++)
            a.positions[3 * n + c] = 0.05f * (a.position_map ? a.position_map[o3 + c * plane] : 0.f) + a.xyz[3 * n + c];
    }
    if (a.colors) {
#pragma unroll
        for (int c = 0; c < 3; c++) a.colors[3 * n + c] = a.color_map ? a.color_map[o3 + c * plane] : 0.f;
    }
    if (!a.opacity) return;
    float m[8];
#pragma unroll
    for (int c = 0; c < 8; c++) m[c] = a.other_map ? a.other_map[o8 + c * plane] : 0.f;
    const float o = m[0] + a.opacity_raw[n];
    a.opacity[n] = 1.0f / (1.0f + expf(-o));
#pragma unroll
    for (int c = 0; c < 3; c++) a.scales[3 * n + c] = expf(m[1 + c] + a.scaling_raw[3 * n + c]);
    float q[4], nn = 0.f;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        q[c] = m[4 + c] + a.rotation_raw[4 * n + c];
        nn += q[c] * q[c];
    }
    const float inv = 1.0f / fmaxf(sqrtf(nn), 1e-12f);   // F.normalize: x / max(||x||, eps)
#pragma unroll
    for (int c = 0; c < 4; c++) a.rotations[4 * n + c] = q[c] * inv;
}

// A null gradient-map pointer skips that part (its upstream-gradient slots are then not read).
__global__ void __launch_bounds__(256) gather_backward_kernel(AgGatherArgs a, float* __restrict__ g_pos_map,
                                                             float* __restrict__ g_other_map, float* __restrict__ g_col_map)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= a.N) return;
    const int pix = a.pix[n];
    const size_t plane = (size_t)a.S * a.S;
    const size_t o3 = map_offset(pix, a.S, 3, 0), o8 = map_offset(pix, a.S, 8, 0);
    if (g_pos_map) {
#pragma unroll
        for (int c = 0; c < 3; c++) g_pos_map[o3 + c * plane] = 0.05f * a.positions[3 * n + c];
    }
    if (g_col_map) {
#pragma unroll
        for (int c = 0; c < 3; c++) g_col_map[o3 + c * plane] = a.colors[3 * n + c];
    }
    if (!g_other_map) return;
    // sigmoid'
    const float o = a.other_map[o8] + a.opacity_raw[n];
    const float sg = 1.0f / (1.0f + expf(-o));
    g_other_map[o8] = a.opacity[n] * sg * (1.0f - sg);
    // exp'
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float e = expf(a.other_map[o8 + (1 + c) * plane] + a.scaling_raw[3 * n + c]);
        g_other_map[o8 + (1 + c) * plane] = a.scales[3 * n + c] * e;
    }
    // normalize': y = x / max(|x|, eps); for |x| > eps: dx = (g - y (y.g)) / |x|, else dx = g / eps
    float q[4], nn = 0.f, g[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        q[c] = a.other_map[o8 + (4 + c) * plane] + a.rotation_raw[4 * n + c];
        nn += q[c] * q[c];
        g[c] = a.rotations[4 * n + c];
    }
    const float norm = sqrtf(nn);
    if (norm > 1e-12f) {
        const float inv = 1.0f / norm;
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < 4; c++) dot += q[c] * inv * g[c];
#pragma unroll
        for (int c = 0; c < 4; c++) g_other_map[o8 + (4 + c) * plane] = (g[c] - q[c] * inv * dot) * inv;
    } else {
#pragma unroll
        for (int c = 0; c < 4; c++) g_other_map[o8 + (4 + c) * plane] = g[c] * 1e12f;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// linear-blend skinning
// ------------------------------------------------------------------------------------------------------------------
constexpr int kMaxJ = 256;

// Blend the joint matrices of one Gaussian: M[3][4] = sum_j w_j A_j[:3, :4].  `row` points at this lane's weights in
// LDS; jnt is wave-uniform (scalar loads).
__device__ __forceinline__ void blend_matrix(const float* row, const float* __restrict__ jnt, int J, float (&M)[12])
{
#pragma unroll
    for (int k = 0; k < 12; k++) M[k] = 0.f;
    for (int j = 0; j < J; j++) {
        const float w = row[j];
#pragma unroll
        for (int k = 0; k < 12; k++) M[k] += w * jnt[16 * j + k];
    }
}

// The same blend from the sparse form of the weight rows: K (joint, weight) pairs per Gaussian in ascending joint order, padded
// with zero weights, stored [K][N] so that lanes read consecutive addresses.  Skipping the exact zeros of a row leaves every
// partial sum unchanged (x + 0 * a == x), so the result equals the dense loop's bit for bit (up to the sign of an exact zero).
__device__ __forceinline__ void blend_matrix_sparse(const uint8_t* __restrict__ idx, const float* __restrict__ wts, int K, int N, int n,
                                                    const float* __restrict__ jnt, float (&M)[12])
{
#pragma unroll
    for (int k = 0; k < 12; k++) M[k] = 0.f;
    for (int s = 0; s < K; s++) {
        const float w = wts[(size_t)s * N + n];
        const float* A = jnt + 16 * (int)idx[(size_t)s * N + n];
#pragma unroll
        for (int k = 0; k < 12; k++) M[k] += w * A[k];
    }
}

// Coalesced load of the 64 weight rows of this wave into its LDS slab (row stride J).
__device__ __forceinline__ void stage_rows(const float* __restrict__ lbs, int N, int J, int first, float* slab, int lane)
{
    const size_t base = (size_t)first * J;
    const int rows = min(64, N - first);
    const int total = rows * J;
    for (int i = lane; i < total; i += 64) slab[i] = lbs[base + i];
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void quat_to_mat(const float (&q)[4], float (&R)[9], float& two_s)
{
    const float r = q[0], i = q[1], j = q[2], k = q[3];
    two_s = 2.0f / (r * r + i * i + j * j + k * k);
    R[0] = 1 - two_s * (j * j + k * k); R[1] = two_s * (i * j - k * r);     R[2] = two_s * (i * k + j * r);
    R[3] = two_s * (i * j + k * r);     R[4] = 1 - two_s * (i * i + k * k); R[5] = two_s * (j * k - i * r);
    R[6] = two_s * (i * k - j * r);     R[7] = two_s * (j * k + i * r);     R[8] = 1 - two_s * (i * i + j * j);
}

template <bool SPARSE>
__global__ void __launch_bounds__(256) lbs_forward_kernel(AgLbsArgs a)
{
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int first = (blockIdx.x * 4 + wave) * 64;
    if (first >= a.N) return;
    const int n = first + lane;
    float M[12];
    if constexpr (SPARSE) {
        if (n >= a.N) return;
        blend_matrix_sparse(a.sp_idx, a.sp_w, a.K, a.N, n, a.jnt_mats, M);
    } else {
        float* slab = lds + (size_t)wave * 64 * a.J;
        stage_rows(a.lbs, a.N, a.J, first, slab, lane);
        if (n >= a.N) return;
        blend_matrix(slab + lane * a.J, a.jnt_mats, a.J, M);
    }
    const float px = a.positions[3 * n], py = a.positions[3 * n + 1], pz = a.positions[3 * n + 2];
    a.out_positions[3 * n + 0] = M[0] * px + M[1] * py + M[2] * pz + M[3];
    a.out_positions[3 * n + 1] = M[4] * px + M[5] * py + M[6] * pz + M[7];
    a.out_positions[3 * n + 2] = M[8] * px + M[9] * py + M[10] * pz + M[11];

    float q[4] = { a.rotations[4 * n], a.rotations[4 * n + 1], a.rotations[4 * n + 2], a.rotations[4 * n + 3] };
    float R[9], two_s;
    quat_to_mat(q, R, two_s);
    float m[9];   // m = M3 * R
#pragma unroll
    for (int x = 0; x < 3; x++)
#pragma unroll
        for (int z = 0; z < 3; z++) m[3 * x + z] = M[4 * x] * R[z] + M[4 * x + 1] * R[3 + z] + M[4 * x + 2] * R[6 + z];
    // pytorch3d 0.7.4 matrix_to_quaternion: sqrt of the positive part, arg-max candidate, 0.1 floor, no sign fix
    const float x4[4] = { 1.0f + m[0] + m[4] + m[8], 1.0f + m[0] - m[4] - m[8], 1.0f - m[0] + m[4] - m[8], 1.0f - m[0] - m[4] + m[8] };
    float qa[4];
#pragma unroll
    for (int c = 0; c < 4; c++) qa[c] = x4[c] > 0.f ? sqrtf(x4[c]) : 0.f;
    int best = 0;
#pragma unroll
    for (int c = 1; c < 4; c++) if (qa[c] > qa[best]) best = c;
    float cand[4];
    if (best == 0)      { cand[0] = qa[0] * qa[0]; cand[1] = m[7] - m[5]; cand[2] = m[2] - m[6]; cand[3] = m[3] - m[1]; }
    else if (best == 1) { cand[0] = m[7] - m[5]; cand[1] = qa[1] * qa[1]; cand[2] = m[3] + m[1]; cand[3] = m[2] + m[6]; }
    else if (best == 2) { cand[0] = m[2] - m[6]; cand[1] = m[3] + m[1]; cand[2] = qa[2] * qa[2]; cand[3] = m[5] + m[7]; }
    else                { cand[0] = m[3] - m[1]; cand[1] = m[6] + m[2]; cand[2] = m[7] + m[5]; cand[3] = qa[3] * qa[3]; }
    const float den = 2.0f * fmaxf(qa[best], 0.1f);
#pragma unroll
    for (int c = 0; c < 4; c++) a.out_rotations[4 * n + c] = cand[c] / den;
}

template <bool SPARSE>
__global__ void __launch_bounds__(256) lbs_backward_kernel(AgLbsArgs a, float* __restrict__ g_positions,
                                                          float* __restrict__ g_rotations)
{
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int first = (blockIdx.x * 4 + wave) * 64;
    if (first >= a.N) return;
    const int n = first + lane;
    float M[12];
    if constexpr (SPARSE) {
        if (n >= a.N) return;
        blend_matrix_sparse(a.sp_idx, a.sp_w, a.K, a.N, n, a.jnt_mats, M);
    } else {
        float* slab = lds + (size_t)wave * 64 * a.J;
        stage_rows(a.lbs, a.N, a.J, first, slab, lane);
        if (n >= a.N) return;
        blend_matrix(slab + lane * a.J, a.jnt_mats, a.J, M);
    }

    // positions: dL/dp = M3^T g
    const float gx = a.out_positions[3 * n], gy = a.out_positions[3 * n + 1], gz = a.out_positions[3 * n + 2];
    g_positions[3 * n + 0] = M[0] * gx + M[4] * gy + M[8] * gz;
    g_positions[3 * n + 1] = M[1] * gx + M[5] * gy + M[9] * gz;
    g_positions[3 * n + 2] = M[2] * gx + M[6] * gy + M[10] * gz;

    // rotations: recompute the forward, then walk the chain backwards
    float q[4] = { a.rotations[4 * n], a.rotations[4 * n + 1], a.rotations[4 * n + 2], a.rotations[4 * n + 3] };
    float R[9], two_s;
    quat_to_mat(q, R, two_s);
    float m[9];
#pragma unroll
    for (int x = 0; x < 3; x++)
#pragma unroll
        for (int z = 0; z < 3; z++) m[3 * x + z] = M[4 * x] * R[z] + M[4 * x + 1] * R[3 + z] + M[4 * x + 2] * R[6 + z];
    const float x4[4] = { 1.0f + m[0] + m[4] + m[8], 1.0f + m[0] - m[4] - m[8], 1.0f - m[0] + m[4] - m[8], 1.0f - m[0] - m[4] + m[8] };
    float qa[4];
#pragma unroll
    for (int c = 0; c < 4; c++) qa[c] = x4[c] > 0.f ? sqrtf(x4[c]) : 0.f;
    int best = 0;
#pragma unroll
    for (int c = 1; c < 4; c++) if (qa[c] > qa[best]) best = c;
    float cand[4];
    if (best == 0)      { cand[0] = qa[0] * qa[0]; cand[1] = m[7] - m[5]; cand[2] = m[2] - m[6]; cand[3] = m[3] - m[1]; }
    else if (best == 1) { cand[0] = m[7] - m[5]; cand[1] = qa[1] * qa[1]; cand[2] = m[3] + m[1]; cand[3] = m[2] + m[6]; }
    else if (best == 2) { cand[0] = m[2] - m[6]; cand[1] = m[3] + m[1]; cand[2] = qa[2] * qa[2]; cand[3] = m[5] + m[7]; }
    else                { cand[0] = m[3] - m[1]; cand[1] = m[6] + m[2]; cand[2] = m[7] + m[5]; cand[3] = qa[3] * qa[3]; }
    const float qsel = qa[best];
    const bool floored = !(qsel > 0.1f);
    const float den = 2.0f * (floored ? 0.1f : qsel);
    float go[4] = { a.out_rotations[4 * n], a.out_rotations[4 * n + 1], a.out_rotations[4 * n + 2], a.out_rotations[4 * n + 3] };
    // out = cand / den
    float gc[4], gden = 0.f;
#pragma unroll
    for (int c = 0; c < 4; c++) { gc[c] = go[c] / den; gden -= go[c] * cand[c] / (den * den); }
    // den = 2 max(q_sel, 0.1): no gradient when the floor is active; the diagonal candidate is q_sel^2
    float gq_sel = floored ? 0.f : 2.0f * gden;
    gq_sel += 2.0f * qsel * gc[best];
    // q_sel = sqrt(max(0, x_sel)) (zero sub-gradient at x <= 0)
    const float gx_sel = (x4[best] > 0.f) ? gq_sel / (2.0f * qsel) : 0.f;
    float gm[9];
#pragma unroll
    for (int c = 0; c < 9; c++) gm[c] = 0.f;
    const float s0 = (best == 0 || best == 1) ? 1.f : -1.f;   // sign of m00 in x_best
    const float s1 = (best == 0 || best == 2) ? 1.f : -1.f;   // sign of m11
    const float s2 = (best == 0 || best == 3) ? 1.f : -1.f;   // sign of m22
    gm[0] += s0 * gx_sel; gm[4] += s1 * gx_sel; gm[8] += s2 * gx_sel;
    if (best == 0)      { gm[7] += gc[1]; gm[5] -= gc[1]; gm[2] += gc[2]; gm[6] -= gc[2]; gm[3] += gc[3]; gm[1] -= gc[3]; }
    else if (best == 1) { gm[7] += gc[0]; gm[5] -= gc[0]; gm[3] += gc[2]; gm[1] += gc[2]; gm[2] += gc[3]; gm[6] += gc[3]; }
    else if (best == 2) { gm[2] += gc[0]; gm[6] -= gc[0]; gm[3] += gc[1]; gm[1] += gc[1]; gm[5] += gc[3]; gm[7] += gc[3]; }
    else                { gm[3] += gc[0]; gm[1] -= gc[0]; gm[6] += gc[1]; gm[2] += gc[1]; gm[7] += gc[2]; gm[5] += gc[2]; }
    // m = M3 R  ->  dL/dR = M3^T gm
    float G[9];
#pragma unroll
    for (int y = 0; y < 3; y++)
#pragma unroll
        for (int z = 0; z < 3; z++) G[3 * y + z] = M[y] * gm[z] + M[4 + y] * gm[3 + z] + M[8 + y] * gm[6 + z];
    // R = I + two_s * U(q); dL/dq = two_s * dU^T G - two_s^2 q (U . G)
    const float r = q[0], i = q[1], j = q[2], k = q[3];
    const float U[9] = { -(j * j + k * k), i * j - k * r, i * k + j * r,
                         i * j + k * r, -(i * i + k * k), j * k - i * r,
                         i * k - j * r, j * k + i * r, -(i * i + j * j) };
    float gs = 0.f;
#pragma unroll
    for (int c = 0; c < 9; c++) gs += G[c] * U[c];
    const float dr = -k * G[1] + j * G[2] + k * G[3] - i * G[5] - j * G[6] + i * G[7];
    const float di = j * G[1] + k * G[2] + j * G[3] - 2 * i * G[4] - r * G[5] + k * G[6] + r * G[7] - 2 * i * G[8];
    const float dj = -2 * j * G[0] + i * G[1] + r * G[2] + i * G[3] + k * G[5] - r * G[6] + k * G[7] - 2 * j * G[8];
    const float dk = -2 * k * G[0] - r * G[1] + i * G[2] + r * G[3] - 2 * k * G[4] + j * G[5] + i * G[6] + j * G[7];
    const float t2 = two_s * two_s * gs;
    g_rotations[4 * n + 0] = two_s * dr - t2 * r;
    g_rotations[4 * n + 1] = two_s * di - t2 * i;
    g_rotations[4 * n + 2] = two_s * dj - t2 * j;
    g_rotations[4 * n + 3] = two_s * dk - t2 * k;
}

}  // namespace ag

using namespace ag;

// Eval-time hand fusion (network/avatar.py:183-200): one thread per Gaussian, 11 attribute floats blended in place.
__global__ void __launch_bounds__(256) hand_fuse_kernel(AgHandFuseArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.N) return;
    const float x = a.xyz[3 * i + 0], y = a.xyz[3 * i + 1];
    const float lmin = a.left_box[0], lmax = a.left_box[1], rmin = a.right_box[0], rmax = a.right_box[1];
    // utils/geo_util.py:104-114 with per_axis: 2 (v - 0.5 (max + min)) / (max - min), first coordinate
    const float nl = 2.0f * (x - 0.5f * (lmax + lmin)) / (lmax - lmin);
    const float nr = 2.0f * (x - 0.5f * (rmax + rmin)) / (rmax - rmin);
    float wl = 1.0f / (1.0f + __expf(-2.5f * (nl + 2.0f)));
    float wr = 1.0f / (1.0f + __expf(2.5f * (nr - 2.0f)));
    if (y < a.centre[1]) { wl = 0.f; wr = 0.f; }
    const float s = fmaxf(wl + wr, 1.0f);
    const float w = wl / s + wr / s;
    const float u = 1.0f - w;
#pragma unroll
    for (int c = 0; c < 3; ++c) a.positions[3 * i + c] = w * a.hand_positions[3 * i + c] + u * a.positions[3 * i + c];
    a.opacity[i] = w * a.hand_opacity[i] + u * a.opacity[i];
#pragma unroll
    for (int c = 0; c < 3; ++c) a.scales[3 * i + c] = w * a.hand_scales[3 * i + c] + u * a.scales[3 * i + c];
#pragma unroll
    for (int c = 0; c < 4; ++c) a.rotations[4 * i + c] = w * a.hand_rotations[4 * i + c] + u * a.rotations[4 * i + c];
}

extern "C" {

int ag_hand_fuse(const AgHandFuseArgs* a, void* stream)
{
    if (!a || a->N < 0) { set_error("bad hand-fuse sizes"); return AG_ERR_INVALID_ARGUMENT; }
    if (a->N == 0) return AG_OK;
    if (!a->xyz || !a->left_box || !a->right_box || !a->centre || !a->hand_positions || !a->hand_opacity || !a->hand_scales ||
        !a->hand_rotations || !a->positions || !a->opacity || !a->scales || !a->rotations) {
        set_error("null pointer in AgHandFuseArgs");
        return AG_ERR_INVALID_ARGUMENT;
    }
    hipLaunchKernelGGL(hand_fuse_kernel, dim3((a->N + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), *a);
    return check_hip(hipGetLastError(), "hand_fuse_kernel");
}

static int check_gather(const AgGatherArgs* a)
{
    if (!a || a->N < 0 || a->S <= 0) { set_error("bad gather sizes"); return AG_ERR_INVALID_ARGUMENT; }
    if (a->N == 0) return AG_OK;
    const bool others_all = a->opacity && a->scales && a->rotations, others_none = !a->opacity && !a->scales && !a->rotations;
    if (!a->pix || (a->positions && !a->xyz) || !(others_all || others_none) ||
        (others_all && (!a->opacity_raw || !a->scaling_raw || !a->rotation_raw)) || (!a->positions && others_none && !a->colors)) {
        set_error("AgGatherArgs: pix, at least one output part, and the canonical parameters of every requested part are required");
        return AG_ERR_INVALID_ARGUMENT;
    }
    return AG_OK;
}

int ag_gather_activate_forward(const AgGatherArgs* a, void* stream)
{
    int rc = check_gather(a);
    if (rc || a->N == 0) return rc;
    hipLaunchKernelGGL(gather_forward_kernel, dim3((a->N + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), *a);
    return check_hip(hipGetLastError(), "gather_forward_kernel");
}

int ag_gather_activate_backward(const AgGatherArgs* a, float* g_pos_map, float* g_other_map, float* g_col_map, void* stream)
{
    if (!a || a->N < 0 || a->S <= 0) { set_error("bad gather sizes"); return AG_ERR_INVALID_ARGUMENT; }
    if (!g_pos_map && !g_other_map && !g_col_map) { set_error("no gradient map requested"); return AG_ERR_INVALID_ARGUMENT; }
    if (a->N > 0 && (!a->pix || (g_pos_map && !a->positions) || (g_col_map && !a->colors) ||
                     (g_other_map && (!a->other_map || !a->opacity_raw || !a->scaling_raw || !a->rotation_raw || !a->opacity ||
                                      !a->scales || !a->rotations)))) {
        set_error("AgGatherArgs: a requested gradient map needs its upstream gradients (and, for the other map, the forward inputs)");
        return AG_ERR_INVALID_ARGUMENT;
    }
    int rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t plane = (size_t)a->S * a->S * sizeof(float);
    if (g_pos_map && (rc = check_hip(hipMemsetAsync(g_pos_map, 0, 6 * plane, s), "memset"))) return rc;
    if (g_other_map && (rc = check_hip(hipMemsetAsync(g_other_map, 0, 16 * plane, s), "memset"))) return rc;
    if (g_col_map && (rc = check_hip(hipMemsetAsync(g_col_map, 0, 6 * plane, s), "memset"))) return rc;
    if (a->N == 0) return AG_OK;
    hipLaunchKernelGGL(gather_backward_kernel, dim3((a->N + 255) / 256), dim3(256), 0, s, *a, g_pos_map, g_other_map, g_col_map);
    return check_hip(hipGetLastError(), "gather_backward_kernel");
}

static int check_lbs(const AgLbsArgs* a)
{
    if (!a || a->N < 0 || a->J < 1 || a->J > kMaxJ) { set_error("bad lbs sizes (1 <= J <= %d)", kMaxJ); return AG_ERR_INVALID_ARGUMENT; }
    if (a->N == 0) return AG_OK;
    const bool sparse = a->K > 0;
    if ((sparse ? (!a->sp_idx || !a->sp_w || a->K > a->J) : !a->lbs) || !a->jnt_mats || !a->positions || !a->rotations || !a->out_positions ||
        !a->out_rotations) {
        set_error("null pointer in AgLbsArgs (dense: lbs; sparse, K > 0: sp_idx + sp_w, K <= J)");
        return AG_ERR_INVALID_ARGUMENT;
    }
    return AG_OK;
}

static int lbs_lds_bytes(int J, const void* fn)
{
    const int bytes = 4 * 64 * J * (int)sizeof(float);
    if (bytes > 48 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    return bytes;
}

int ag_lbs_forward(const AgLbsArgs* a, void* stream)
{
    int rc = check_lbs(a);
    if (rc || a->N == 0) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (a->K > 0) {
        hipLaunchKernelGGL(lbs_forward_kernel<true>, dim3((a->N + 255) / 256), dim3(256), 0, s, *a);
    } else {
        const int lds = lbs_lds_bytes(a->J, reinterpret_cast<const void*>(&lbs_forward_kernel<false>));
        hipLaunchKernelGGL(lbs_forward_kernel<false>, dim3((a->N + 255) / 256), dim3(256), lds, s, *a);
    }
    return check_hip(hipGetLastError(), "lbs_forward_kernel");
}

int ag_lbs_backward(const AgLbsArgs* a, float* g_positions, float* g_rotations, void* stream)
{
    int rc = check_lbs(a);
    if (rc || a->N == 0) return rc;
    if (!g_positions || !g_rotations) { set_error("null gradient output"); return AG_ERR_INVALID_ARGUMENT; }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (a->K > 0) {
        hipLaunchKernelGGL(lbs_backward_kernel<true>, dim3((a->N + 255) / 256), dim3(256), 0, s, *a, g_positions, g_rotations);
    } else {
        const int lds = lbs_lds_bytes(a->J, reinterpret_cast<const void*>(&lbs_backward_kernel<false>));
        hipLaunchKernelGGL(lbs_backward_kernel<false>, dim3((a->N + 255) / 256), dim3(256), lds, s, *a, g_positions, g_rotations);
    }
    return check_hip(hipGetLastError(), "lbs_backward_kernel");
}

}  // extern "C"
