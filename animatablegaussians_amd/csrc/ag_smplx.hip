// SMPL-X body-model forward for B poses of one subject (gfx950).  See include/ag_smplx.h for the reference lines each
// stage follows.  All of it is HBM / latency bound and tiny next to the render path -- the point of having it on the
// device is that `cano2live_jnt_mats` is produced where it is consumed (no CPU data-loader stage, no upload) and that the
// three model evaluations a data item needs (live, canonical, live without root) read the 61-MB pose-corrective basis
// once.  Three launches per call:
//   shape_kernel   v_shaped = v_template + shapedirs . components           thread per (pose, coordinate)
//   chain_kernel   rest joints, Rodrigues, pose features, kinematic chain, A  one wave per pose, level-synchronous over the tree
//                  (rest joints = J_regressor . v_shaped is linear in the components: J_regressor . v_template and
//                  J_regressor . shapedirs are folded once per model by ag_smplx_prepare -> 60 FMAs per joint instead of a
//                  10475-long reduction per joint and pose)
//   skin_kernel    pose_offsets = features . posedirs (the 61 MB stream), v_posed, T = W . A, vertices
//                  workgroup = 32 vertices x 8 slices of the 486 features; lanes along coordinates (256-B segments)
#include "ag_common.h"
#include "../../include/ag_smplx.h"

namespace ag {

constexpr int kMaxJoints = 64;
constexpr int kSkinVerts = 32;                 // vertices per workgroup of skin_kernel
constexpr int kSkinCoords = 3 * kSkinVerts;    // 96 coordinates = 384 B of every posedirs row
constexpr int kSkinSlices = 8;                 // the feature loop is dealt over 8 slices
constexpr int kSkinThreads = kSkinCoords * kSkinSlices;   // 768

__global__ void __launch_bounds__(256) smplx_shape_kernel(float* __restrict__ v_shaped, const float* __restrict__ v_template,
                                                         const float* __restrict__ shapedirs, const float* __restrict__ comps,
                                                         int n_coord, int NB)
{
    extern __shared__ float s_comp[];
    const int b = blockIdx.y;
    for (int l = threadIdx.x; l < NB; l += 256) s_comp[l] = comps[(size_t)b * NB + l];
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_coord) return;
    const float* row = shapedirs + (size_t)i * NB;
    float acc = 0.f;
    for (int l = 0; l < NB; ++l) acc = fmaf(s_comp[l], row[l], acc);
    v_shaped[(size_t)b * n_coord + i] = v_template[i] + acc;
}

// out[(j * 3 + c) * out_stride + b] = sum_v J_regressor[j][v] * src[(3 v + c) * src_stride + b]   (once per model: folds
// the regressor into v_template (stride 1, one column) and into every column of shapedirs (stride NB, NB columns))
__global__ void __launch_bounds__(256) smplx_joints_kernel(float* __restrict__ out, const float* __restrict__ J_regressor,
                                                          const float* __restrict__ src, int V, int src_stride, int out_stride)
{
    __shared__ float s_part[4][3];
    const int j = blockIdx.x, b = blockIdx.y;
    const float* reg = J_regressor + (size_t)j * V;
    const float* vs = src + b;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int v = threadIdx.x; v < V; v += 256) {
        const float w = reg[v];
        a0 = fmaf(w, vs[(size_t)(3 * v + 0) * src_stride], a0);
        a1 = fmaf(w, vs[(size_t)(3 * v + 1) * src_stride], a1);
        a2 = fmaf(w, vs[(size_t)(3 * v + 2) * src_stride], a2);
    }
    for (int o = 32; o > 0; o >>= 1) {
        a0 += __shfl_xor(a0, o);
        a1 += __shfl_xor(a1, o);
        a2 += __shfl_xor(a2, o);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_part[wave][0] = a0; s_part[wave][1] = a1; s_part[wave][2] = a2; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        out[(size_t)(j * 3 + c) * out_stride + b] = (s_part[0][c] + s_part[1][c]) + (s_part[2][c] + s_part[3][c]);
    }
}

struct Affine {   // [R | t], last row (0 0 0 1) implied
    float r[9], t[3];
};

// lbs.py:299-330.  The `+ 1e-8` goes into the norm only, the direction divides the untouched vector by it.
__device__ __forceinline__ void rodrigues(const float* rv, float* R)
{
    const float ex = rv[0] + 1e-8f, ey = rv[1] + 1e-8f, ez = rv[2] + 1e-8f;
    const float angle = sqrtf(ex * ex + ey * ey + ez * ez);
    const float rx = rv[0] / angle, ry = rv[1] / angle, rz = rv[2] / angle;
    const float s = sinf(angle), c1 = 1.f - cosf(angle);
    const float K[9] = {0.f, -rz, ry, rz, 0.f, -rx, -ry, rx, 0.f};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            float kk = 0.f;
            for (int k = 0; k < 3; ++k) kk = fmaf(K[3 * r + k], K[3 * k + c], kk);
            R[3 * r + c] = (r == c ? 1.f : 0.f) + s * K[3 * r + c] + c1 * kk;
        }
}

__global__ void __launch_bounds__(64) smplx_chain_kernel(float* __restrict__ A_out, float* __restrict__ A_skin,
                                                        float* __restrict__ joints_out, float* __restrict__ pose_feature,
                                                        const float* __restrict__ full_pose, const float* __restrict__ comps,
                                                        const float* __restrict__ joint_template, const float* __restrict__ joint_dirs,
                                                        const int* __restrict__ parents, const float* __restrict__ transl, int J, int NB)
{
    __shared__ Affine s_glob[kMaxJoints];
    __shared__ float s_rest[kMaxJoints][3];
    __shared__ int s_maxdepth;
    const int b = blockIdx.x, j = threadIdx.x;
    const bool on = j < J;
    if (j == 0) s_maxdepth = 0;
    __syncthreads();

    Affine loc;
    float rest[3] = {0.f, 0.f, 0.f};
    int parent = -1, depth = 0;
    if (on) {
        // lbs.py:208-212 with the regressor folded: J_regressor . (v_template + dirs . comps) = joint_template + joint_dirs . comps
        for (int c = 0; c < 3; ++c) {
            const float* d = joint_dirs + (size_t)(3 * j + c) * NB;
            float acc = 0.f;
            for (int l = 0; l < NB; ++l) acc = fmaf(d[l], comps[(size_t)b * NB + l], acc);
            rest[c] = joint_template[3 * j + c] + acc;
            s_rest[j][c] = rest[c];
        }
    }
    __syncthreads();
    if (on) {
        rodrigues(full_pose + ((size_t)b * J + j) * 3, loc.r);
        parent = parents[j];
        for (int c = 0; c < 3; ++c) loc.t[c] = parent >= 0 ? rest[c] - s_rest[parent][c] : rest[c];
        if (j >= 1) {
            float* pf = pose_feature + (size_t)b * 9 * (J - 1) + 9 * (j - 1);
            for (int k = 0; k < 9; ++k) pf[k] = loc.r[k] - ((k == 0 || k == 4 || k == 8) ? 1.f : 0.f);
        }
        for (int p = parent; p >= 0; p = parents[p]) ++depth;
        atomicMax(&s_maxdepth, depth);
        if (depth == 0) s_glob[j] = loc;
    }
    __syncthreads();
    const int maxdepth = s_maxdepth;
    // lbs.py:389-395: transform_chain[i] = transform_chain[parents[i]] @ transforms_mat[i], one tree level per step
    for (int d = 1; d <= maxdepth; ++d) {
        if (on && depth == d) {
            const Affine& P = s_glob[parent];
            Affine g;
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c)
                    g.r[3 * r + c] = fmaf(P.r[3 * r + 2], loc.r[6 + c], fmaf(P.r[3 * r + 1], loc.r[3 + c], P.r[3 * r] * loc.r[c]));
                g.t[r] = fmaf(P.r[3 * r + 2], loc.t[2], fmaf(P.r[3 * r + 1], loc.t[1], P.r[3 * r] * loc.t[0])) + P.t[r];
            }
            s_glob[j] = g;
        }
        __syncthreads();
    }
    if (!on) return;
    const Affine g = s_glob[j];
    float* Aj = A_out + ((size_t)b * J + j) * 16;
    float* As = A_skin + ((size_t)b * J + j) * 12;
    for (int r = 0; r < 3; ++r) {
        // lbs.py:402-403: rel = T - pad(T @ [J; 0]) -> translation column minus R . J_rest
        const float rj = fmaf(g.r[3 * r + 2], rest[2], fmaf(g.r[3 * r + 1], rest[1], g.r[3 * r] * rest[0]));
        const float tr = transl ? transl[3 * b + r] : 0.f;
        for (int c = 0; c < 3; ++c) Aj[4 * r + c] = As[4 * r + c] = g.r[3 * r + c];
        As[4 * r + 3] = g.t[r] - rj;                     // what the skinning uses (lbs.py:241)
        // body_models.py:1272-1275: joints += transl; A[:, :, :3, 3] += transl -- AFTER the skinning used the un-translated A
        Aj[4 * r + 3] = transl ? (g.t[r] - rj) + tr : g.t[r] - rj;
        joints_out[((size_t)b * J + j) * 3 + r] = transl ? g.t[r] + tr : g.t[r];
    }
    Aj[12] = 0.f; Aj[13] = 0.f; Aj[14] = 0.f; Aj[15] = 1.f;
}

template <int NBATCH>
__global__ void __launch_bounds__(kSkinThreads) smplx_skin_kernel(float* __restrict__ vertices, const float* __restrict__ posedirs,
                                                                 const float* __restrict__ pose_feature,
                                                                 const float* __restrict__ v_shaped, const float* __restrict__ A,
                                                                 const float* __restrict__ lbs_weights,
                                                                 const float* __restrict__ transl, int V, int J, int P)
{
    extern __shared__ float smem[];
    float* s_feat = smem;                                   // [NBATCH][P]
    float* s_A = s_feat + NBATCH * P;                       // [NBATCH][J][12]
    float* s_part = s_A + NBATCH * J * 12;                  // [kSkinSlices][NBATCH][kSkinCoords]
    float* s_vp = s_part + kSkinSlices * NBATCH * kSkinCoords;   // [NBATCH][kSkinCoords]
    const int tid = threadIdx.x;
    const int n_coord = 3 * V;
    for (int i = tid; i < NBATCH * P; i += kSkinThreads) s_feat[i] = pose_feature[i];
    for (int i = tid; i < NBATCH * J * 12; i += kSkinThreads) s_A[i] = A[i];     // the un-translated affine rows (chain_kernel)
    __syncthreads();

    const int cx = tid % kSkinCoords, slice = tid / kSkinCoords;
    const int coord = blockIdx.x * kSkinCoords + cx;
    const bool live = coord < n_coord;
    float acc[NBATCH];
    for (int b = 0; b < NBATCH; ++b) acc[b] = 0.f;
    if (live) {
        const float* col = posedirs + coord;
        int p = slice;
        // four rows in flight per thread (the loop is latency bound: 61 MB over ~330 workgroups)
        for (; p + 3 * kSkinSlices < P; p += 4 * kSkinSlices) {
            const float d0 = col[(size_t)p * n_coord];
            const float d1 = col[(size_t)(p + kSkinSlices) * n_coord];
            const float d2 = col[(size_t)(p + 2 * kSkinSlices) * n_coord];
            const float d3 = col[(size_t)(p + 3 * kSkinSlices) * n_coord];
            for (int b = 0; b < NBATCH; ++b) {
                const float* f = s_feat + b * P + p;
                acc[b] = fmaf(f[3 * kSkinSlices], d3, fmaf(f[2 * kSkinSlices], d2, fmaf(f[kSkinSlices], d1, fmaf(f[0], d0, acc[b]))));
            }
        }
        for (; p < P; p += kSkinSlices) {
            const float d0 = col[(size_t)p * n_coord];
            for (int b = 0; b < NBATCH; ++b) acc[b] = fmaf(s_feat[b * P + p], d0, acc[b]);
        }
    }
    for (int b = 0; b < NBATCH; ++b) s_part[(slice * NBATCH + b) * kSkinCoords + cx] = acc[b];
    __syncthreads();

    const bool worker = tid < NBATCH * kSkinCoords;
    const int wb = tid / kSkinCoords;      // pose handled by this thread in the tail (cx is unchanged)
    if (worker && live) {
        float off = 0.f;
        for (int s = 0; s < kSkinSlices; ++s) off += s_part[(s * NBATCH + wb) * kSkinCoords + cx];
        s_vp[wb * kSkinCoords + cx] = off + v_shaped[(size_t)wb * n_coord + coord];     // lbs.py:233
    }
    __syncthreads();
    if (worker && live) {
        const int lv = cx / 3, c = cx % 3;
        const float* w = lbs_weights + (size_t)(blockIdx.x * kSkinVerts + lv) * J;
        const float* Ab = s_A + wb * J * 12 + 4 * c;
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
        for (int j = 0; j < J; ++j) {       // lbs.py:241-242: row c of T = W @ A
            const float wj = w[j];
            t0 = fmaf(wj, Ab[12 * j + 0], t0);
            t1 = fmaf(wj, Ab[12 * j + 1], t1);
            t2 = fmaf(wj, Ab[12 * j + 2], t2);
            t3 = fmaf(wj, Ab[12 * j + 3], t3);
        }
        const float* vp = s_vp + wb * kSkinCoords + 3 * lv;
        // the reference skins with the un-translated A and adds transl to the vertices afterwards (body_models.py:1274)
        float out = fmaf(t2, vp[2], fmaf(t1, vp[1], t0 * vp[0])) + t3;
        if (transl) out += transl[3 * wb + c];
        vertices[(size_t)wb * n_coord + coord] = out;
    }
}

__global__ void __launch_bounds__(64) smplx_keypoints_kernel(float* __restrict__ out, const float* __restrict__ vertices,
                                                            const int* __restrict__ idx, const float* __restrict__ w, int V, int K)
{
    const int i = blockIdx.x * 64 + threadIdx.x;   // (k, c)
    const int b = blockIdx.y;
    if (i >= 3 * K) return;
    const int k = i / 3, c = i % 3;
    const float* vb = vertices + (size_t)b * V * 3;
    float acc = 0.f;
    for (int t = 0; t < 3; ++t) acc = fmaf(w[3 * k + t], vb[3 * idx[3 * k + t] + c], acc);
    out[((size_t)b * K + k) * 3 + c] = acc;
}

__global__ void __launch_bounds__(64) mat4_mul_inverse_kernel(float* __restrict__ out, const float* __restrict__ a,
                                                             const float* __restrict__ bm, int n, int b_batch)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    float m[16], inv[16];
    for (int k = 0; k < 16; ++k) m[k] = bm[(size_t)(i % b_batch) * 16 + k];
    // general 4x4 inverse through the 2x2 minors of the top and bottom row pairs (adjugate / determinant)
    const float s0 = m[0] * m[5] - m[4] * m[1], s1 = m[0] * m[6] - m[4] * m[2], s2 = m[0] * m[7] - m[4] * m[3];
    const float s3 = m[1] * m[6] - m[5] * m[2], s4 = m[1] * m[7] - m[5] * m[3], s5 = m[2] * m[7] - m[6] * m[3];
    const float c5 = m[10] * m[15] - m[14] * m[11], c4 = m[9] * m[15] - m[13] * m[11], c3 = m[9] * m[14] - m[13] * m[10];
    const float c2 = m[8] * m[15] - m[12] * m[11], c1 = m[8] * m[14] - m[12] * m[10], c0 = m[8] * m[13] - m[12] * m[9];
    const float det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
    const float id = 1.f / det;
    inv[0] = (m[5] * c5 - m[6] * c4 + m[7] * c3) * id;
    inv[1] = (-m[1] * c5 + m[2] * c4 - m[3] * c3) * id;
    inv[2] = (m[13] * s5 - m[14] * s4 + m[15] * s3) * id;
    inv[3] = (-m[9] * s5 + m[10] * s4 - m[11] * s3) * id;
    inv[4] = (-m[4] * c5 + m[6] * c2 - m[7] * c1) * id;
    inv[5] = (m[0] * c5 - m[2] * c2 + m[3] * c1) * id;
    inv[6] = (-m[12] * s5 + m[14] * s2 - m[15] * s1) * id;
    inv[7] = (m[8] * s5 - m[10] * s2 + m[11] * s1) * id;
    inv[8] = (m[4] * c4 - m[5] * c2 + m[7] * c0) * id;
    inv[9] = (-m[0] * c4 + m[1] * c2 - m[3] * c0) * id;
    inv[10] = (m[12] * s4 - m[13] * s2 + m[15] * s0) * id;
    inv[11] = (-m[8] * s4 + m[9] * s2 - m[11] * s0) * id;
    inv[12] = (-m[4] * c3 + m[5] * c1 - m[6] * c0) * id;
    inv[13] = (m[0] * c3 - m[1] * c1 + m[2] * c0) * id;
    inv[14] = (-m[12] * s3 + m[13] * s1 - m[14] * s0) * id;
    inv[15] = (m[8] * s3 - m[9] * s1 + m[10] * s0) * id;
    const float* ai = a + (size_t)i * 16;
    float* oi = out + (size_t)i * 16;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            float acc = 0.f;
            for (int k = 0; k < 4; ++k) acc = fmaf(ai[4 * r + k], inv[4 * k + c], acc);
            oi[4 * r + c] = acc;
        }
}

static bool model_ok(const AgSmplxModel* m)
{
    return m && m->V > 0 && m->J > 0 && m->J <= kMaxJoints && m->NB >= 0 && m->NB <= 4096 && m->v_template && m->posedirs &&
           m->J_regressor && m->parents && m->lbs_weights && (m->NB == 0 || m->shapedirs);
}

static bool folded_ok(const AgSmplxModel* m)
{
    return m->joint_template && (m->NB == 0 || m->joint_dirs);
}

template <int NBATCH>
static void launch_skin(const AgSmplxModel* m, float* vertices, const float* feat, const float* v_shaped, const float* A,
                        const float* transl, hipStream_t s)
{
    const int P = 9 * (m->J - 1);
    const size_t lds = sizeof(float) * ((size_t)NBATCH * P + (size_t)NBATCH * m->J * 12 + (size_t)kSkinSlices * NBATCH * kSkinCoords +
                                        (size_t)NBATCH * kSkinCoords);
    const int grid = (m->V + kSkinVerts - 1) / kSkinVerts;
    hipLaunchKernelGGL(smplx_skin_kernel<NBATCH>, dim3(grid), dim3(kSkinThreads), lds, s, vertices, m->posedirs, feat, v_shaped, A,
                       m->lbs_weights, transl, m->V, m->J, P);
}

}  // namespace ag

using namespace ag;

extern "C" {

size_t ag_smplx_workspace_floats(const AgSmplxModel* m, int32_t B)
{
    if (!m || B <= 0) return 0;
    return (size_t)B * ((size_t)3 * m->V + (size_t)12 * m->J + (size_t)9 * (m->J - 1));
}

int ag_smplx_forward(const AgSmplxModel* m, int32_t B, const float* shape_components, const float* full_pose, const float* transl,
                     float* vertices, float* joints, float* A, float* workspace, size_t workspace_floats, void* stream)
{
    if (!model_ok(m)) { set_error("smplx: bad model (need 0 < J <= 64, non-null arrays)"); return AG_ERR_INVALID_ARGUMENT; }
    if (!folded_ok(m)) { set_error("smplx: joint_template / joint_dirs missing -- run ag_smplx_prepare once per model"); return AG_ERR_INVALID_ARGUMENT; }
    if (B < 0) { set_error("smplx: B < 0"); return AG_ERR_INVALID_ARGUMENT; }
    if (B == 0) return AG_OK;
    if (!full_pose || !vertices || !joints || !A || !workspace || (m->NB > 0 && !shape_components)) {
        set_error("smplx: null pointer");
        return AG_ERR_INVALID_ARGUMENT;
    }
    if (workspace_floats < ag_smplx_workspace_floats(m, B)) { set_error("smplx: workspace too small"); return AG_ERR_SCRATCH_TOO_SMALL; }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int n_coord = 3 * m->V, P = 9 * (m->J - 1);
    float* v_shaped = workspace;
    float* A_skin = v_shaped + (size_t)B * n_coord;
    float* feat = A_skin + (size_t)B * 12 * m->J;

    hipLaunchKernelGGL(smplx_shape_kernel, dim3((n_coord + 255) / 256, B), dim3(256), sizeof(float) * (m->NB > 0 ? m->NB : 1), s, v_shaped,
                       m->v_template, m->shapedirs, shape_components, n_coord, m->NB);
    hipLaunchKernelGGL(smplx_chain_kernel, dim3(B), dim3(64), 0, s, A, A_skin, joints, feat, full_pose, shape_components, m->joint_template,
                       m->joint_dirs, m->parents, transl, m->J, m->NB);
    for (int b0 = 0; b0 < B; b0 += 4) {
        const int nb = B - b0 < 4 ? B - b0 : 4;
        float* vo = vertices + (size_t)b0 * n_coord;
        const float* f = feat + (size_t)b0 * P;
        const float* vs = v_shaped + (size_t)b0 * n_coord;
        const float* Ab = A_skin + (size_t)b0 * m->J * 12;
        const float* tb = transl ? transl + (size_t)3 * b0 : nullptr;
        switch (nb) {
            case 1: launch_skin<1>(m, vo, f, vs, Ab, tb, s); break;
            case 2: launch_skin<2>(m, vo, f, vs, Ab, tb, s); break;
            case 3: launch_skin<3>(m, vo, f, vs, Ab, tb, s); break;
            default: launch_skin<4>(m, vo, f, vs, Ab, tb, s); break;
        }
    }
    return check_hip(hipGetLastError(), "ag_smplx_forward");
}

int ag_smplx_prepare(const AgSmplxModel* m, float* joint_template, float* joint_dirs, void* stream)
{
    if (!model_ok(m)) { set_error("smplx_prepare: bad model"); return AG_ERR_INVALID_ARGUMENT; }
    if (!joint_template || (m->NB > 0 && !joint_dirs)) { set_error("null pointer"); return AG_ERR_INVALID_ARGUMENT; }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(smplx_joints_kernel, dim3(m->J, 1), dim3(256), 0, s, joint_template, m->J_regressor, m->v_template, m->V, 1, 1);
    if (m->NB > 0)
        hipLaunchKernelGGL(smplx_joints_kernel, dim3(m->J, m->NB), dim3(256), 0, s, joint_dirs, m->J_regressor, m->shapedirs, m->V, m->NB, m->NB);
    return check_hip(hipGetLastError(), "smplx_joints_kernel");
}

int ag_smplx_shape(const AgSmplxModel* m, int32_t B, const float* shape_components, float* v_shaped, void* stream)
{
    if (!model_ok(m) || B < 0) { set_error("smplx_shape: bad model or B < 0"); return AG_ERR_INVALID_ARGUMENT; }
    if (B == 0) return AG_OK;
    if (!v_shaped || (m->NB > 0 && !shape_components)) { set_error("null pointer"); return AG_ERR_INVALID_ARGUMENT; }
    const int n_coord = 3 * m->V;
    hipLaunchKernelGGL(smplx_shape_kernel, dim3((n_coord + 255) / 256, B), dim3(256), sizeof(float) * (m->NB > 0 ? m->NB : 1),
                       reinterpret_cast<hipStream_t>(stream), v_shaped, m->v_template, m->shapedirs, shape_components, n_coord, m->NB);
    return check_hip(hipGetLastError(), "smplx_shape_kernel");
}

int ag_mat4_mul_inverse(float* out, const float* a, const float* b, int32_t n, int32_t b_batch, void* stream)
{
    if (n < 0 || b_batch <= 0) { set_error("mat4_mul_inverse: need n >= 0, b_batch >= 1"); return AG_ERR_INVALID_ARGUMENT; }
    if (n == 0) return AG_OK;
    if (!out || !a || !b) { set_error("null pointer"); return AG_ERR_INVALID_ARGUMENT; }
    hipLaunchKernelGGL(mat4_mul_inverse_kernel, dim3((n + 63) / 64), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), out, a, b, n, b_batch);
    return check_hip(hipGetLastError(), "mat4_mul_inverse_kernel");
}

int ag_smplx_keypoints(float* out, const float* vertices, const int32_t* idx, const float* w, int32_t B, int32_t V, int32_t K, void* stream)
{
    if (B < 0 || K < 0 || V <= 0) { set_error("smplx_keypoints: bad sizes"); return AG_ERR_INVALID_ARGUMENT; }
    if (B == 0 || K == 0) return AG_OK;
    if (!out || !vertices || !idx || !w) { set_error("null pointer"); return AG_ERR_INVALID_ARGUMENT; }
    hipLaunchKernelGGL(smplx_keypoints_kernel, dim3((3 * K + 63) / 64, B), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), out, vertices,
                       idx, w, V, K);
    return check_hip(hipGetLastError(), "smplx_keypoints_kernel");
}

}  // extern "C"
