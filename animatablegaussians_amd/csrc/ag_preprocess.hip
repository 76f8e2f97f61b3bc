// Forward per-Gaussian preprocess + per-tile instance counting (gfx950).
//
// Replaces preprocessCUDA<3> (reference cuda_rasterizer/forward.cu:155-256) and produces, instead of the
// reference's Gaussian-major `tiles_touched` prefix sum, per-TILE instance counts (one global atomic per
// (Gaussian, tile) instance).  The tile-major counts make the later binning a counting sort by tile, so no
// device-wide 64-bit radix sort is needed (see ag_binning.hip).
//
// THIS TRANSLATION UNIT IS COMPILED WITH -ffp-contract=off: the fp32 expression order below is the contract that
// makes radii / tile rects / depth keys bit-identical to the CPU oracle (oracle/raster_oracle.c), which in turn
// restates the reference's expression order (GLM column-major mat3 products, left-to-right sums).
// HBM-bound streaming kernel: 44 B in, 48 B record + 24 B cov3D + 8 B out per Gaussian.
#include "ag_common.h"
#include "ag_sh.h"

namespace ag {

struct M3 { float m[3][3]; };  // m[col][row], GLM convention

__device__ __forceinline__ M3 m3_cols(float a, float b, float c, float d, float e, float f, float g, float h, float i)
{
    M3 r;
    r.m[0][0] = a; r.m[0][1] = b; r.m[0][2] = c;
    r.m[1][0] = d; r.m[1][1] = e; r.m[1][2] = f;
    r.m[2][0] = g; r.m[2][1] = h; r.m[2][2] = i;
    return r;
}

__device__ __forceinline__ M3 m3_mul(const M3& a, const M3& b)
{
    M3 r;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int q = 0; q < 3; q++)
            r.m[c][q] = a.m[0][q] * b.m[c][0] + a.m[1][q] * b.m[c][1] + a.m[2][q] * b.m[c][2];
    return r;
}

__device__ __forceinline__ M3 m3_t(const M3& a)
{
    M3 r;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int q = 0; q < 3; q++) r.m[c][q] = a.m[q][c];
    return r;
}

__device__ __forceinline__ float ndc2pix(float v, int S)
{
    return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5);
}

__device__ __forceinline__ void get_rect(float px, float py, int max_radius, int gx, int gy,
                                         uint32_t& x0o, uint32_t& y0o, uint32_t& x1o, uint32_t& y1o)
{
    const float r = (float)max_radius;
    int x0 = (int)((px - r) / (float)kTileX);
    int y0 = (int)((py - r) / (float)kTileY);
    int x1 = (int)((px + r + (float)kTileX - 1.0f) / (float)kTileX);
    int y1 = (int)((py + r + (float)kTileY - 1.0f) / (float)kTileY);
    x0 = max(0, x0); y0 = max(0, y0); x1 = max(0, x1); y1 = max(0, y1);
    x0o = (uint32_t)min(gx, x0); y0o = (uint32_t)min(gy, y0);
    x1o = (uint32_t)min(gx, x1); y1o = (uint32_t)min(gy, y1);
}

constexpr int kWinBins = 2048;   // tile-window histogram bins per workgroup (8 KiB of LDS)

struct PreParams {
    int P, W, H, gx, gy;
    float tan_fovx, tan_fovy, focal_x, focal_y, scale_modifier;
    const float* __restrict__ means3D;
    const float* __restrict__ scales;
    const float* __restrict__ rotations;
    const float* __restrict__ opacities;
    const float* __restrict__ colors;       // precomputed colours, or NULL -> spherical harmonics below
    const float* __restrict__ shs;          // [P][sh_coeffs][3]
    const float* __restrict__ campos;
    uint8_t* __restrict__ clamped;          // [P][3], written on the SH path (the backward zeroes clamped channels' gradients)
    int sh_degree, sh_coeffs;
    const float* __restrict__ cov3D_precomp;
    const float* __restrict__ view;
    const float* __restrict__ proj;
    int* __restrict__ radii;
    GaussRec* __restrict__ rec;
    float* __restrict__ cov3Ds;
    uint32_t* __restrict__ tiles_touched;
    uint32_t* __restrict__ tile_count;
};

__global__ void __launch_bounds__(256) preprocess_kernel(PreParams p)
{
    // Workgroup-local tile histogram: the 256 Gaussians of a workgroup are neighbours on the canonical map, so
    // their tile rects fall into a small window of the tile grid.  Instances are counted with LDS atomics inside that
    // window and flushed with ONE global atomic per (workgroup, tile): ~20x fewer same-address global atomics than
    // one per instance (which serialise in L2 at ~5 M/s per address and cost 190 us at avatar scale).
    __shared__ int s_win[4];             // min x, min y, max x, max y (tile units, max exclusive)
    __shared__ uint32_t s_hist[kWinBins];
    // The 48-byte records and the 24-byte covariances leave through LDS: a thread's own record would be 12 + 6 dword stores at a
    // 48 / 24-byte lane stride (every store instruction touches 24 cache lines); staged, the workgroup writes its 12 + 6 KB as 16-byte
    // stores, lanes along addresses.  Round 2: 14.8 -> see DESIGN.md section 5.
    // (Round 5: the INPUT rows through LDS the same way -- 13 strided load instructions -> 4 contiguous ones + LDS reads, two more barriers --
    // measured 14.0 -> 15.3 us, profiles/r05b_ab_stage.txt: the loads are latency the 4 resident workgroups already overlap; not kept.)
    __shared__ __attribute__((aligned(16))) float s_rec[256 * 12];
    __shared__ __attribute__((aligned(16))) float s_cov[256 * 6];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x == 0) { s_win[0] = 0x7fffffff; s_win[1] = 0x7fffffff; s_win[2] = 0; s_win[3] = 0; }
    __syncthreads();

    // wave-uniform camera: scalar loads
    float V[16], Pm[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { V[i] = p.view[i]; Pm[i] = p.proj[i]; }

    int my_radii = 0;
    bool cov_set = false, rec_set = false;
    uint32_t touched = 0;
    uint32_t rx0 = 0, ry0 = 0, rx1 = 0, ry1 = 0;
    const bool valid = idx < p.P;
    const int sidx = valid ? idx : 0;
    const float ox = p.means3D[3 * sidx + 0], oy = p.means3D[3 * sidx + 1], oz = p.means3D[3 * sidx + 2];

    // near-plane cull only (auxiliary.h:154)
    const float vz = V[2] * ox + V[6] * oy + V[10] * oz + V[14];
    if (valid && vz > 0.2f) {
        const float hx = Pm[0] * ox + Pm[4] * oy + Pm[8] * oz + Pm[12];
        const float hy = Pm[1] * ox + Pm[5] * oy + Pm[9] * oz + Pm[13];
        const float hw = Pm[3] * ox + Pm[7] * oy + Pm[11] * oz + Pm[15];
        const float p_w = 1.0f / (hw + 0.0000001f);
        const float projx = hx * p_w, projy = hy * p_w;

        float c3[6];
        if (p.cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 6; k++) c3[k] = p.cov3D_precomp[6 * idx + k];
        } else {
            // computeCov3D, forward.cu:116-152 — quaternion used un-normalised
            const float mod = p.scale_modifier;
            M3 S = m3_cols(1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f);
            S.m[0][0] = mod * p.scales[3 * idx + 0];
            S.m[1][1] = mod * p.scales[3 * idx + 1];
            S.m[2][2] = mod * p.scales[3 * idx + 2];
            const float r = p.rotations[4 * idx + 0], x = p.rotations[4 * idx + 1];
            const float y = p.rotations[4 * idx + 2], z = p.rotations[4 * idx + 3];
            const M3 R = m3_cols(
                1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
            const M3 M = m3_mul(S, R);
            const M3 Sigma = m3_mul(m3_t(M), M);
            c3[0] = Sigma.m[0][0]; c3[1] = Sigma.m[0][1]; c3[2] = Sigma.m[0][2];
            c3[3] = Sigma.m[1][1]; c3[4] = Sigma.m[1][2]; c3[5] = Sigma.m[2][2];
        }
#pragma unroll
        for (int k = 0; k < 6; k++) s_cov[6 * threadIdx.x + k] = c3[k];
        cov_set = true;

        // computeCov2D, forward.cu:74-113
        float tx = V[0] * ox + V[4] * oy + V[8] * oz + V[12];
        float ty = V[1] * ox + V[5] * oy + V[9] * oz + V[13];
        const float tz = vz;
        const float limx = 1.3f * p.tan_fovx, limy = 1.3f * p.tan_fovy;
        const float txtz = tx / tz, tytz = ty / tz;
        tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
        ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
        const M3 J = m3_cols(
            p.focal_x / tz, 0.0f, -(p.focal_x * tx) / (tz * tz),
            0.0f, p.focal_y / tz, -(p.focal_y * ty) / (tz * tz),
            0.f, 0.f, 0.f);
        const M3 Wm = m3_cols(V[0], V[4], V[8], V[1], V[5], V[9], V[2], V[6], V[10]);
        const M3 T = m3_mul(Wm, J);
        const M3 Vrk = m3_cols(c3[0], c3[1], c3[2], c3[1], c3[3], c3[4], c3[2], c3[4], c3[5]);
        const M3 cov = m3_mul(m3_mul(m3_t(T), m3_t(Vrk)), T);
        const float cxx = cov.m[0][0] + 0.3f, cxy = cov.m[0][1], cyy = cov.m[1][1] + 0.3f;

        const float det = (cxx * cyy - cxy * cxy);
        if (det != 0.0f) {
            const float det_inv = 1.f / det;
            const float ca = cyy * det_inv, cb = -cxy * det_inv, cc = cxx * det_inv;
            const float mid = 0.5f * (cxx + cyy);
            const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
            const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
            const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
            const float px = ndc2pix(projx, p.W), py = ndc2pix(projy, p.H);
            uint32_t x0, y0, x1, y1;
            get_rect(px, py, (int)my_radius, p.gx, p.gy, x0, y0, x1, y1);
            const uint32_t n = (x1 - x0) * (y1 - y0);
            if (n != 0) {
                my_radii = (int)my_radius;
                touched = n;
                const float op = p.opacities[idx];
                GaussRec g;
                g.x = px; g.y = py; g.ca = ca; g.cb = cb; g.cc = cc; g.op = op;
                if (p.colors) {
                    g.r = p.colors[3 * idx + 0]; g.g = p.colors[3 * idx + 1]; g.b = p.colors[3 * idx + 2];
                } else {                                    // forward.cu:238-245: after the culls, only for Gaussians that are drawn
                    float rgb[3];
                    uint8_t cl[3];
                    sh_colour(p.sh_degree, sh_direction(ox, oy, oz, p.campos), p.shs + (size_t)idx * p.sh_coeffs * 3, rgb, cl);
                    g.r = rgb[0]; g.g = rgb[1]; g.b = rgb[2];
                    p.clamped[3 * idx + 0] = cl[0]; p.clamped[3 * idx + 1] = cl[1]; p.clamped[3 * idx + 2] = cl[2];
                }
                g.depth = vz;
                // Wave-level cull radius for the blend kernels.  power <= -0.5*d^2/lambda_max(cov2D) and
                // alpha = op*exp(power) < 1/255  <=>  power < -ln(255*op); lambda1 >= lambda_max (0.1 floor above).
                // 1 % + 0.01 slack in log space dwarfs every fp32 rounding of conic/power; huge or degenerate
                // splats are never culled.
                // qcut (round 5): the same threshold on the quadratic form itself, q = a dx^2 + 2 b dx dy + c dy^2 = -2 power: alpha < 1 / 255
                // wherever q > 2 ln(255 op).  The blend backward minimises q over its 4 x 4 pixels' rectangle exactly (ag_blend_backward.hip), which
                // needs a positive definite conic; any other conic is never culled by it.
                float r2 = 3.0e38f, qc = 3.0e38f;
                if (lambda1 < 1.0e4f && lambda1 > 0.f && op >= 0.f) {
                    const float lg = __logf(255.0f * op) + 0.01f;   // -inf for op == 0
                    r2 = (lg > 0.f) ? 2.0f * lambda1 * lg * 1.01f : -1.0f;
                    const bool pd = ca > 0.f && cc > 0.f && (ca * cc - cb * cb) > 0.f;
                    qc = (lg > 0.f) ? (pd ? 2.0f * lg * 1.01f : 3.0e38f) : -1.0f;
                }
                g.r2cut = r2;
                g.qcut = qc;
                *reinterpret_cast<GaussRec*>(s_rec + 12 * threadIdx.x) = g;
                rec_set = true;
                rx0 = x0; ry0 = y0; rx1 = x1; ry1 = y1;
            }
        }
    }
    if (valid) {
        p.radii[idx] = my_radii;
        p.tiles_touched[idx] = touched;
    }
    // culled Gaussians: their record / covariance is never read (no tile list names them); zeros keep the scratch deterministic
    if (!rec_set) {
#pragma unroll
        for (int k = 0; k < 12; k++) s_rec[12 * threadIdx.x + k] = 0.f;
    }
    if (!cov_set) {
#pragma unroll
        for (int k = 0; k < 6; k++) s_cov[6 * threadIdx.x + k] = 0.f;
    }
    window_accumulate(s_win, touched != 0, (int)rx0, (int)ry0, (int)rx1, (int)ry1);
    __syncthreads();
    {
        const int first = blockIdx.x * 256, nv = min(256, p.P - first);          // Gaussians of this workgroup
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4* drec = reinterpret_cast<f4*>(p.rec + first);                           // 48 B * 256 * blockIdx: 16-byte aligned
        const f4* srec = reinterpret_cast<const f4*>(s_rec);
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int i = threadIdx.x + 256 * j;
            if (i < nv * 3) drec[i] = srec[i];
        }
        float* dcov = p.cov3Ds + (size_t)first * 6;                                // 24 B * 256 * blockIdx: 16-byte aligned
        const int nf = nv * 6, nf4 = nf >> 2;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int i = threadIdx.x + 256 * j;
            if (i < nf4) reinterpret_cast<f4*>(dcov)[i] = reinterpret_cast<const f4*>(s_cov)[i];
        }
        if ((int)threadIdx.x < (nf & 3)) dcov[4 * nf4 + threadIdx.x] = s_cov[4 * nf4 + threadIdx.x];
    }
    const int wx0 = s_win[0], wy0 = s_win[1];
    const int bw = s_win[2] - wx0, bh = s_win[3] - wy0;
    if (bw <= 0 || bh <= 0) return;                       // nothing visible in this workgroup (uniform)
    const int nb = bw * bh;
    if (nb <= kWinBins) {
        for (int i = threadIdx.x; i < nb; i += 256) s_hist[i] = 0u;
        __syncthreads();
        for (uint32_t y = ry0; y < ry1; y++)
            for (uint32_t x = rx0; x < rx1; x++) atomicAdd(&s_hist[((int)y - wy0) * bw + ((int)x - wx0)], 1u);
        __syncthreads();
        for (int i = threadIdx.x; i < nb; i += 256) {
            const uint32_t c = s_hist[i];
            if (c) atomicAdd(&p.tile_count[(uint32_t)(wy0 + i / bw) * (uint32_t)p.gx + (uint32_t)(wx0 + i % bw)], c);
        }
    } else {
        // spatially incoherent input (e.g. randomly ordered Gaussians): plain per-instance atomics
        for (uint32_t y = ry0; y < ry1; y++)
            for (uint32_t x = rx0; x < rx1; x++) atomicAdd(&p.tile_count[y * (uint32_t)p.gx + x], 1u);
    }
}

int launch_preprocess(const AgRasterForwardArgs& a, hipStream_t s)
{
    PreParams p;
    p.P = a.P; p.W = a.W; p.H = a.H;
    p.gx = (a.W + kTileX - 1) / kTileX;
    p.gy = (a.H + kTileY - 1) / kTileY;
    p.tan_fovx = a.tan_fovx; p.tan_fovy = a.tan_fovy;
    p.focal_y = a.H / (2.0f * a.tan_fovy);   // rasterizer_impl.cu:223-224
    p.focal_x = a.W / (2.0f * a.tan_fovx);
    p.scale_modifier = a.scale_modifier;
    p.means3D = a.means3D; p.scales = a.scales; p.rotations = a.rotations; p.opacities = a.opacities;
    p.colors = a.colors_precomp; p.cov3D_precomp = a.cov3D_precomp;
    p.shs = a.shs; p.campos = a.campos; p.sh_degree = a.sh_degree; p.sh_coeffs = a.sh_coeffs;
    p.view = a.viewmatrix; p.proj = a.projmatrix;
    p.radii = a.radii;
    char* gb = aligned_base(a.geom_buffer);
    char* ib = aligned_base(a.image_buffer);
    GeomLayout gl((size_t)a.P);
    ImageLayout il((size_t)a.W, (size_t)a.H);
    p.rec = reinterpret_cast<GaussRec*>(gb + gl.rec);
    p.cov3Ds = reinterpret_cast<float*>(gb + gl.cov3d);
    p.tiles_touched = reinterpret_cast<uint32_t*>(gb + gl.tiles_touched);
    p.clamped = reinterpret_cast<uint8_t*>(gb + gl.clamped);
    p.tile_count = reinterpret_cast<uint32_t*>(ib + il.tile_count);
    const size_t T = (size_t)p.gx * p.gy;
    if (check_hip(hipMemsetAsync(p.tile_count, 0, T * sizeof(uint32_t), s), "memset tile_count")) return AG_ERR_HIP;
    { ProfScope ps(AG_K_PREPROCESS, s); hipLaunchKernelGGL(preprocess_kernel, dim3((a.P + 255) / 256), dim3(256), 0, s, p); }
    return check_hip(hipGetLastError(), "preprocess_kernel");
}

__global__ void __launch_bounds__(256) mark_visible_kernel(int P, const float* __restrict__ means3D,
                                                          const float* __restrict__ view, uint8_t* __restrict__ present)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const float ox = means3D[3 * idx + 0], oy = means3D[3 * idx + 1], oz = means3D[3 * idx + 2];
    const float vz = view[2] * ox + view[6] * oy + view[10] * oz + view[14];
    present[idx] = (vz <= 0.2f) ? 0 : 1;
}

int launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t s)
{
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, view, present);
    return check_hip(hipGetLastError(), "mark_visible_kernel");
}

}  // namespace ag
