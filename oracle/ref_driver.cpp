// C entry points around the REFERENCE's own CudaRasterizer::Rasterizer (compiled for the CPU through
// oracle/cuda_cpu).  TEST INFRASTRUCTURE ONLY: built into oracle/_ref/libref_raster.so by oracle/ref_build.py,
// used by tests/test_oracle_vs_ref.py and oracle/make_golden.py to pin oracle/raster_oracle.c.
//
// A handle owns the three scratch chunks the reference's torch wrapper would own (rasterize_points.cu:73-80); the
// state accessors decode them with the reference's own GeometryState/BinningState/ImageState::fromChunk.
#include <cstring>
#include <functional>
#include <vector>

#include "rasterizer_impl.h"   // from the reference tree (include path set by ref_build.py)

namespace {
struct Handle {
    std::vector<char> geom, binning, img;
    int P = 0, W = 0, H = 0, R = 0;
};
std::function<char*(size_t)> resizer(std::vector<char>& v)
{
    return [&v](size_t n) { v.assign(n + 256, 0); return v.data(); };
}
}  // namespace

extern "C" {

void* ref_create() { return new Handle(); }
void ref_destroy(void* h) { delete static_cast<Handle*>(h); }

int ref_forward(void* hv, int P, int W, int H, const float* bg, const float* means3D, const float* colors,
                const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                const float* cov3D_precomp, const float* view, const float* proj, const float* campos,
                float tan_fovx, float tan_fovy, float* out_color, float* out_depth, float* out_alpha, int* radii)
{
    Handle* h = static_cast<Handle*>(hv);
    h->P = P; h->W = W; h->H = H;
    h->R = CudaRasterizer::Rasterizer::forward(resizer(h->geom), resizer(h->binning), resizer(h->img), P, 0, 0, bg, W, H,
                                               means3D, nullptr, colors, opacities, scales, scale_modifier, rotations,
                                               cov3D_precomp, view, proj, campos, tan_fovx, tan_fovy, false, out_color,
                                               out_depth, out_alpha, radii, false);
    return h->R;
}

// Same through the spherical-harmonics colour path (colors_precomp = nullptr): D = active degree, M = coefficients per Gaussian.
int ref_forward_sh(void* hv, int P, int D, int M, int W, int H, const float* bg, const float* means3D, const float* shs,
                   const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                   const float* cov3D_precomp, const float* view, const float* proj, const float* campos,
                   float tan_fovx, float tan_fovy, float* out_color, float* out_depth, float* out_alpha, int* radii)
{
    Handle* h = static_cast<Handle*>(hv);
    h->P = P; h->W = W; h->H = H;
    h->R = CudaRasterizer::Rasterizer::forward(resizer(h->geom), resizer(h->binning), resizer(h->img), P, D, M, bg, W, H,
                                               means3D, shs, nullptr, opacities, scales, scale_modifier, rotations,
                                               cov3D_precomp, view, proj, campos, tan_fovx, tan_fovy, false, out_color,
                                               out_depth, out_alpha, radii, false);
    return h->R;
}

// rgb [P,3] and clamped [P,3] as the forward left them in the geometry chunk
void ref_get_colors(void* hv, float* rgb, uint8_t* clamped)
{
    Handle* h = static_cast<Handle*>(hv);
    char* g = h->geom.data();
    CudaRasterizer::GeometryState gs = CudaRasterizer::GeometryState::fromChunk(g, h->P);
    memcpy(rgb, gs.rgb, (size_t)h->P * 12);
    for (size_t i = 0; i < (size_t)h->P * 3; i++) clamped[i] = gs.clamped[i] ? 1 : 0;
}

void ref_backward_sh(void* hv, int D, int M, const float* bg, const float* means3D, const float* shs, const float* alphas,
                     const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                     const float* view, const float* proj, const float* campos, float tan_fovx, float tan_fovy,
                     const int* radii, const float* dL_dpix, const float* dL_dpix_depth, const float* dL_dalphas,
                     float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_ddepth,
                     float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot)
{
    Handle* h = static_cast<Handle*>(hv);
    CudaRasterizer::Rasterizer::backward(h->P, D, M, h->R, bg, h->W, h->H, means3D, shs, nullptr, alphas, scales,
                                         scale_modifier, rotations, cov3D_precomp, view, proj, campos, tan_fovx, tan_fovy,
                                         radii, h->geom.data(), h->binning.data(), h->img.data(), dL_dpix, dL_dpix_depth,
                                         dL_dalphas, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_ddepth, dL_dmean3D,
                                         dL_dcov3D, dL_dsh, dL_dscale, dL_drot, false);
}

// Copies the reference's internal state out of its chunks (sizes: P, P*2, P, P*6, P*4, P, P | R, R | T*2, W*H).
void ref_get_state(void* hv, float* depths, float* means2D, float* cov3D, float* conic_opacity, uint32_t* tiles_touched,
                   uint32_t* point_offsets, uint64_t* keys_sorted, uint32_t* point_list, uint32_t* ranges,
                   uint32_t* n_contrib)
{
    Handle* h = static_cast<Handle*>(hv);
    char* g = h->geom.data();
    CudaRasterizer::GeometryState gs = CudaRasterizer::GeometryState::fromChunk(g, h->P);
    char* b = h->binning.data();
    CudaRasterizer::BinningState bs = CudaRasterizer::BinningState::fromChunk(b, h->R);
    char* i = h->img.data();
    CudaRasterizer::ImageState is = CudaRasterizer::ImageState::fromChunk(i, (size_t)h->W * h->H);
    const size_t P = h->P, R = h->R, T = (size_t)((h->W + 15) / 16) * ((h->H + 15) / 16);
    memcpy(depths, gs.depths, P * 4);
    memcpy(means2D, gs.means2D, P * 8);
    memcpy(cov3D, gs.cov3D, P * 24);
    memcpy(conic_opacity, gs.conic_opacity, P * 16);
    memcpy(tiles_touched, gs.tiles_touched, P * 4);
    memcpy(point_offsets, gs.point_offsets, P * 4);
    memcpy(keys_sorted, bs.point_list_keys, R * 8);
    memcpy(point_list, bs.point_list, R * 4);
    memcpy(ranges, is.ranges, T * 8);
    memcpy(n_contrib, is.n_contrib, (size_t)h->W * h->H * 4);
}

void ref_backward(void* hv, const float* bg, const float* means3D, const float* colors, const float* alphas,
                  const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                  const float* view, const float* proj, const float* campos, float tan_fovx, float tan_fovy,
                  const int* radii, const float* dL_dpix, const float* dL_dpix_depth, const float* dL_dalphas,
                  float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_ddepth,
                  float* dL_dmean3D, float* dL_dcov3D, float* dL_dscale, float* dL_drot)
{
    Handle* h = static_cast<Handle*>(hv);
    CudaRasterizer::Rasterizer::backward(h->P, 0, 0, h->R, bg, h->W, h->H, means3D, nullptr, colors, alphas, scales,
                                         scale_modifier, rotations, cov3D_precomp, view, proj, campos, tan_fovx, tan_fovy,
                                         radii, h->geom.data(), h->binning.data(), h->img.data(), dL_dpix, dL_dpix_depth,
                                         dL_dalphas, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_ddepth, dL_dmean3D,
                                         dL_dcov3D, nullptr, dL_dscale, dL_drot, false);
}

void ref_mark_visible(int P, float* means3D, float* view, float* proj, bool* present)
{
    CudaRasterizer::Rasterizer::markVisible(P, means3D, view, proj, present);
}

}  // extern "C"
