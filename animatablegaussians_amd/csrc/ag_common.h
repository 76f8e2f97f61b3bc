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
    float pad;
};
static_assert(sizeof(GaussRec) == 48, "GaussRec must be 48 bytes");

// Per-Gaussian accumulator row of the blend backward: one 64-byte line.
constexpr int kAccumFloats = 16;
enum AccumSlot { A_M2X = 0, A_M2Y, A_CONX, A_CONY, A_CONW, A_OPAC, A_COLR, A_COLG, A_COLB, A_DEPTH };

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
    size_t ranges, tile_count, cursor, n_contrib, num_rendered, total;
    __host__ __device__ ImageLayout(size_t W, size_t H)
    {
        const size_t T = ((W + kTileX - 1) / kTileX) * ((H + kTileY - 1) / kTileY);
        size_t o = 0;
        ranges = o;        o = align_up(o + T * 2 * sizeof(uint32_t), 256);
        tile_count = o;    o = align_up(o + T * sizeof(uint32_t), 256);
        cursor = o;        o = align_up(o + T * sizeof(uint32_t), 256);
        n_contrib = o;     o = align_up(o + W * H * sizeof(uint32_t), 256);
        num_rendered = o;  o = align_up(o + 16, 256);
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
void prof_begin(int kernel_id, hipStream_t s);
void prof_end(int kernel_id, hipStream_t s);
struct ProfScope {
    int id; hipStream_t s;
    ProfScope(int id_, hipStream_t s_) : id(id_), s(s_) { prof_begin(id, s); }
    ~ProfScope() { prof_end(id, s); }
};
int check_hip(hipError_t e, const char* what);

// launchers (one per translation unit)
int launch_preprocess(const AgRasterForwardArgs& a, hipStream_t s);
int launch_tile_scan(const AgRasterForwardArgs& a, hipStream_t s);
int launch_bin_sort(const AgRasterForwardArgs& a, int R, hipStream_t s);
int launch_blend_forward(const AgRasterForwardArgs& a, int R, hipStream_t s);
int launch_blend_backward(const AgRasterBackwardArgs& a, hipStream_t s);
int launch_preprocess_backward(const AgRasterBackwardArgs& a, hipStream_t s);
int launch_debug_wave_reduce16(const float* in, float* out, hipStream_t s);
int launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t s);

}  // namespace ag
