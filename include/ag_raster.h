/*
 * ag_raster.h — C ABI of the MI355X-native depth+alpha Gaussian rasterizer (libag_hip.so).
 *
 * Drop-in boundary for the reference's native rasterizer extension
 *   gaussians/diff_gaussian_rasterization_depth_alpha/ext.cpp:15-19 (pybind module `_C`)
 *   .../rasterize_points.h:17-69 / rasterize_points.cu:35-229  (RasterizeGaussiansCUDA,
 *                                                               RasterizeGaussiansBackwardCUDA, markVisible)
 *   .../cuda_rasterizer/rasterizer.h:20-90 (CudaRasterizer::Rasterizer::forward/backward/markVisible)
 *
 * Plain C: device pointers, sizes, scalars; no torch / pybind types.  All pointers are DEVICE pointers unless
 * a name ends in `_host`.  Every float array is fp32, contiguous, laid out exactly as the reference's tensors
 * (means3D [P,3], scales [P,3], rotations [P,4] as (r,x,y,z), opacities [P], colors [P,3], cov3D [P,6],
 *  viewmatrix/projmatrix: 16 floats, element (i,j) of the TRANSPOSED matrix at 4*i+j, i.e. what
 *  gaussian_renderer.py:49-51 uploads).  Images are CHW ([3,H,W], [1,H,W]).
 *
 * Scratch memory is owned by the caller (torch uint8 tensors in the Python host) and opaque: its layout is
 * private to the library (ag_raster_describe_scratch exposes it for the parity tests only).  The three buffers
 * play the roles of the reference's geomBuffer / binningBuffer / imgBuffer (rasterize_points.cu:73-80) and, like
 * there, must be handed unchanged to the backward call.
 *
 * `stream` is a hipStream_t passed as void* (NULL = the null stream).  Functions return 0 on success or a
 * negative AG_ERR_* code; ag_last_error() describes the most recent failure on the calling thread.
 */
#ifndef AG_RASTER_H
#define AG_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AG_ABI_VERSION 1

#define AG_OK 0
#define AG_ERR_INVALID_ARGUMENT (-1)
#define AG_ERR_SCRATCH_TOO_SMALL (-2)
#define AG_ERR_HIP (-3)
#define AG_ERR_UNSUPPORTED (-4)
#define AG_ERR_RANGE (-5)          /* a convolution met a non-finite accumulator: include/ag_conv.h ag_conv_status */

#define AG_TILE_X 16 /* cuda_rasterizer/config.h:16-17 — baked into the bit-exact tile/sort indices */
#define AG_TILE_Y 16

typedef struct AgRasterForwardArgs {
    /* ---- problem ---- */
    int32_t P;              /* number of Gaussians (means3D.size(0)) */
    int32_t W, H;           /* image_width, image_height */
    int32_t sh_degree;      /* active SH degree D (0..3); ignored when colors_precomp != NULL */
    int32_t sh_coeffs;      /* M = shs.size(1); 0 when shs == NULL */
    int32_t prefiltered;    /* reference flag; a culled Gaussian with prefiltered != 0 is an error there, ignored here */
    float tan_fovx, tan_fovy;
    float scale_modifier;
    /* ---- inputs ---- */
    const float* bg;             /* [3] */
    const float* means3D;        /* [P,3] */
    const float* colors_precomp; /* [P,3] or NULL */
    const float* shs;            /* [P,M,3] or NULL */
    const float* opacities;      /* [P] */
    const float* scales;         /* [P,3] or NULL (then cov3D_precomp) */
    const float* rotations;      /* [P,4] or NULL */
    const float* cov3D_precomp;  /* [P,6] or NULL */
    const float* viewmatrix;     /* [16] */
    const float* projmatrix;     /* [16] */
    const float* campos;         /* [3] */
    /* ---- outputs (caller-allocated; need NOT be zero-filled) ---- */
    float* out_color;            /* [3,H,W] */
    float* out_depth;            /* [1,H,W] */
    float* out_alpha;            /* [1,H,W] */
    int32_t* radii;              /* [P] */
    /* ---- scratch ---- */
    void* geom_buffer;    size_t geom_bytes;     /* >= ag_raster_geom_bytes(P) */
    void* image_buffer;   size_t image_bytes;    /* >= ag_raster_image_bytes(W,H) */
    void* binning_buffer; size_t binning_bytes;  /* >= ag_raster_binning_bytes(num_rendered); only for _render */
} AgRasterForwardArgs;

typedef struct AgRasterBackwardArgs {
    int32_t P, W, H;
    int32_t sh_degree, sh_coeffs;
    int32_t num_rendered;        /* R returned by the forward (the capacity, after ag_raster_forward_optimistic) */
    float tan_fovx, tan_fovy;
    float scale_modifier;
    const float* bg;
    const float* means3D;
    const int32_t* radii;        /* forward output */
    const float* colors_precomp;
    const float* shs;
    const float* scales;
    const float* rotations;
    const float* cov3D_precomp;
    const float* viewmatrix;
    const float* projmatrix;
    const float* campos;
    const float* alphas;         /* forward out_alpha [1,H,W] */
    const float* dL_dout_color;  /* [3,H,W] */
    const float* dL_dout_depth;  /* [1,H,W] */
    const float* dL_dout_alpha;  /* [1,H,W] */
    const void* geom_buffer;
    const void* image_buffer;
    const void* binning_buffer;
    /* ---- outputs (caller-allocated, need NOT be zero-filled; every element is written) ---- */
    float* dL_dmeans2D;   /* [P,3] (x,y in NDC-scaled units, z = 0)   rasterize_points.cu:159 */
    float* dL_dcolors;    /* [P,3] */
    float* dL_dopacity;   /* [P,1] */
    float* dL_dmeans3D;   /* [P,3] */
    float* dL_dcov3D;     /* [P,6] */
    float* dL_dsh;        /* [P,M,3] or NULL when M == 0 */
    float* dL_dscales;    /* [P,3] */
    float* dL_drotations; /* [P,4] */
    /* ---- scratch: per-Gaussian accumulators of the blend backward (dL_dconic, dL_ddepths, ...) ---- */
    void* accum_buffer; size_t accum_bytes;      /* >= ag_raster_accum_bytes(P) */
    /* 1: ADD this view's gradients into the eight output arrays instead of overwriting them -- the sum over the camera views of a
     * multi-view step (the quantity view-sharded training exchanges) without V temporaries and V-1 add kernels.
     * colours-precomp path only (dL_dsh is not accumulated).  0: every element is written (the reference's semantics). */
    int32_t accumulate;
    int32_t reserved;
} AgRasterBackwardArgs;

/* Byte offsets of the private scratch sub-arrays, for the parity tests (tests/ only). */
typedef struct AgRasterScratchLayout {
    /* geom_buffer */
    size_t geom_rec_off;      size_t geom_rec_stride;   /* per-Gaussian record: x,y,conic a,b,c,opacity,r,g,b,depth,r2cut,qcut (floats) */
    size_t geom_cov3d_off;                               /* [P,6] float */
    size_t geom_tiles_touched_off;                       /* [P] u32 */
    /* image_buffer */
    size_t img_ranges_off;                               /* [T,2] u32 */
    size_t img_n_contrib_off;                            /* [H*W] u32 */
    size_t img_tile_count_off;                           /* [T] u32 */
    size_t img_num_rendered_off;                         /* u32 */
    /* binning_buffer */
    size_t bin_point_list_off;                           /* [R] u32, sorted */
    size_t bin_keys_off;                                 /* [R] u64 (depth_bits<<32 | gaussian index), sorted per tile */
} AgRasterScratchLayout;

int ag_abi_version(void);
const char* ag_last_error(void);

size_t ag_raster_geom_bytes(int32_t P);
size_t ag_raster_image_bytes(int32_t W, int32_t H);
size_t ag_raster_binning_bytes(int32_t num_rendered);
size_t ag_raster_accum_bytes(int32_t P);
int ag_raster_describe_scratch(int32_t P, int32_t W, int32_t H, int32_t num_rendered, AgRasterScratchLayout* out);

/*
 * Forward, stage 1 (replaces forward.cu preprocess + the InclusiveSum + the blocking D2H read of
 * num_rendered, rasterizer_impl.cu:249-282): per-Gaussian projection/covariance/tile rect, per-tile counts and
 * their scan.  Blocks until the number of (Gaussian, tile) instances is known and stores it in
 * *num_rendered_host.  P == 0 is legal (returns 0 instances and touches nothing).
 */
int ag_raster_forward_plan(const AgRasterForwardArgs* args, void* stream, int32_t* num_rendered_host);

/*
 * Forward, stage 2 (replaces duplicateWithKeys + SortPairs + identifyTileRanges + renderCUDA,
 * rasterizer_impl.cu:284-337): bins instances per tile, depth-sorts every tile list (ties by Gaussian index,
 * identical to the reference's stable (tile|depth) sort), blends.  Asynchronous on `stream`.
 */
int ag_raster_forward_render(const AgRasterForwardArgs* args, int32_t num_rendered, void* stream);

/*
 * Forward in one call without a GPU-side bubble: like _plan + _render, but stage 2 is enqueued before the host has read the
 * instance count, against a binning buffer the caller sized for `capacity` instances (ag_raster_binning_bytes(capacity);
 * typically 1.25 x the count of the previous frame of the same scene).  The call still blocks until the count is known
 * (the reference's semantics: rasterize_gaussians returns it) -- but only the host waits, the stream keeps running.
 * Returns AG_ERR_SCRATCH_TOO_SMALL when the frame has more instances than `capacity`: *num_rendered_host then holds the true
 * count, the outputs are undefined and nothing was written to the binning buffer; redo the frame with ag_raster_forward_plan
 * + ag_raster_forward_render.  A successful frame's scratch is laid out for `capacity`: pass THAT as
 * AgRasterBackwardArgs.num_rendered.
 */
int ag_raster_forward_optimistic(const AgRasterForwardArgs* args, int32_t capacity, void* stream, int32_t* num_rendered_host);

/* Backward (replaces CudaRasterizer::Rasterizer::backward, rasterizer_impl.cu:341-446).  Asynchronous. */
int ag_raster_backward(const AgRasterBackwardArgs* args, void* stream);

/*
 * One view, forward AND backward, in one call: ag_raster_forward_optimistic followed by ag_raster_backward with nothing but kernel
 * launches in between -- for callers that already hold the upstream image gradients when they render (a multi-view trainer's inner
 * loop, the throughput benchmark): no autograd bookkeeping and no host round trip separates the two halves.  `bwd` is completed
 * from `fwd` by the library (inputs, radii, out_alpha as `alphas`, the three scratch buffers, num_rendered = capacity); the caller
 * fills its upstream gradients, gradient outputs, accum_buffer and `accumulate`.  Blocks the HOST until the instance count of this
 * view is known (everything is enqueued before that).  AG_ERR_SCRATCH_TOO_SMALL as for ag_raster_forward_optimistic: the outputs
 * are undefined and, if `accumulate`, the gradient sums are unchanged (the backward of an overflowing frame finds zero active tiles
 * and adds exact zeros) -- redo the view with a larger capacity.
 */
int ag_raster_forward_backward(const AgRasterForwardArgs* fwd, AgRasterBackwardArgs* bwd, int32_t capacity, void* stream,
                               int32_t* num_rendered_host);

/*
 * The same in two halves, so that the host does not block once per view (round 3): _enqueue issues every kernel of the view and returns a
 * ticket; ag_raster_collect(ticket) waits for the view's instance count (its preprocess + scan, not its blend kernels), returns it and
 * reports AG_ERR_SCRATCH_TOO_SMALL exactly as above (the view then did not touch its outputs / sums and has to be redone).  Collect every
 * ticket (at the latest before the scratch buffers of the view are reused); up to 64 views may be pending.  P == 0: no ticket (-1).
 */
int ag_raster_forward_backward_enqueue(const AgRasterForwardArgs* fwd, AgRasterBackwardArgs* bwd, int32_t capacity, void* stream,
                                       int32_t* ticket);
int ag_raster_collect(int32_t ticket, int32_t* num_rendered_host);

/* mark_visible (rasterize_points.cu:210-229, rasterizer_impl.cu:54-66,141-152): present[i] = view-space z > 0.2 */
int ag_raster_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                           uint8_t* present, void* stream);

/*
 * Kernel timing hooks (bench.py).  ag_prof_enable(mask) makes every subsequent launch of a kernel whose bit is set
 * in `mask` be bracketed by a pair of HIP events recorded on the launch stream; ag_prof_collect synchronises those
 * events and returns, per kernel id, the number of bracketed launches, their summed duration in milliseconds and
 * (work != NULL) the summed work the launches declared -- FLOPs (2 per multiply-add of the un-padded problem) for the two
 * convolution kernels of ag_conv.h, 0 for the others -- then clears the log.  mask == 0 (the default) disables recording
 * entirely.  Every record keeps its own event pair, stream and device, so concurrent streams / threads / devices can be timed.
 */
enum AgKernelId {
    AG_K_PREPROCESS = 0,
    AG_K_TILE_SCAN = 1,
    AG_K_SCATTER = 2,
    AG_K_TILE_SORT = 3,
    AG_K_BLEND_FORWARD = 4,
    AG_K_BLEND_BACKWARD = 5,
    AG_K_PREPROCESS_BACKWARD = 6,
    AG_K_GATHER_CONV = 7,       /* ag_conv.h: forward / input-gradient implicit GEMM (+ its split-K finish) */
    AG_K_WGRAD = 8,             /* ag_conv.h: weight-gradient implicit GEMM */
    AG_K_COUNT = 9
};
/*
 * The large-class launch of the tile sort (tiles of >= 2048 instances: 256 workgroups that each take a whole CU's LDS).  The paths that enqueue a
 * whole frame before the host knows its counts (ag_raster_forward_optimistic, ag_raster_forward_backward_enqueue) skip that launch until some
 * frame of the process has had such a tile: avatar views never have one, and the empty launch cost 4 us per view and a 27-us slot of the
 * overlapped pipeline.  A frame that needs the launch although it was skipped is REFUSED like an overflow (AG_ERR_SCRATCH_TOO_SMALL, outputs and
 * sums untouched): the caller's redo -- the same code path it has for an outgrown capacity -- then gets the launch, and so does every later frame.
 * mode -1: query; 0: forget (skip again until seen); 1: always launch.  Returns the state after the call (0 / 1) or an error code.
 */
int ag_raster_large_tile_sort(int32_t mode);

const char* ag_prof_kernel_name(int32_t kernel_id);
int ag_prof_enable(uint32_t kernel_mask);
int ag_prof_collect(int32_t* launches /*[AG_K_COUNT]*/, float* total_ms /*[AG_K_COUNT]*/, double* work /*[AG_K_COUNT] or NULL*/);
/* The same, and one CSV line per bracketed launch in launch order into `path`: kernel,tag,work,ms -- the tag is what the launcher declared
 * (the convolutions put their shape, tile and split-K count there).  Diagnostic: profiles/conv_launch_table.py. */
int ag_prof_collect_to(const char* path, int32_t* launches, float* total_ms, double* work);

/*
 * Calibration hook (profiles/atomic_rate.py): `blocks` workgroups of 8 waves; every wave issues `iters` instructions, each adding
 * 1.0f to the first `comps` (<= 16) floats of 4 pseudo-random 64-byte lines of accum [lines][16] -- the access shape of the blend
 * backward's flush.  Measures the line-atomic rate of the memory side.
 */
int ag_debug_atomic_rate(float* accum, int32_t lines, int32_t blocks, int32_t iters, int32_t comps, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AG_RASTER_H */
