// MFMA convolutions for the StyleUNet, batch 1, fp32 (gfx950).  See include/ag_conv.h.
//
// Replaces the cuDNN calls behind network/styleunet/conv2d_gradfix.py (conv2d / conv_transpose2d) and their
// backward.  Every variant is one of two implicit GEMMs, in one of two engines (ag_conv_set_math):
//   * the split engine: fp32 operands as sums of 16-bit parts, fp32 accumulation.  Default (round 4): two fp16 parts under a per-tensor
//     power-of-two scale, three products per fp32 product on v_mfma_f32_32x32x16_f16 (section "fp32 products on the fp16 matrix pipe");
//     before: three bf16 parts, six products on v_mfma_f32_32x32x16_bf16 -- products within 2^-23 of exact either way; an opt-in
//     three-product bf16 form (2^-16);
//   * the fp32 engine: v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate; 157 TF peak = the fp32 vector rate, but it
//     leaves the VALU free for the gather arithmetic).
// Both share the problem decomposition, the tiles, the K order, split-K and the epilogues:
//
//   gather-conv  Y[m][gy, gx] = sum_{t, c} A[m][(t, c)] * Xin[c][gy*sy + dy_t][gx*sx + dx_t]
//       forward conv (any stride/pad), input gradient of a stride-1 conv (taps mirrored), input gradient of a transposed
//       conv (= stride-2 conv) and -- split into the 4 output-parity classes so no zero taps are multiplied -- the
//       stride-2 transposed conv forward and the input gradient of a stride-2 conv.
//   wgrad        dW[m][(c, t)] = sum_{gy, gx} A[m][gy, gx] * Xin[c][gy*sy + dy_t][gx*sx + dx_t]      (split-K + atomics)
//
// Tile engine (shared): BM x BN outputs per workgroup of 8 waves (2 x 4), each wave WMB x WNB blocks of 32 x 32
// (128 x 128 = 2x1 blocks per wave, and 64 x 256 = 1x2 for the <= 64-row problems), BK = 16.  Eight waves rather than
// four: the loader / address work per wave halves while the MFMA work per SIMD stays, and two waves of the same
// workgroup share every SIMD, so the matrix pipe has a second issuer between barriers (PMC: with 4-wave workgroups the
// pipe sat at 62 % with both co-resident workgroups marching in phase).  Both operands sit in LDS **k-contiguous** ([row][16 k + 4 pad]);
// since the order of the k's inside a tile is free as long as A and B agree, MFMA step i takes k = 8*(lane>>5) + i,
// so each lane's 8 operand values of a 32-row block are two ds_read_b128 (8 per wave and tile instead of 32 scalar
// reads).  Global->register loads of tile k+1 are issued before the MFMAs of tile k and written to the other LDS buffer
// after them (one LDS-only barrier per K step).
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "ag_common.h"
#include "ag_groups.h"
#include "../../include/ag_conv.h"
#include "../../include/ag_raster.h"   // AgKernelId (timing hooks)

namespace ag {

// ag_conv_pointwise.hip: 1 x 1 convolutions with <= 32 output rows or <= 4 input channels as streaming VALU kernels.
// Return 1 when they handled the call, 0 when it is not theirs (the MFMA path below runs), < 0 on error.
// Grouped (ag_groups.h): G instances, activations at the given float strides, weights / biases from pointer tables, dw stacked.
int pointwise_forward(const AgConvDesc* d, int G, const float* x, long long x_gs, const PtrTable& w, const float* out_scale, const PtrTable& bias,
                      float* y, long long y_gs, hipStream_t s);
int pointwise_backward_input(const AgConvDesc* d, int G, const float* dy, long long dy_gs, const PtrTable& w, float* dx, long long dx_gs, hipStream_t s);
int pointwise_backward_weight(const AgConvDesc* d, int G, const float* x, long long x_gs, const float* dy, long long dy_gs, float* dw, long long dw_gs,
                              void* workspace, size_t workspace_bytes, hipStream_t s);

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: HIP's float4 struct in arrays defeats SROA (scratch spills)

#ifndef AG_CONV_WAVES_PER_SIMD
#define AG_CONV_WAVES_PER_SIMD 4     // __launch_bounds__ 2nd argument (waves per SIMD): 4 = two 8-wave workgroups per CU (<= 128 VGPRs)
#endif

constexpr int BK = 16;
constexpr int LDK = BK + 4;     // row pitch in floats: 80 B keeps b128 accesses aligned and spreads 16 rows over all 64 banks
constexpr int kMaxTaps = 16;

// WVM x WVN waves per workgroup, each owning WMB x WNB blocks of 32 x 32
template <int WMB, int WNB, int WVM, int WVN>
struct Tile {
    static constexpr int BM = 32 * WMB * WVM, BN = 32 * WNB * WVN, NT = 64 * WVM * WVN;
    static constexpr int lds_floats = 2 * (BM + BN) * LDK;
};

// Operand registers of one K tile for a wave: 8 k-values (two b128) per 32-row block
template <int WMB, int WNB>
struct Operands {
    f32x4 a[WMB][2], b[WNB][2];
};

template <int WMB, int WNB>
__device__ __forceinline__ void read_operands(const float* __restrict__ As, const float* __restrict__ Bs, int wm, int wn, int lane,
                                              Operands<WMB, WNB>& o)
{
    const int r = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int i = 0; i < WMB; i++)
#pragma unroll
        for (int h = 0; h < 2; h++)
            o.a[i][h] = *reinterpret_cast<const f32x4*>(As + ((wm * WMB + i) * 32 + r) * LDK + kh * 8 + 4 * h);
#pragma unroll
    for (int j = 0; j < WNB; j++)
#pragma unroll
        for (int h = 0; h < 2; h++)
            o.b[j][h] = *reinterpret_cast<const f32x4*>(Bs + ((wn * WNB + j) * 32 + r) * LDK + kh * 8 + 4 * h);
}

// half a K tile (4 MFMA k-steps) on every block of the wave
template <int WMB, int WNB>
__device__ __forceinline__ void mma_half(const Operands<WMB, WNB>& o, int h, f32x16 (&acc)[WMB][WNB])
{
#pragma unroll
    for (int e = 0; e < 4; e++)
#pragma unroll
        for (int i = 0; i < WMB; i++)
#pragma unroll
            for (int j = 0; j < WNB; j++)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a[i][h][e], o.b[j][h][e], acc[i][j], 0, 0, 0);
}

// 32 x 32 x 16 update of a wave's WMB x WNB blocks from the k-contiguous LDS tiles (un-pipelined form, used by wgrad)
template <int WMB, int WNB>
__device__ __forceinline__ void mma_tile(const float* __restrict__ As, const float* __restrict__ Bs, int wm, int wn, int lane,
                                         f32x16 (&acc)[WMB][WNB])
{
    Operands<WMB, WNB> o;
    read_operands<WMB, WNB>(As, Bs, wm, wn, lane, o);
    mma_half<WMB, WNB>(o, 0, acc);
    mma_half<WMB, WNB>(o, 1, acc);
}

// ------------------------------------------------------------------------------------------------------------------
// gather-conv
// ------------------------------------------------------------------------------------------------------------------
constexpr int kMaxClasses = 4;

// One output class: the whole output of a plain gather, or one of the 4 output-parity classes of a stride-2 scatter
// (transposed conv forward, input gradient of a stride-2 conv).  All classes of a problem run in ONE launch: blockIdx.x walks
// the classes' N tiles back to back (heaviest class first), so a layer whose classes have 256 tiles each still fills the chip.
struct GatherClass {
    int gh, gw, y0, x0;    // class grid, and its origin in the output (output pixel = (y0 + gy * os, x0 + gx * os))
    int ntaps, nkt;        // K tiles of the class = ntaps * Cpad / 16 (K tile = 16 channels of ONE tap)
    int tile_begin;        // first blockIdx.x of the class
    int col_begin;         // first column of the class in the split-K partial image [z][Mpad][Ncols]
    long long at_off;      // float offset of the class's packed weights inside At
    int zero_weights;      // degenerate class (no tap reaches it): its weights are packed as zeros
    int pad_;
    int dy[kMaxTaps], dx[kMaxTaps];
};

struct GatherProblem {
    const float* xin;      // [Cg][Hg][Wg]
    const float* At;       // per class: tile-blocked packed weights [Mpad / BM][nkt][BM][16]
    float* yout;           // [M][OHf][OWf]
    const float* out_scale;  // [M] or null (G = 1 only)
    int Cg, Cpad, Hg, Wg;  // Cpad = channels rounded up to BK: every K tile lies inside one tap
    int M, Mpad, OHf, OWf;
    int os;                // output stride of the class grids (1: plain gather, 2: parity classes)
    int sy, sx;            // input stride of the gather
    int nclasses, Ncols;   // Ncols = sum over classes of gh * gw
    int kt_per_split;      // K tiles per blockIdx.z slice (gridDim.z == 1: all of them)
    float* partial;        // gridDim.z > 1: raw accumulators go to partial[z][Mpad][Ncols] and reduce_splits_kernel finishes
    // grouped launch (ag_groups.h): gridDim.y = G * mtiles, instance = blockIdx.y / mtiles
    int G, mtiles;
    long long x_gs, y_gs;  // floats between the instances' inputs (0: one shared input) / outputs
    long long at_gs;       // packed-weight elements (the unit of GatherClass::at_off) between the instances' packed weights
    long long part_gs;     // floats between the instances' split-K partial images (= splits * Mpad * Ncols)
    PtrTable bias_t;       // per-instance bias (replaces `bias`, which is kept for the epilogue's signature and set per workgroup)
    // activation inside the epilogue / the split-K finish (round 4; plain gathers only): y = lrelu((acc + nw * noise[pix] | + addend[m][pix])
    // + bias[m], slope) * act_scale -- the expression of noise_bias_act_forward_kernel, so the pre-activation tensor never goes to memory
    int act;               // 0: none, 1: noise form, 2: addend form
    float slope, act_scale;
    PtrTable noise_t;      // act 1: per-instance noise [OHf * OWf] (null: no noise); act 2: per-instance addend [M][OHf * OWf]
    PtrTable nw_t;         // act 1: per-instance noise weight [1]
    int xcd_order;         // 1: workgroups take their (N tile, instance * M tile) in the XCD-aware order of tile_of_workgroup() (host: gridDim.x * gridDim.y % 8 == 0)
    float* out_amax;       // act != 0: zeroed [G][kAmaxParts] slots for the largest magnitude of the activated output (ConvAct::out_amax), or null
    int* status;           // host-visible sticky flag raised on a non-finite accumulator (ag_conv_status), or null
    int status_tag;        // what the offending launch stores there: kind << 28 | rows << 14 | channels (status_tag_of)
    // fp16 split form: partial maxima (absmax_kernel) of the packed weights' source tensors [G][kAmaxParts] and of the gathered tensor
    // ([G][kAmaxParts], or one row when the instances share their input); amax_a_mult = |weight_scale| (the packed values are w * scale)
    const float* amax_a;
    const float* amax_b;
    float amax_a_mult;
    GatherClass cls[kMaxClasses];
};

// what a workgroup of a grouped launch works on
struct GroupView {
    int grp, my;                 // instance, M tile inside the instance
    const float* xin;
    float* yout;
    float* partial;
    const float* bias;
    const float* noise;          // activation operands of the instance (GatherProblem::act)
    float nw;
};
// XCD-aware order (round 4).  The hardware deals workgroups to the 8 XCDs round-robin by linear id, so neighbouring N tiles -- neighbouring
// image rows, whose 3 x 3 gathers read the same input rows -- and the M tiles of one N tile (which gather the SAME input) landed on eight
// different L2s: PMC on a 256 -> 256 layer at 256^2 showed 523 MB fetched by L2 per launch for 69.5 MB of unique input
// (profiles/r04_pmc_traffic_gather_conv_f16_256_256_256.txt).  Here XCD j takes the j-th eighth of the (N tile, instance x M tile) pairs,
// M tiles and instances of one N tile back to back, N tiles in image order.
struct TileId { int bx, by; };
__device__ __forceinline__ TileId tile_of_workgroup(const GatherProblem& p)
{
    TileId t{ (int)blockIdx.x, (int)blockIdx.y };
    if (p.xcd_order) {
        const int L = (int)blockIdx.x + (int)gridDim.x * (int)blockIdx.y;
        const int Lg = (L & 7) * (((int)gridDim.x * (int)gridDim.y) >> 3) + (L >> 3);
        t.bx = Lg / (int)gridDim.y;
        t.by = Lg - t.bx * (int)gridDim.y;
    }
    return t;
}
__device__ __forceinline__ GroupView group_view(const GatherProblem& p, int by)
{
    GroupView v;
    v.grp = by / p.mtiles;
    v.my = by - v.grp * p.mtiles;
    v.xin = p.xin + (size_t)v.grp * p.x_gs;
    v.yout = p.yout + (size_t)v.grp * p.y_gs;
    v.partial = p.partial ? p.partial + (size_t)v.grp * p.part_gs : nullptr;
    v.bias = p.bias_t.p[v.grp];
    v.noise = p.act ? p.noise_t.p[v.grp] : nullptr;
    v.nw = (p.act == 1 && v.noise) ? p.nw_t.p[v.grp][0] : 1.0f;
    return v;
}

// Largest magnitude a wave stored -> one of the kAmaxParts zeroed slots of its instance: an unsigned atomic maximum of the float bits
// (order-independent, hence deterministic), skipped when the slot already holds a larger value (a stale read only costs an atomic).
// Every lane of the wave must arrive here.
__device__ __forceinline__ void emit_amax(float* slots, float mx, int salt)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) {
        unsigned int* slot = reinterpret_cast<unsigned int*>(slots) + ((unsigned)salt & (kAmaxParts - 1));
        const unsigned int bits = __float_as_uint(mx);
        if (bits > __builtin_nontemporal_load(slot)) atomicMax(slot, bits);
    }
}

// Range guard (ag_conv_status): a lane that holds a non-finite accumulator raises the host-visible flag.  Runs once per tile; the store is
// executed by offending lanes only, so a healthy launch never touches the word.
template <int WMB, int WNB>
__device__ __forceinline__ void flag_non_finite(int* status, int tag, const f32x16 (&acc)[WMB][WNB])
{
    if (!status) return;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < WMB; i++)
#pragma unroll
        for (int j = 0; j < WNB; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) bad |= !(fabsf(acc[i][j][r]) <= 3.4028234664e38f);      // inf or NaN
    if (bad) __hip_atomic_store(status, tag ? tag : 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Epilogue shared by the gather kernels.  C/D layout of the 32 x 32 MFMAs: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
template <int WMB, int WNB>
__device__ __forceinline__ void gather_epilogue(const GatherProblem& p, const GroupView& gv, const GatherClass& cl, f32x16 (&acc)[WMB][WNB], int m0,
                                                int n0, int N, int wm, int wn, int lane)
{
    const int gw = cl.gw;
    const int col = lane & 31, rbase = 4 * (lane >> 5);
    flag_non_finite<WMB, WNB>(p.status, p.status_tag, acc);
    if (gridDim.z > 1) {
        float* part = gv.partial + (size_t)blockIdx.z * p.Mpad * p.Ncols + cl.col_begin;
#pragma unroll
        for (int i = 0; i < WMB; i++)
#pragma unroll
            for (int j = 0; j < WNB; j++) {
                const int nn = n0 + (wn * WNB + j) * 32 + col;
                if (nn >= N) continue;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int m = m0 + (wm * WMB + i) * 32 + (r & 3) + 8 * (r >> 2) + rbase;
                    part[(size_t)m * p.Ncols + nn] = acc[i][j][r];
                }
            }
        return;
    }
    int nn[WNB];
    size_t opix[WNB];
#pragma unroll
    for (int j = 0; j < WNB; j++) {
        nn[j] = n0 + (wn * WNB + j) * 32 + col;
        const int q = min(nn[j], N - 1);
        const int oy = q / gw, ox = q - oy * gw;
        opix[j] = (size_t)(cl.y0 + oy * p.os) * p.OWf + (cl.x0 + ox * p.os);
    }
    const bool has_scale = p.out_scale != nullptr, has_bias = gv.bias != nullptr;
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < WMB; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int m = m0 + (wm * WMB + i) * 32 + (r & 3) + 8 * (r >> 2) + rbase;
            if (m >= p.M) continue;
            const float sc = has_scale ? p.out_scale[m] : 1.f, bi = has_bias ? gv.bias[m] : 0.f;
            float* row = gv.yout + (size_t)m * p.OHf * p.OWf;
            if (p.act) {
                // noise: one value per pixel; addend: one per (row, pixel)
                const float* nz = gv.noise ? (p.act == 2 ? gv.noise + (size_t)m * p.OHf * p.OWf : gv.noise) : nullptr;
#pragma unroll
                for (int j = 0; j < WNB; j++)
                    if (nn[j] < N) {
                        const float t = fmaf(gv.nw, nz ? nz[opix[j]] : 0.f, acc[i][j][r]) + bi;
                        const float y = (t > 0.f ? t : t * p.slope) * p.act_scale;
                        row[opix[j]] = y;
                        mx = fmaxf(mx, fabsf(y));
                    }
                continue;
            }
#pragma unroll
            for (int j = 0; j < WNB; j++)
                if (nn[j] < N) row[opix[j]] = acc[i][j][r] * sc + bi;
        }
    if (p.act && p.out_amax) emit_amax(p.out_amax + (size_t)gv.grp * kAmaxParts, mx, (int)blockIdx.x * 8 + (int)(threadIdx.x >> 6));
}

// K is ordered (channel block of 16, tap, channel in block), so the 16 rows of a K tile are 16 consecutive channels of ONE tap:
// the tap (and with it the input offset and the padding test) is a wave-uniform scalar per tile, each thread keeps the
// 16-bit in-bounds mask of its output pixel over the taps (consecutive tiles walk the taps of one channel block, so the
// shifted re-reads of the same input lines stay in L1/L2), and the gathers of a tile are unconditional loads
// `global_load_dword v, v_off, s[base]` (uniform 64-bit channel base, 32-bit per-thread byte offset) -- all in flight
// under the MFMAs; spatial padding is applied when the tile is written to LDS.  Nothing in the K loop touches memory for
// bookkeeping: the per-tap input offsets sit in one VGPR (lane t = tap t) and are fetched with v_readlane; channels past Cg (only
// the 3- and 12-channel inputs have them) are clamped to the last real channel -- their packed weights are zero.
template <int WMB, int WNB, int WVM, int WVN>
__global__ void __launch_bounds__(64 * WVM * WVN, AG_CONV_WAVES_PER_SIMD) gather_conv_kernel(GatherProblem p)
{
    using T = Tile<WMB, WNB, WVM, WVN>;
    constexpr int BM = T::BM, BN = T::BN, NT = T::NT;
    constexpr int G = NT / BN, KG = BK / G;           // B loader: G thread groups, each KG consecutive k's of a pixel
    constexpr int AF = BM * 4;                        // A loader: float4's in a tile (one per thread for the first AF threads)
    static_assert(AF <= NT && KG % 4 == 0 && BN >= 64, "loader shapes");
    __shared__ __attribute__((aligned(16))) float smem[T::lds_floats];
    float* const As0 = smem;
    float* const Bs0 = smem + 2 * BM * LDK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WVN, wn = wave % WVN;

    // class of this workgroup (wave-uniform)
    const TileId tw = tile_of_workgroup(p);
    int ci = 0;
#pragma unroll
    for (int c = 1; c < kMaxClasses; c++)
        if (c < p.nclasses && tw.bx >= p.cls[c].tile_begin) ci = c;
    const GatherClass& cl = p.cls[ci];
    const int gw = cl.gw, ntaps = cl.ntaps;
    const int N = cl.gh * gw;
    const GroupView gv = group_view(p, tw.by);
    const int m0 = gv.my * BM, n0 = (tw.bx - cl.tile_begin) * BN;

    const int n_loc = tid % BN, g = (wave * 64) / BN;   // g is wave-uniform (BN >= 64)
    const int n = n0 + n_loc;
    const bool n_ok = n < N;
    const int gy = n_ok ? n / gw : 0, gx = n_ok ? n - (n / gw) * gw : 0;
    const int iy0 = gy * p.sy, ix0 = gx * p.sx;
    uint32_t vmask = 0;
    for (int t = 0; t < ntaps; t++) {
        const int iy = iy0 + cl.dy[t], ix = ix0 + cl.dx[t];
        if (n_ok && iy >= 0 && iy < p.Hg && ix >= 0 && ix < p.Wg) vmask |= 1u << t;
    }
    const int pix = iy0 * p.Wg + ix0;
    const int tl = lane & (kMaxTaps - 1);
    const int toff_vec = cl.dy[tl] * p.Wg + cl.dx[tl];          // lane t: input offset of tap t (lanes >= ntaps: unused)

    f32x16 acc[WMB][WNB];
#pragma unroll
    for (int i = 0; i < WMB; i++)
#pragma unroll
        for (int j = 0; j < WNB; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int nkt_all = cl.nkt;
    const int kt_beg = min(nkt_all, (int)blockIdx.z * p.kt_per_split), kt_end = min(nkt_all, kt_beg + p.kt_per_split);
    const int nkt = kt_end - kt_beg;                       // may be 0 for the short classes of a split launch: zeros are written
    // K tile kt = (channel block kt / ntaps, tap kt % ntaps); running (tap, first channel, pointers) of the next tile to load
    int t_cur = kt_beg % ntaps;
    int c_cur = (kt_beg / ntaps) * BK + g * KG;
    const size_t plane_b = (size_t)p.Hg * p.Wg * sizeof(float);
    const char* const xin_b = reinterpret_cast<const char*>(gv.xin);
    const int c_last = p.Cg - 1;
    const char* a_ptr = reinterpret_cast<const char*>(p.At + (size_t)gv.grp * p.at_gs + cl.at_off + ((size_t)gv.my * nkt_all + kt_beg) * (BM * BK));
    const uint32_t a_voff = (uint32_t)min(tid, AF - 1) * 16u;       // threads past the tile re-read its last float4

    // Three-stage software pipeline.  While the MFMAs of tile k run from operand registers, tile k+1 moves
    // registers -> LDS -> operand registers (one barrier, placed in the middle of the MFMA phase so that the LDS write
    // burst and the operand read burst of all waves are covered by matrix work), and the global loads of tile k+2 are in
    // flight.  Two sets of staging registers (S) and of operand registers (O) alternate by tile parity.
    struct Stage {
        f32x4 ra;
        float rb[KG];
        bool tap_ok;        // rb[] are real samples for this thread's pixel (else spatial padding -> 0 at the LDS write)
    };
    Stage S[2];
    Operands<WMB, WNB> O[2];
    const bool a_thread = tid < AF;
    auto gload = [&](Stage& st) {
        st.ra = *reinterpret_cast<const f32x4*>(a_ptr + a_voff);
        a_ptr += BM * BK * sizeof(float);
        st.tap_ok = (vmask >> t_cur) & 1u;
        const int toff = __builtin_amdgcn_readlane(toff_vec, t_cur);
        const uint32_t voff = st.tap_ok ? (uint32_t)(pix + toff) * 4u : 0u;
#pragma unroll
        for (int j = 0; j < KG; j++) {
            const char* src = xin_b + (size_t)min(c_cur + j, c_last) * plane_b;     // uniform
            st.rb[j] = *reinterpret_cast<const float*>(src + voff);
        }
        t_cur++;
        if (t_cur == ntaps) { t_cur = 0; c_cur += BK; }
    };
    auto lstore = [&](int buf, const Stage& st) {
        float* As = As0 + buf * BM * LDK;
        float* Bs = Bs0 + buf * BN * LDK;
        if (a_thread) *reinterpret_cast<f32x4*>(As + (tid >> 2) * LDK + (tid & 3) * 4) = st.ra;   // float4 #tid of the [BM][16] tile
#pragma unroll
        for (int q = 0; q < KG / 4; q++) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = st.tap_ok ? st.rb[4 * q + e] : 0.f;
            *reinterpret_cast<f32x4*>(Bs + n_loc * LDK + g * KG + 4 * q) = v;
        }
    };
    auto lread = [&](int buf, Operands<WMB, WNB>& o) {
        read_operands<WMB, WNB>(As0 + buf * BM * LDK, Bs0 + buf * BN * LDK, wm, wn, lane, o);
    };

    if (nkt > 0) {
        gload(S[0]);
        if (nkt > 1) gload(S[1]);
        lstore(0, S[0]);
        lds_barrier();
        lread(0, O[0]);
        // Steady-state step for tile kt (kt + 2 < nkt): branch-free, so the compiler's vmcnt bookkeeping stays exact and the
        // wait before the LDS write covers only the loads issued one step earlier.
        auto step_full = [&](int kt, Stage& s_same, Stage& s_next, Operands<WMB, WNB>& o_cur, Operands<WMB, WNB>& o_next) {
            gload(s_same);                                           // s_same held tile kt, already written to LDS
            mma_half<WMB, WNB>(o_cur, 0, acc);
            lstore((kt + 1) & 1, s_next);
            lds_barrier();
            lread((kt + 1) & 1, o_next);
            mma_half<WMB, WNB>(o_cur, 1, acc);
        };
        auto step_tail = [&](int kt, Stage& s_next, Operands<WMB, WNB>& o_cur, Operands<WMB, WNB>& o_next, bool has_next) {
            mma_half<WMB, WNB>(o_cur, 0, acc);
            if (has_next) {
                lstore((kt + 1) & 1, s_next);
                lds_barrier();
                lread((kt + 1) & 1, o_next);
            }
            mma_half<WMB, WNB>(o_cur, 1, acc);
        };
        int kt = 0;
        for (; kt + 3 < nkt; kt += 2) {
            step_full(kt, S[0], S[1], O[0], O[1]);
            step_full(kt + 1, S[1], S[0], O[1], O[0]);
        }
        const int rem = nkt - kt;                                    // 1, 2 or 3 tiles left, kt even
        if (rem == 3) {
            step_full(kt, S[0], S[1], O[0], O[1]);
            step_tail(kt + 1, S[0], O[1], O[0], true);
            step_tail(kt + 2, S[1], O[0], O[1], false);
        } else if (rem == 2) {
            step_tail(kt, S[1], O[0], O[1], true);
            step_tail(kt + 1, S[0], O[1], O[0], false);
        } else {
            step_tail(kt, S[1], O[0], O[1], false);
        }
    }
    // (the gather kernel above and the wgrad kernel below share this loop shape)

    gather_epilogue<WMB, WNB>(p, gv, cl, acc, m0, n0, N, wm, wn, lane);
}

// ------------------------------------------------------------------------------------------------------------------
// fp32 products on the bf16 matrix pipe: three-way split, six MFMAs per K tile
// ------------------------------------------------------------------------------------------------------------------
// v_mfma_f32_32x32x2_f32 retires 64 FLOP / clk / SIMD, v_mfma_f32_32x32x16_bf16 1024.  Every fp32 operand is written as the sum of
// three bf16 numbers x = x0 + x1 + x2 (x0 = rn_bf16(x), x1 = rn_bf16(x - x0), x2 = rn_bf16(x - x0 - x1): each subtraction is exact
// in fp32 and |x - x0 - x1 - x2| <= 2^-27 |x|), and a product a * b is formed as the six terms a_i b_j with i + j <= 2, each an
// exact fp32 product inside the matrix core, accumulated in fp32.  The dropped terms (a1 b2, a2 b1, a2 b2, and the split
// remainders) are <= 2^-23 |a| |b| in the worst case and ~2^-25 rms -- the size of the rounding of the fp32 product itself.  One K
// tile of a 32 x 32 block costs 6 x 32 = 192 matrix-pipe cycles instead of 8 x 64 = 512.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kPlanes = 3;
constexpr int kRowB = 2 * BK;        // bytes of one row (16 bf16) of a plane: two 16-byte chunks

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi)       // v_cvt_pk_bf16_f32 (round to nearest even)
{
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){ lo, hi }, bf16x2));
}
// two fp32 values -> three words holding (lo, hi) bf16 pairs of the three planes
__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t& p0, uint32_t& p1, uint32_t& p2)
{
    p0 = pack_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(p0 << 16), r1 = x1 - __uint_as_float(p0 & 0xffff0000u);
    p1 = pack_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(p1 << 16), s1 = r1 - __uint_as_float(p1 & 0xffff0000u);
    p2 = pack_bf16(s0, s1);
}
// ------------------------------------------------------------------------------------------------------------------
// fp32 products on the fp16 matrix pipe: two-way split under a per-tensor power-of-two scale, three MFMAs per K tile (round 4)
// ------------------------------------------------------------------------------------------------------------------
// fp16 carries 11 significant bits against bf16's 8, so TWO parts hold 22 bits where the bf16 split needs three: with y = s x
// (s a power of two that puts the tensor's largest magnitude M into [2^14, 2^15): exact),  h = rn_f16(y),  l = rn_f16(y - h)  (y - h is
// exact in fp32 and <= 2^-12 |y|),  |y - h - l| <= max(2^-24 |y|, 2^-25): the first bound wherever l is a normal fp16 number (|y| >= 2^-2,
// i.e. |x| >= 2^-17 M), the second -- 2^-39 M in x's units -- below that, where l is subnormal.  A product is a_h b_h + a_h b_l + a_l b_h
// (three exact fp32 products inside the matrix core, one fp32 accumulator); the dropped a_l b_l is <= 2^-24 |a| |b|.  Per output:
//     |error| <= 3 * 2^-24 sum |a| |b|  +  2^-39 (M_b sum |a| + M_a sum |b|)
// -- the six-product bf16 form's grade (2^-23) for every output whose operands are on average within 2^16 of their tensors' maxima, and for
// the others an ABSOLUTE error 2^-15 of the rounding fp32 itself commits on the tensor's large outputs.  Half the matrix instructions and
// two thirds of the LDS traffic of the bf16 form.  The price is the scale: fp16 has 5 exponent bits, so every operand tensor needs its
// largest magnitude known before its consumer starts (absmax_kernel below: per-workgroup partial maxima, finished by the consumer; no
// atomics, no host round trip).
// (A variant that kept the low parts normal down to 2^-28 M -- l = rn_f16(2^11 (y - h)), the cross terms in a second accumulator scaled
// by 2^-11 at the end -- measured the same deviations (profiles/r04_split_f16_products.txt) but needs 190 registers: one workgroup per CU,
// 175 us per launch where this form runs 145; the three-product bf16 kernel confined to one workgroup per CU lands on the same 171 us:
// profiles/r04_split_f16_ab.txt.  Commit c06b403 has it.)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr int kF16 = 2;                 // value of the kernels' NTERMS parameter that selects this form (6 / 3: the bf16 forms)
// ONE fp16 part under the same per-tensor scale (round 5, AG_CONV_MATH_F16): h = rn_f16(s x), one product a_h b_h per fp32 product.  11
// significant bits per operand = the operand grade of cuDNN's TF32 path (10 explicit mantissa bits, fp32 accumulation), which is what the
// reference's convolutions run on its own hardware (network/styleunet/conv2d_gradfix.py:185-189 hands torch.backends.cudnn.allow_tf32 --
// True by default -- to cuDNN and nothing in the repository clears it).  Per product <= (2^-11 + 2^-22) |a| |b| wherever both scaled operands
// are normal fp16 numbers (|x| >= 2^-29 M); one plane in LDS and in the packed weights, a third of the matrix instructions.
constexpr int kF16S = 1;
constexpr bool is_f16_form(int nterms) { return nterms == kF16 || nterms == kF16S; }

__device__ __forceinline__ uint32_t pack_f16(float lo, float hi)        // round to nearest even
{
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){ lo, hi }, f16x2));
}
// two fp32 values, already multiplied by the tensor's scale -> two words holding (lo, hi) fp16 pairs of the two planes
__device__ __forceinline__ void split_pair_h(float y0, float y1, uint32_t& p0, uint32_t& p1)
{
    p0 = pack_f16(y0, y1);
    const f16x2 h = __builtin_bit_cast(f16x2, p0);
    p1 = pack_f16(y0 - (float)h[0], y1 - (float)h[1]);
}
// scale of a tensor instance from the partial maxima of absmax_kernel (wave-uniform): s = 2^(14 - e) for a maximum in [2^e, 2^(e+1)),
// 1 / s exactly; an all-zero (or non-finite) tensor gets s = 1
struct TensorScale { float s, inv; };
__device__ __forceinline__ TensorScale tensor_scale(const float* __restrict__ parts, float mult, int lane)
{
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < kAmaxParts / 64; i++) m = fmaxf(m, parts[i * 64 + lane]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    m *= mult;
    int e = (int)((__float_as_uint(m) >> 23) & 0xffu);          // biased exponent
    TensorScale t;
    if (e == 0 || e == 255) { t.s = 1.f; t.inv = 1.f; return t; }
    e = max(e, 16);
    e = __builtin_amdgcn_readfirstlane(e);
    t.s = __uint_as_float((uint32_t)(268 - e) << 23);           // 2^(14 - (e - 127))
    t.inv = __uint_as_float((uint32_t)(e - 14) << 23);
    return t;
}

// Partial maxima of |x| of up to 2 * kMaxGroups tensors in one launch: grid (kAmaxParts, jobs).  A job is `rows` runs of `len` floats
// `stride` floats apart (an activation stack: one run; a channel slice of a wider weight tensor: one run per output channel).
struct AmaxJobs {
    const float* ptr[2 * kMaxGroups];
    long long len[2 * kMaxGroups], stride[2 * kMaxGroups];
    int rows[2 * kMaxGroups];
    float* outp[2 * kMaxGroups];        // job j writes outp[j][kAmaxParts] (a job of length 0 zeroes them: the slots a producer kernel will
                                        // raise with atomic maxima, ConvAct::out_amax)
};
constexpr int kAmaxThreads = 1024;      // 16 waves per workgroup: 256 workgroups per tensor have to keep HBM busy on their own
__global__ void __launch_bounds__(kAmaxThreads) absmax_kernel(AmaxJobs J)
{
    const int job = blockIdx.y;
    const float* __restrict__ x = J.ptr[job];
    const long long len = J.len[job];
    float m = 0.f;
    if (J.rows[job] == 1 && (len & 3) == 0 && (((size_t)x) & 15) == 0) {
        const long long n4 = len >> 2, step = (long long)gridDim.x * kAmaxThreads;
        const f32x4* __restrict__ x4 = reinterpret_cast<const f32x4*>(x);
        long long i = (long long)blockIdx.x * kAmaxThreads + threadIdx.x;
        for (; i + 3 * step < n4; i += 4 * step) {              // four loads in flight per thread: the sweep is latency-bound otherwise
            const f32x4 v0 = x4[i], v1 = x4[i + step], v2 = x4[i + 2 * step], v3 = x4[i + 3 * step];
#pragma unroll
            for (int e = 0; e < 4; e++) m = fmaxf(fmaxf(m, fmaxf(fabsf(v0[e]), fabsf(v1[e]))), fmaxf(fabsf(v2[e]), fabsf(v3[e])));
        }
        for (; i < n4; i += step) {
            const f32x4 v = x4[i];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
    } else {
        const long long n = len * J.rows[job];
        for (long long i = (long long)blockIdx.x * kAmaxThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kAmaxThreads) {
            const long long r = i / len;
            m = fmaxf(m, fabsf(x[r * J.stride[job] + (i - r * len)]));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float wm[kAmaxThreads / 64];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 1; i < kAmaxThreads / 64; i++) m = fmaxf(m, wm[i]);
        J.outp[job][blockIdx.x] = m;
    }
    // a launch of fewer than kAmaxParts workgroups per job (third session: conv_absmax sizes the grid to the largest job): the slots nobody
    // owns are zeroed, dealt round-robin to the workgroups that exist
    if ((int)gridDim.x < kAmaxParts && (int)threadIdx.x < kAmaxParts && threadIdx.x >= gridDim.x && threadIdx.x % gridDim.x == blockIdx.x)
        J.outp[job][threadIdx.x] = 0.f;
}

// The same for a launch whose jobs are all short (the bookkeeping launches of a layer call whose operands' maxima are all known: 256 handed-over
// partial maxima to copy into a slot, slots to zero for a producer's atomic maxima): one workgroup per job instead of 256 x 16 waves that find
// nothing to do -- a frame of inference with frozen weights issues ~40 of them.  Same result layout: slot 0 = the job's maximum, the others 0.
__global__ void __launch_bounds__(256) absmax_small_kernel(AmaxJobs J)
{
    const int job = blockIdx.x;
    const float* __restrict__ x = J.ptr[job];
    const long long len = J.len[job], n = len * J.rows[job];
    float m = 0.f;
    for (long long i = threadIdx.x; i < n; i += 256) {
        const long long r = i / (len > 0 ? len : 1);
        m = fmaxf(m, fabsf(x[r * J.stride[job] + (i - r * len)]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    J.outp[job][threadIdx.x] = threadIdx.x == 0 ? fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3])) : 0.f;
}

// Byte offset of the 16-byte chunk (row, kh) inside a plane of 32-byte rows.  The XOR puts rows r and r + 8 (same banks at a 32-byte
// pitch) on different halves, so the ds_read_b128 of 16 consecutive rows covers all 64 banks once.
__device__ __forceinline__ int chunk_off(int row, int kh) { return row * kRowB + ((kh ^ ((row >> 3) & 1)) << 4); }

constexpr int planes_of(int nterms) { return nterms == 6 ? 3 : (nterms == kF16S ? 1 : 2); }      // parts per operand that the form stores and reads

template <int WMB, int WNB, int WVM, int WVN, int NPL>
struct SplitTile {
    static constexpr int BM = 32 * WMB * WVM, BN = 32 * WNB * WVN, NT = 64 * WVM * WVN;
    static constexpr int a_bytes = NPL * BM * kRowB, b_bytes = NPL * BN * kRowB;
    static constexpr int lds_bytes = 2 * (a_bytes + b_bytes);
};

template <int WMB, int WNB>
struct SplitOperands {
    bf16x8 a[WMB][kPlanes], b[WNB][kPlanes];
};

// NTERMS = 6: all products a_i b_j with i + j <= 2; NTERMS = 3: i + j <= 1 (the third planes are not read)
template <int WMB, int WNB, int BM, int BN, int NTERMS>
__device__ __forceinline__ void read_split_operands(const char* __restrict__ As, const char* __restrict__ Bs, int wm, int wn, int lane,
                                                    SplitOperands<WMB, WNB>& o)
{
    constexpr int NPL = planes_of(NTERMS);
    const int r = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int i = 0; i < WMB; i++)
#pragma unroll
        for (int pl = 0; pl < NPL; pl++)
            o.a[i][pl] = *reinterpret_cast<const bf16x8*>(As + pl * BM * kRowB + chunk_off((wm * WMB + i) * 32 + r, kh));
#pragma unroll
    for (int j = 0; j < WNB; j++)
#pragma unroll
        for (int pl = 0; pl < NPL; pl++)
            o.b[j][pl] = *reinterpret_cast<const bf16x8*>(Bs + pl * BN * kRowB + chunk_off((wn * WNB + j) * 32 + r, kh));
}

// the products of one K tile, smallest terms first; the blocks of the wave alternate so consecutive MFMAs use different accumulators
template <int WMB, int WNB, int NTERMS>
__device__ __forceinline__ void mma_split(const SplitOperands<WMB, WNB>& o, f32x16 (&acc)[WMB][WNB])
{
    constexpr int ta[6] = { 2, 1, 0, 1, 0, 0 }, tb[6] = { 0, 1, 2, 0, 1, 0 };
#pragma unroll
    for (int t = 6 - NTERMS; t < 6; t++)
#pragma unroll
        for (int i = 0; i < WMB; i++)
#pragma unroll
            for (int j = 0; j < WNB; j++)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.a[i][ta[t]], o.b[j][tb[t]], acc[i][j], 0, 0, 0);
}

// the fp16 form: smallest terms first, like mma_split
template <int WMB, int WNB, int NTERMS = kF16>
__device__ __forceinline__ void mma_split_h(const SplitOperands<WMB, WNB>& o, f32x16 (&acc)[WMB][WNB])
{
#pragma unroll
    for (int t = (NTERMS == kF16S ? 2 : 0); t < 3; t++)
#pragma unroll
        for (int i = 0; i < WMB; i++)
#pragma unroll
            for (int j = 0; j < WNB; j++) {
                const f16x8 a = __builtin_bit_cast(f16x8, o.a[i][t == 1 ? 1 : 0]), b = __builtin_bit_cast(f16x8, o.b[j][t == 0 ? 1 : 0]);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i][j], 0, 0, 0);
            }
}
template <int WMB, int WNB>
__device__ __forceinline__ void unscale_f16(f32x16 (&acc)[WMB][WNB], float inv_a, float inv_b)
{
#pragma unroll
    for (int i = 0; i < WMB; i++)
#pragma unroll
        for (int j = 0; j < WNB; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = acc[i][j][r] * inv_a * inv_b;
}

// gather_conv_kernel on the split engine.  Same K order, same classes, same epilogue; differences: the packed weights arrive already
// split (pack_weights_split_kernel: per K tile three planes [BM][16] bf16, chunks pre-swizzled, so the loader copies 16-byte chunks
// straight through), the gathered activations are split by the loader thread between the global load and the LDS write, and one set
// of operand registers is used (two would not fit beside the staging registers at 128 VGPRs): LDS reads of tile k are issued right
// after the barrier, the split + LDS writes of tile k + 1 run under their latency, then the twelve MFMAs.
// CEXACT: the channel count is a multiple of 16 (every layer but the 3- and 12-channel inputs): no channel clamp, the KG plane
// pointers of a thread group are kernel constants in scalar registers and the channel-block advance is one scalar byte offset.
// (Round 3 carried compile-time knock-outs of the kernel's phases here -- AG_CONV_KNOCKOUT, used to bisect which phase of this kernel disturbs
// packed-fp32 instructions of a co-resident wave: profiles/r03_packed_fp32_hazard.md, stand-alone reproducer profiles/ub/pk_hazard.hip.
// Removed in round 4; commit fa8b937 has them.)

template <int WMB, int WNB, int WVM, int WVN, bool CEXACT, int NTERMS>
__global__ void __launch_bounds__(64 * WVM * WVN, AG_CONV_WAVES_PER_SIMD) gather_conv_split_kernel(GatherProblem p)
{
    constexpr int NPL = planes_of(NTERMS);
    constexpr bool F16 = is_f16_form(NTERMS);
    using T = SplitTile<WMB, WNB, WVM, WVN, NPL>;
    constexpr int BM = T::BM, BN = T::BN, NT = T::NT;
    constexpr int G = NT / BN, KG = BK / G;           // B loader: G thread groups, each KG consecutive k's of a pixel
    constexpr int AC = NPL * BM * 2;                  // A loader: 16-byte chunks in a tile
    constexpr bool A2 = AC > NT;                      // the first AC - NT threads move a second chunk
    static_assert(AC <= 2 * NT && (KG == 4 || KG == 8) && BN >= 64, "loader shapes");
    __shared__ __attribute__((aligned(16))) char smem[T::lds_bytes];
    char* const As0 = smem;
    char* const Bs0 = smem + 2 * T::a_bytes;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WVN, wn = wave % WVN;

    const TileId tw = tile_of_workgroup(p);
    int ci = 0;
#pragma unroll
    for (int c = 1; c < kMaxClasses; c++)
        if (c < p.nclasses && tw.bx >= p.cls[c].tile_begin) ci = c;
    const GatherClass& cl = p.cls[ci];
    const int gw = cl.gw, ntaps = cl.ntaps;
    const int N = cl.gh * gw;
    const GroupView gv = group_view(p, tw.by);
    const int m0 = gv.my * BM, n0 = (tw.bx - cl.tile_begin) * BN;

    const int n_loc = tid % BN, g = (wave * 64) / BN;
    const int n = n0 + n_loc;
    const bool n_ok = n < N;
    const int gy = n_ok ? n / gw : 0, gx = n_ok ? n - (n / gw) * gw : 0;
    const int iy0 = gy * p.sy, ix0 = gx * p.sx;
    uint32_t vmask = 0;
    for (int t = 0; t < ntaps; t++) {
        const int iy = iy0 + cl.dy[t], ix = ix0 + cl.dx[t];
        if (n_ok && iy >= 0 && iy < p.Hg && ix >= 0 && ix < p.Wg) vmask |= 1u << t;
    }
    const int pix = iy0 * p.Wg + ix0;
    const int tl = lane & (kMaxTaps - 1);
    const int toff_vec = cl.dy[tl] * p.Wg + cl.dx[tl];

    f32x16 acc[WMB][WNB];
#pragma unroll
    for (int i = 0; i < WMB; i++)
#pragma unroll
        for (int j = 0; j < WNB; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int nkt_all = cl.nkt;
    const int kt_beg = min(nkt_all, (int)blockIdx.z * p.kt_per_split), kt_end = min(nkt_all, kt_beg + p.kt_per_split);
    const int nkt = kt_end - kt_beg;
    int t_cur = kt_beg % ntaps;
    int c_cur = (kt_beg / ntaps) * BK + g * KG;
    const size_t plane_b = (size_t)p.Hg * p.Wg * sizeof(float);
    const char* const xin_b = reinterpret_cast<const char*>(gv.xin);
    const int c_last = p.Cg - 1;
    // packed weights: cl.at_off counts fp32-tile floats (BM * 16 per tile); a split tile has a_bytes = 2 bytes per element and plane
    const char* a_ptr = reinterpret_cast<const char*>(p.At) + ((size_t)gv.grp * p.at_gs + (size_t)cl.at_off) * (2 * NPL) + ((size_t)gv.my * nkt_all + kt_beg) * T::a_bytes;
    TensorScale sa{ 1.f, 1.f }, sb{ 1.f, 1.f };
    if constexpr (F16) {
        sa = tensor_scale(p.amax_a + (size_t)gv.grp * kAmaxParts, p.amax_a_mult, lane);
        sb = tensor_scale(p.amax_b + (p.x_gs ? (size_t)gv.grp * kAmaxParts : 0), 1.f, lane);
    }

    const uint32_t a_voff0 = (uint32_t)min(tid, AC - 1) * 16u;
    const uint32_t a_voff1 = (uint32_t)min(NT + tid, AC - 1) * 16u;
    const bool a_thread0 = tid < AC;                  // wave-uniform (AC is a multiple of 64)
    const bool a_thread1 = A2 && NT + tid < AC;
    // LDS write position of this thread's KG values: row n_loc, k = g * KG ...
    const int b_woff = chunk_off(n_loc, (g * KG) >> 3) + ((g * KG) & 7) * 2;

    // uniform base pointers of the KG channel planes of the current channel block (scalar registers; refreshed once per block)
    const char* cbase[KG];
    uint32_t cb_off = 0;                              // CEXACT: byte offset of the current channel block (host checks the tensor < 4 GB)
    const uint32_t cb_step = (uint32_t)(BK * plane_b);
    auto set_bases = [&]() {
#pragma unroll
        for (int j = 0; j < KG; j++) cbase[j] = xin_b + (size_t)min(c_cur + j, c_last) * plane_b;
    };
    set_bases();

    struct Stage {
        u32x4 ra0, ra1;
        float rb[KG];
        bool tap_ok;
    };
    Stage S[2];
    SplitOperands<WMB, WNB> O;
    auto gload = [&](Stage& st) {
        st.ra0 = *reinterpret_cast<const u32x4*>(a_ptr + a_voff0);
        if constexpr (A2) st.ra1 = *reinterpret_cast<const u32x4*>(a_ptr + a_voff1);
        a_ptr += T::a_bytes;
        st.tap_ok = (vmask >> t_cur) & 1u;
        const int toff = __builtin_amdgcn_readlane(toff_vec, t_cur);
        const uint32_t voff = st.tap_ok ? (uint32_t)(pix + toff) * 4u + (CEXACT ? cb_off : 0u) : 0u;
#pragma unroll
        for (int j = 0; j < KG; j++) {
            st.rb[j] = *reinterpret_cast<const float*>(cbase[j] + voff);     // global_load_dword v, v_off, s[base]
        }
        t_cur++;
        if constexpr (CEXACT) {
            const bool wrap = t_cur == ntaps;
            t_cur = wrap ? 0 : t_cur;
            cb_off += wrap ? cb_step : 0u;
        } else {
            if (t_cur == ntaps) { t_cur = 0; c_cur += BK; set_bases(); }
        }
    };
    auto lstore = [&](int buf, const Stage& st) {
        char* As = As0 + buf * T::a_bytes;
        char* Bs = Bs0 + buf * T::b_bytes;
        if (a_thread0) *reinterpret_cast<u32x4*>(As + tid * 16) = st.ra0;
        if constexpr (A2) {
            if (a_thread1) *reinterpret_cast<u32x4*>(As + (NT + tid) * 16) = st.ra1;
        }
#pragma unroll
        for (int q = 0; q < KG / 4; q++) {
            u32x2 w0, w1, w2;
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const float x0 = st.tap_ok ? st.rb[4 * q + 2 * e] : 0.f, x1 = st.tap_ok ? st.rb[4 * q + 2 * e + 1] : 0.f;
                uint32_t a, b = 0, c = 0;
                if constexpr (NTERMS == kF16S) a = pack_f16(x0 * sb.s, x1 * sb.s);
                else if constexpr (F16)        split_pair_h(x0 * sb.s, x1 * sb.s, a, b);
                else                           split_pair(x0, x1, a, b, c);
                w0[e] = a; w1[e] = b; w2[e] = c;
            }
            *reinterpret_cast<u32x2*>(Bs + 0 * BN * kRowB + b_woff + 8 * q) = w0;
            if constexpr (NPL >= 2) *reinterpret_cast<u32x2*>(Bs + 1 * BN * kRowB + b_woff + 8 * q) = w1;
            if constexpr (NPL == 3) *reinterpret_cast<u32x2*>(Bs + 2 * BN * kRowB + b_woff + 8 * q) = w2;
        }
    };
    auto lread = [&](int buf) {
        read_split_operands<WMB, WNB, BM, BN, NTERMS>(As0 + buf * T::a_bytes, Bs0 + buf * T::b_bytes, wm, wn, lane, O);
    };
    auto mma = [&]() {
        if constexpr (F16) mma_split_h<WMB, WNB, NTERMS>(O, acc);
        else               mma_split<WMB, WNB, NTERMS>(O, acc);
    };

    if (nkt > 0) {
        gload(S[0]);
        if (nkt > 1) gload(S[1]);
        lstore(0, S[0]);
        lds_barrier();
        // steady state for tile kt (kt + 2 < nkt), branch-free
        auto step_full = [&](int kt, Stage& s_same, Stage& s_next) {
            lread(kt & 1);
            gload(s_same);                         // tile kt + 2 into the registers whose tile (kt) is already in LDS
            lstore((kt + 1) & 1, s_next);
            mma();
            lds_barrier();
        };
        auto step_tail = [&](int kt, Stage& s_next, bool has_next) {
            lread(kt & 1);
            if (has_next) lstore((kt + 1) & 1, s_next);
            mma();
            if (has_next) lds_barrier();
        };
        int kt = 0;
        for (; kt + 3 < nkt; kt += 2) {
            step_full(kt, S[0], S[1]);
            step_full(kt + 1, S[1], S[0]);
        }
        const int rem = nkt - kt;
        if (rem == 3) {
            step_full(kt, S[0], S[1]);
            step_tail(kt + 1, S[0], true);
            step_tail(kt + 2, S[1], false);
        } else if (rem == 2) {
            step_tail(kt, S[1], true);
            step_tail(kt + 1, S[0], false);
        } else {
            step_tail(kt, S[1], false);
        }
    }
    if constexpr (F16) unscale_f16<WMB, WNB>(acc, sa.inv, sb.inv);
    gather_epilogue<WMB, WNB>(p, gv, cl, acc, m0, n0, N, wm, wn, lane);
}

// ------------------------------------------------------------------------------------------------------------------
// Round 6: the same GEMM with an LDS-DMA loader over PRE-SPLIT activation planes
// ------------------------------------------------------------------------------------------------------------------
// gather_conv_split_kernel converts every gathered activation fp32 -> (hi, lo) fp16 in the loader thread, once per TAP and per OUTPUT-CHANNEL
// TILE that reads it (18 times on a 256 -> 256 layer), through VGPRs: 8 scalar global loads, ~40 VALU and the LDS stores per thread and
// 16-deep K tile -- as much issue time as the tile's MFMAs, serial with them inside a wave (PMC, profiles/r05_pmc_conv_modes: matrix pipe
// 48 % busy).  Here the split happens ONCE per launch, in a streaming pass of its own (presplit_kernel: fp32 [C][H][W] -> fp16 planes
// [plane][C / 8][H W][8], 4 bytes per element like the fp32 tensor), and a K tile's operands reach LDS by global_load_lds_dwordx4: a lane's
// 16-byte chunk = 8 consecutive channels of ONE pixel of one plane, so the lane-linear LDS image of a DMA (lane i -> base + 16 i) is the
// swizzled [row][chunk] layout read_split_operands wants once the lane picks the matching (row, chunk) as its SOURCE; a tap that falls
// outside the image, or a row past the class, reads a 64-byte zero page instead (the source address is per lane, so padding costs two
// selects).  No staging registers, no conversion, no LDS stores; per K tile a wave issues its DMAs (one per 1-KB piece: a 32-row slice of
// an activation plane, or 1 KB of the packed weight tile, which already is its LDS image) and twelve MFMAs on a 64 x 64 wave tile.  Three
// LDS buffers, DMAs two tiles ahead, one barrier per tile: wait vmcnt(own DMAs of ONE tile) -> barrier -> issue tile kt + 2 -> read -> MFMA.
__device__ __attribute__((aligned(64))) uint32_t g_zero_page[16];

struct PresplitProblem {
    const float* x;            // [G'][C][HW]
    char* xs;                  // [G'][planes][Cpad / 8][HW][8] fp16
    const float* amax;         // [G' or 1][kAmaxParts]
    long long x_gs;            // floats between instances (0: one shared instance)
    long long xs_gs;           // bytes between instances of xs
    int C, Cpad, HW, planes;
};
__global__ void __launch_bounds__(256) presplit_kernel(PresplitProblem p)
{
    const int grp = blockIdx.z, cb = blockIdx.y;
    const TensorScale sc = tensor_scale(p.amax + (p.x_gs ? (size_t)grp * kAmaxParts : 0), 1.f, threadIdx.x & 63);
    const float* __restrict__ x = p.x + (size_t)grp * p.x_gs;
    char* __restrict__ xs = p.xs + (size_t)grp * p.xs_gs;
    const size_t plane_b = (size_t)(p.Cpad / 8) * p.HW * 16;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < p.HW; q += gridDim.x * 256) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int c = cb * 8 + j;
            v[j] = c < p.C ? x[(size_t)c * p.HW + q] * sc.s : 0.f;
        }
        u32x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            uint32_t a, b;
            split_pair_h(v[2 * e], v[2 * e + 1], a, b);
            h[e] = a; l[e] = b;
        }
        const size_t off = ((size_t)cb * p.HW + q) * 16;
        *reinterpret_cast<u32x4*>(xs + off) = h;
        if (p.planes == 2) *reinterpret_cast<u32x4*>(xs + plane_b + off) = l;
    }
}

// waves per SIMD the register budget is cut for: one 8-wave workgroup per CU for the 128 x 64 wave tiles (128 accumulator registers), two for the
// 64 x 64 ones, three 4-wave workgroups
constexpr int dma_wps(int wmb, int wnb, int nt) { return wmb * wnb > 4 ? 2 : (nt == 512 ? 4 : 3); }
template <int WMB, int WNB, int WVM, int WVN, int NTERMS>
__global__ void __launch_bounds__(64 * WVM * WVN, dma_wps(WMB, WNB, 64 * WVM * WVN)) gather_conv_dma_kernel(GatherProblem p, const char* __restrict__ xs, long long xs_gs)
{
    constexpr int NPL = planes_of(NTERMS);
    static_assert(is_f16_form(NTERMS), "the DMA loader serves the fp16 forms");
    using T = SplitTile<WMB, WNB, WVM, WVN, NPL>;
    constexpr int BM = T::BM, BN = T::BN, NT = T::NT, NW = NT / 64;
    constexpr int PA = T::a_bytes / 1024, PB = T::b_bytes / 1024;      // 1-KB pieces of a K tile's two operand images
    static_assert((PA + PB) % NW == 0, "every wave issues the same number of DMAs per tile (the vmcnt count is an immediate)");
    static_assert(PA % NW == 0, "piece k of every wave is on the same side (A or B): no branch per piece");
    constexpr int PER = (PA + PB) / NW;
    constexpr int NBUF = 3;
    constexpr int stage_bytes = T::a_bytes + T::b_bytes;
    __shared__ __attribute__((aligned(1024))) char smem[NBUF * stage_bytes];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WVN, wn = wave % WVN;

    const TileId tw = tile_of_workgroup(p);
    int ci = 0;
#pragma unroll
    for (int c = 1; c < kMaxClasses; c++)
        if (c < p.nclasses && tw.bx >= p.cls[c].tile_begin) ci = c;
    const GatherClass& cl = p.cls[ci];
    const int gw = cl.gw, ntaps = cl.ntaps;
    const int N = cl.gh * gw;
    const GroupView gv = group_view(p, tw.by);
    const int m0 = gv.my * BM, n0 = (tw.bx - cl.tile_begin) * BN;
    const int HW = p.Hg * p.Wg;

    f32x16 acc[WMB][WNB];
#pragma unroll
    for (int i = 0; i < WMB; i++)
#pragma unroll
        for (int j = 0; j < WNB; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int nkt_all = cl.nkt;
    const int kt_beg = min(nkt_all, (int)blockIdx.z * p.kt_per_split), kt_end = min(nkt_all, kt_beg + p.kt_per_split);
    const int nkt = kt_end - kt_beg;
    TensorScale sa = tensor_scale(p.amax_a + (size_t)gv.grp * kAmaxParts, p.amax_a_mult, lane);
    TensorScale sb = tensor_scale(p.amax_b + (p.x_gs ? (size_t)gv.grp * kAmaxParts : 0), 1.f, lane);

    // ---- this wave's pieces.  Piece q < PA: bytes [1024 q, 1024 q + 1024) of the packed weight tile; piece PA + j: plane j / (BN / 32), rows
    //      32 (j % (BN / 32)) ... + 31 of the activation tile; lane i of a B piece owns LDS chunk i = (row i / 2, slot i & 1), slot = kh ^ bit 3 of row
    const char* const a_tile0 = reinterpret_cast<const char*>(p.At) + ((size_t)gv.grp * p.at_gs + (size_t)cl.at_off) * (2 * NPL) + ((size_t)gv.my * nkt_all + kt_beg) * T::a_bytes;
    const char* const xs_g = xs + (size_t)(p.x_gs ? gv.grp : 0) * xs_gs;
    const size_t plane_b = (size_t)(p.Cpad / 8) * HW * 16;
    const char* src[PER];          // per-lane source of piece k at tile 0 / tap offset 0 / channel block 0
    uint32_t vmask[PER];           // B pieces: taps of this lane's pixel that lie inside the image; A pieces: all ones
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int q = wave + k * NW;                       // wave-uniform
        if (k * NW < PA) {
            src[k] = a_tile0 + q * 1024 + lane * 16;
            vmask[k] = 0xffffffffu;
        } else {
            const int j = q - PA, pl = j / (BN / 32), rp = j % (BN / 32);
            const int row = rp * 32 + (lane >> 1), kh = (lane & 1) ^ ((row >> 3) & 1);
            const int n = n0 + row;
            const bool n_ok = n < N;
            const int gy = n_ok ? n / gw : 0, gx = n_ok ? n - (n / gw) * gw : 0;
            const int iy0 = gy * p.sy, ix0 = gx * p.sx;
            uint32_t vm = 0;
            for (int t = 0; t < ntaps; t++) {
                const int iy = iy0 + cl.dy[t], ix = ix0 + cl.dx[t];
                if (n_ok && iy >= 0 && iy < p.Hg && ix >= 0 && ix < p.Wg) vm |= 1u << t;
            }
            vmask[k] = vm;
            src[k] = xs_g + (size_t)pl * plane_b + ((long long)kh * HW + (long long)iy0 * p.Wg + ix0) * 16;
        }
    }
    const int tl = lane & (kMaxTaps - 1);
    const int toff_vec = cl.dy[tl] * p.Wg + cl.dx[tl];
    const char* const zero = reinterpret_cast<const char*>(g_zero_page) + (lane & 3) * 16;

    int t_cur = kt_beg % ntaps;
    long long cb_off = (long long)(kt_beg / ntaps) * 2 * HW * 16;      // byte offset of the tile's channel-block pair inside a plane
    long long a_off = 0;
    auto issue = [&](int buf) {
        char* const stage = smem + buf * stage_bytes;
        const long long uoff = cb_off + (long long)__builtin_amdgcn_readlane(toff_vec, t_cur) * 16;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int q = wave + k * NW;
            const char* g;
            if (k * NW < PA) g = src[k] + a_off;
            else          g = ((vmask[k] >> t_cur) & 1u) ? src[k] + uoff : zero;
            char* dst = stage + q * 1024;                  // A pieces first, then the B planes: the stage IS [a image][b image]
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
        a_off += T::a_bytes;
        t_cur++;
        const bool wrap = t_cur == ntaps;
        t_cur = wrap ? 0 : t_cur;
        cb_off += wrap ? (long long)2 * HW * 16 : 0;
    };
    SplitOperands<WMB, WNB> O;

    if (nkt > 0) {
        issue(0);
        if (nkt > 1) issue(1);
        int buf = 0;
        for (int kt = 0; kt < nkt; kt++) {
            if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER) : "memory");
            else              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
            if (kt + 2 < nkt) issue(buf >= 1 ? buf - 1 : NBUF - 1);          // (kt + 2) % 3
            const char* stage = smem + buf * stage_bytes;
            read_split_operands<WMB, WNB, BM, BN, NTERMS>(stage, stage + T::a_bytes, wm, wn, lane, O);
            mma_split_h<WMB, WNB, NTERMS>(O, acc);
            buf = buf + 1 == NBUF ? 0 : buf + 1;
        }
    }
    unscale_f16<WMB, WNB>(acc, sa.inv, sb.inv);
    gather_epilogue<WMB, WNB>(p, gv, cl, acc, m0, n0, N, wm, wn, lane);
}

// split-K finish: y = (sum_z partial[z][m][col]) * out_scale[m] + bias[m], in a fixed order (deterministic)
__global__ void __launch_bounds__(256) reduce_splits_kernel(GatherProblem p, int splits)
{
    const int Nc = p.Ncols;
    const long long total = (long long)p.M * Nc;
    const size_t zstride = (size_t)p.Mpad * Nc;
    const int grp = blockIdx.y;                                   // grouped launch: one grid row per instance
    const float* __restrict__ partial = p.partial + (size_t)grp * p.part_gs;
    const float* __restrict__ bias = p.bias_t.p[grp];
    float* __restrict__ yout = p.yout + (size_t)grp * p.y_gs;
    const float* __restrict__ noise = p.act ? p.noise_t.p[grp] : nullptr;
    const float nw = (p.act == 1 && noise) ? p.nw_t.p[grp][0] : 1.0f;
    float mx = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int m = (int)(i / Nc), colg = (int)(i - (long long)m * Nc);
        float v = 0.f;
        for (int z = 0; z < splits; z++) v += partial[z * zstride + i];
        if (p.act) {        // plain gather (one class, os = 1): column = output pixel
            const float nv = noise ? (p.act == 2 ? noise[(size_t)m * Nc + colg] : noise[colg]) : 0.f;
            const float t = fmaf(nw, nv, v) + (bias ? bias[m] : 0.f);
            const float y = (t > 0.f ? t : t * p.slope) * p.act_scale;
            yout[(size_t)m * Nc + colg] = y;
            mx = fmaxf(mx, fabsf(y));
            continue;
        }
        if (p.out_scale) v *= p.out_scale[m];
        if (bias) v += bias[m];
        int ci = 0;
#pragma unroll
        for (int c = 1; c < kMaxClasses; c++)
            if (c < p.nclasses && colg >= p.cls[c].col_begin) ci = c;
        const int nn = colg - p.cls[ci].col_begin;
        const int oy = nn / p.cls[ci].gw, ox = nn - oy * p.cls[ci].gw;
        yout[(size_t)m * p.OHf * p.OWf + (size_t)(p.cls[ci].y0 + oy * p.os) * p.OWf + (p.cls[ci].x0 + ox * p.os)] = v;
    }
    if (p.act && p.out_amax) {        // one non-returning atomic per workgroup (these workgroups are short: nothing may wait at their end)
        __shared__ float s_mx[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if ((threadIdx.x & 63) == 0) s_mx[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0)
            atomicMax(reinterpret_cast<unsigned int*>(p.out_amax) + (size_t)grp * kAmaxParts + (blockIdx.x & (kAmaxParts - 1)),
                      __float_as_uint(fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]))));
    }
}

// weight re-pack into the tile-blocked images of all classes in one launch:
//   At[class][m / BM][kt][m % BM][kl] = w[c * stride_c + m * stride_m + tapoff[class][t]]
// with K tile kt = (c / 16) * ntaps + t and kl = c % 16, zero padded; optional exact scalar pre-multiplication (EqualConv2d's
// `weight * scale`, dual_styleunet.py:114-117: the same fp32 product the reference forms before its convolution).
// One workgroup moves a 16 (m) x 16 (c) x k*k block through LDS: in the source one of the two indices is contiguous with the taps
// (16 * k*k floats per row of the other index: coalesced reads), in the destination every (class, tap) is one 1-KB run [16 m][16 c]
// (coalesced writes) -- 38 MB for the 1024 -> 512 layer in ~15 us instead of 90 with a gather per element.
struct PackProblem {
    PtrTable w_t;            // weights of the instances (blockIdx.z)
    long long at_gs;         // packed elements between the instances' images
    float* At;
    int C, Cpad, M, Mpad, BM, nclasses, k2;
    long long stride_c, stride_m;
    float wscale;
    int has_wscale;
    int planes;              // split engine: parts stored per element (planes_of)
    const float* amax;       // fp16 split form: partial maxima of the instances' weight tensors [G][kAmaxParts]; null: the bf16 forms
    int ntaps[kMaxClasses], zero[kMaxClasses];
    long long begin[kMaxClasses + 1];        // element offsets of the classes inside At
    int tapoff[kMaxClasses][kMaxTaps];
};

// 16 x 16 x k*k block of the natural weight layout -> LDS.  Thread (outer o, inner i) owns the k*k taps of one (row, channel) pair: they
// are contiguous in memory and the pairs of consecutive threads follow each other, so a wave reads one contiguous run; all loads of a
// thread are issued before the first LDS write (the earlier element-indexed loop ran k*k dependent iterations with two integer
// divisions each).
__device__ __forceinline__ void load_weight_block(float (&blk)[16][16 * kMaxTaps + 1], const PackProblem& p, int cb, int m16, int tid,
                                                  bool m_inner)
{
    const int o = tid >> 4, inner = tid & 15, k2 = p.k2;
    const int c = cb * 16 + (m_inner ? o : inner), m = m16 * 16 + (m_inner ? inner : o);
    const bool ok = c < p.C && m < p.M;
    const float* src = p.w_t.p[blockIdx.z] + (ok ? (long long)c * p.stride_c + (long long)m * p.stride_m : 0);
    float v[kMaxTaps];
#pragma unroll
    for (int t = 0; t < kMaxTaps; t++) v[t] = (ok && t < k2) ? src[t] : 0.f;
    const float ws = p.has_wscale ? p.wscale : 1.f;
#pragma unroll
    for (int t = 0; t < kMaxTaps; t++)
        if (t < k2) blk[o][inner * k2 + t] = p.has_wscale ? v[t] * ws : v[t];
}

__global__ void __launch_bounds__(256) pack_weights_kernel(PackProblem p)
{
    __shared__ float blk[16][16 * kMaxTaps + 1];
    const int tid = threadIdx.x;
    const int cb = blockIdx.x, m16 = blockIdx.y;             // channel block, 16-row group
    const int k2 = p.k2;
    const bool m_inner = p.stride_m == k2;                   // else stride_c == k2: channels contiguous with the taps
    // load: 16 rows of the outer index, each one contiguous run of 16 * k2 floats
    load_weight_block(blk, p, cb, m16, tid, m_inner);
    __syncthreads();
    const int ml = tid >> 4, kl = tid & 15;                  // destination element (m, c) of the block
    const int m = m16 * 16 + ml;
    const int mt = m / p.BM, mrow = m - mt * p.BM;
    const int o = m_inner ? kl : ml, inner = m_inner ? ml : kl;
    const int ctiles = p.Cpad / BK;
    for (int ci = 0; ci < p.nclasses; ci++) {
        const int ntaps = p.ntaps[ci];
        const long long nkt = (long long)ntaps * ctiles;
        float* dst = p.At + (size_t)blockIdx.z * p.at_gs + p.begin[ci] + ((mt * nkt + (long long)cb * ntaps) * p.BM + mrow) * 16 + kl;
        const bool zero = p.zero[ci] != 0;
        for (int t = 0; t < ntaps; t++) dst[(long long)t * p.BM * 16] = zero ? 0.f : blk[o][inner * k2 + p.tapoff[ci][t]];
    }
}

// pack_weights_kernel for the split engine: At[class][m / BM][kt][plane][m % BM][16 bf16, chunks swizzled as chunk_off()], the
// fp32 value (after the optional scalar pre-multiplication) split three ways.  A thread owns two adjacent channels of a row
// (one 32-bit word per plane) and every other tap.
__global__ void __launch_bounds__(256) pack_weights_split_kernel(PackProblem p)
{
    __shared__ float blk[16][16 * kMaxTaps + 1];
    const int tid = threadIdx.x;
    const int cb = blockIdx.x, m16 = blockIdx.y;
    const int k2 = p.k2;
    const bool m_inner = p.stride_m == k2;
    load_weight_block(blk, p, cb, m16, tid, m_inner);
    __syncthreads();
    const int ml = (tid >> 3) & 15, kp = tid & 7, th = tid >> 7;      // row, channel pair (2 kp, 2 kp + 1), tap parity
    const int m = m16 * 16 + ml;
    const int mt = m / p.BM, mrow = m - mt * p.BM;
    const int o0 = m_inner ? 2 * kp : ml, o1 = m_inner ? 2 * kp + 1 : ml;
    const int i0 = (m_inner ? ml : 2 * kp) * k2, i1 = (m_inner ? ml : 2 * kp + 1) * k2;
    const int ctiles = p.Cpad / BK;
    const size_t plane = (size_t)p.BM * kRowB;
    const int npl = p.planes;
    const bool f16 = p.amax != nullptr;
    float sc = 1.f;
    if (f16) sc = tensor_scale(p.amax + (size_t)blockIdx.z * kAmaxParts, p.has_wscale ? fabsf(p.wscale) : 1.f, tid & 63).s;
    for (int ci = 0; ci < p.nclasses; ci++) {
        const int ntaps = p.ntaps[ci];
        const long long nkt = (long long)ntaps * ctiles;
        char* base = reinterpret_cast<char*>(p.At) + ((size_t)blockIdx.z * p.at_gs + (size_t)p.begin[ci]) * (2 * npl) + (size_t)(mt * nkt + (long long)cb * ntaps) * (npl * plane)
                     + chunk_off(mrow, kp >> 2) + (kp & 3) * 4;
        const bool zero = p.zero[ci] != 0;
        for (int t = th; t < ntaps; t += 2) {
            const int to = p.tapoff[ci][t];
            const float v0 = zero ? 0.f : blk[o0][i0 + to], v1 = zero ? 0.f : blk[o1][i1 + to];
            uint32_t a, b = 0, c = 0;
            if (f16 && npl == 1) a = pack_f16(v0 * sc, v1 * sc);
            else if (f16)        split_pair_h(v0 * sc, v1 * sc, a, b);
            else                 split_pair(v0, v1, a, b, c);
            char* d = base + (size_t)t * (npl * plane);
            *reinterpret_cast<uint32_t*>(d) = a;
            if (npl >= 2) *reinterpret_cast<uint32_t*>(d + plane) = b;
            if (npl == 3) *reinterpret_cast<uint32_t*>(d + 2 * plane) = c;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// wgrad: C[m][(c, t)] += sum over a slice of pixels of A[m][pix] * Xin[c][gy*sy + dy_t][gx*sx + dx_t]
// ------------------------------------------------------------------------------------------------------------------
struct WgradProblem {
    const float* a;        // [Mw][gh * gw]
    const float* xin;      // [Cg][Hg][Wg]
    float* c;              // [Mw][Cg * ntaps]; every element is written (round 5: no zero fill, no atomics)
    int Mw, Cg, Hg, Wg, gh, gw, sy, sx, ntaps;
    int ksplit_len;        // pixels per split (multiple of BK)
    float wscale;          // the forward convolved with w * wscale: dL/dw = wscale * dL/d(w * wscale)
    int G, mtiles;         // grouped launch: gridDim.y = G * mtiles
    long long a_gs, xin_gs, c_gs;     // floats between the instances' operands / outputs
    PtrTable c_t;          // p[0] != null: instance g writes to c_t.p[g] instead of c + g * c_gs (ConvOpts::dw_table)
    float* partial;        // non-null: every workgroup stores its slice's tile to partial[z][instance][Mpad][Npad] and wgrad_reduce_kernel adds the
                           // slices (and the instances that share a destination) in a fixed order; null: one slice, stored straight to the destination
    int Mpad, Npad;
    int* status;           // host-visible sticky flag raised on a non-finite accumulator (ag_conv_status), or null
    int status_tag;        // what the offending launch stores there: kind << 28 | rows << 14 | channels (status_tag_of)
    const float* amax_a;   // fp16 split form: partial maxima of `a` and of `xin` ([G][kAmaxParts], one row where the instances share the tensor)
    const float* amax_b;
    int c_row_stride, c_chan_stride;  // element (m, nn = (channel, tap)) of the output sits at m * c_row_stride + channel * c_chan_stride + tap:
                                      // (Nw, ntaps) = the natural [Mw][Cg][taps]; (ntaps, Mw * ntaps) = [Cg][Mw][taps] (a transposed convolution's
                                      // weight gradient written in the [Cout][Cin][k][k] layout its modulated weight is kept in)
    int dy[kMaxTaps], dx[kMaxTaps];
};

// Epilogue of the weight-gradient kernels (round 5: deterministic).  Rounds 1-4 added every pixel slice's tile to a zeroed dw with float
// atomics: the sum's order was the slices' arrival order, so a weight gradient differed from run to run in its last bits (and through the
// leaky-ReLU slope selections downstream, occasionally in much more).  Now a slice stores its tile -- the only slice straight into dw, several
// into partial[z][instance][Mpad][Npad] -- and wgrad_reduce_kernel adds them in slice order.  No zero fill of dw, no atomics, same bits every run.
template <int WMB, int WNB>
__device__ __forceinline__ void wgrad_store(const WgradProblem& p, const f32x16 (&acc)[WMB][WNB], float unscale, int grp, float* __restrict__ c_g, int m0,
                                            int n0, int Nw, int wm, int wn, int lane)
{
    const int col = lane & 31, rbase = 4 * (lane >> 5);
    const float f = unscale * p.wscale;
    flag_non_finite<WMB, WNB>(p.status, p.status_tag, acc);
    if (p.partial) {
        float* __restrict__ part = p.partial + ((size_t)blockIdx.z * p.G + grp) * p.Mpad * p.Npad;
#pragma unroll
        for (int i = 0; i < WMB; i++)
#pragma unroll
            for (int j = 0; j < WNB; j++) {
                const int nn = n0 + (wn * WNB + j) * 32 + col;           // < Npad by construction
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int m = m0 + (wm * WMB + i) * 32 + (r & 3) + 8 * (r >> 2) + rbase;
                    part[(size_t)m * p.Npad + nn] = acc[i][j][r] * f;
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < WMB; i++)
#pragma unroll
        for (int j = 0; j < WNB; j++) {
            const int nn = n0 + (wn * WNB + j) * 32 + col;
            if (nn >= Nw) continue;
            const int ch = nn / p.ntaps;
            float* cn = c_g + (size_t)ch * p.c_chan_stride + (nn - ch * p.ntaps);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = m0 + (wm * WMB + i) * 32 + (r & 3) + 8 * (r >> 2) + rbase;
                if (m < p.Mw) cn[(size_t)m * p.c_row_stride] = acc[i][j][r] * f;     // natural layout: 32 lanes = 128 contiguous bytes
            }
        }
}

// dw = the sum of the slices' tiles, and of the instances that share a destination (ConvOpts::dw_table with repeated entries: the members of a
// network in the comb convolutions), in a fixed order.  grid (blocks, runs); run r = instances [begin[r], begin[r + 1]) -> dst.p[r].
struct WgradReduce {
    const float* partial;
    PtrTable dst;
    int begin[kMaxGroups + 1];
    int G, splits, Mw, Nw, Mpad, Npad, ntaps, c_row_stride, c_chan_stride;
};
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(WgradReduce r)
{
    const int run = blockIdx.y;
    const int g0 = r.begin[run], g1 = r.begin[run + 1];
    float* __restrict__ dst = const_cast<float*>(r.dst.p[run]);
    const size_t inst = (size_t)r.Mpad * r.Npad, zstride = inst * r.G;
    const long long total = (long long)r.Mw * r.Nw;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int m = (int)(i / r.Nw), nn = (int)(i - (long long)m * r.Nw);
        const float* __restrict__ src = r.partial + (size_t)m * r.Npad + nn;
        float v = 0.f;
        for (int g = g0; g < g1; g++)
            for (int z = 0; z < r.splits; z++) v += src[(size_t)z * zstride + (size_t)g * inst];
        const int ch = nn / r.ntaps;
        dst[(size_t)m * r.c_row_stride + (size_t)ch * r.c_chan_stride + (nn - ch * r.ntaps)] = v;
    }
}

// The same sum for MANY terms (round 5, third session).  The 64-row layers at 512 x 512 (the per-view stage of the colour network) have 10 output
// tiles, so their pixel loop is cut into 28 slices; one thread per element then walks all of them in a single dependent chain of additions
// behind loads that the compiler issues four at a time, on 1 150 waves in all: ~38 us per launch for 9 MB, four launches per view.  Here an element's
// terms (term t = (instance - g0) * splits + slice) are dealt to PAR = 256 / EPB threads -- thread (e, j) of a workgroup owns element e of the
// workgroup's EPB consecutive elements (EPB lanes = one coalesced segment) and adds the terms t = j, j + PAR, ... in ascending order, eight loads in
// flight -- and thread (e, 0) adds the PAR partial sums in ascending j.  A fixed order: the same bits on every run (not the bits of the
// one-thread kernel: a different association of the same terms).
template <int EPB>
__global__ void __launch_bounds__(256) wgrad_reduce_par_kernel(WgradReduce r)
{
    constexpr int PAR = 256 / EPB;
    __shared__ float s_part[PAR][EPB];
    const int run = blockIdx.y;
    const int g0 = r.begin[run], g1 = r.begin[run + 1];
    float* __restrict__ dst = const_cast<float*>(r.dst.p[run]);
    const size_t inst = (size_t)r.Mpad * r.Npad, zstride = inst * r.G;
    const long long total = (long long)r.Mw * r.Nw;
    const int e = threadIdx.x % EPB, j = threadIdx.x / EPB;
    const int T = (g1 - g0) * r.splits;
    for (long long base = (long long)blockIdx.x * EPB; base < total; base += (long long)gridDim.x * EPB) {
        const long long i = base + e;
        const bool ok = i < total;
        const int m = ok ? (int)(i / r.Nw) : 0, nn = ok ? (int)(i - (long long)m * r.Nw) : 0;
        const float* __restrict__ src = r.partial + (size_t)m * r.Npad + nn;
        float v = 0.f;
        int t = j;
        for (; t + 7 * PAR < T; t += 8 * PAR) {
            float x[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int tt = t + u * PAR, g = g0 + tt / r.splits, z = tt - (tt / r.splits) * r.splits;
                x[u] = src[(size_t)z * zstride + (size_t)g * inst];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) v += x[u];
        }
        for (; t < T; t += PAR) {
            const int g = g0 + t / r.splits, z = t - (t / r.splits) * r.splits;
            v += src[(size_t)z * zstride + (size_t)g * inst];
        }
        s_part[j][e] = v;
        __syncthreads();
        if (j == 0 && ok) {
            float acc = s_part[0][e];
#pragma unroll
            for (int q = 1; q < PAR; q++) acc += s_part[q][e];
            const int ch = nn / r.ntaps;
            dst[(size_t)m * r.c_row_stride + (size_t)ch * r.c_chan_stride + (nn - ch * r.ntaps)] = acc;
        }
        __syncthreads();
    }
}

// AVEC: the rows of A are 16-byte aligned (pixel count a multiple of 4, always true in the product): one dwordx4 per thread and
// tile; the scalar form is kept for arbitrary sizes.  Everything in the K loop is branch-free: loads are unconditional from clamped
// addresses, validity (image border, end of the K slice) travels as a bit mask and is applied at the LDS write.
template <int WMB, int WNB, int WVM, int WVN, bool AVEC>
__global__ void __launch_bounds__(64 * WVM * WVN, AG_CONV_WAVES_PER_SIMD) wgrad_kernel(WgradProblem p)
{
    using T = Tile<WMB, WNB, WVM, WVN>;
    constexpr int BM = T::BM, BN = T::BN, NT = T::NT;
    constexpr int BSTEP = NT / 16;                     // B loader: columns covered per pass
    constexpr int BC = BN / BSTEP;                     // columns per thread
    static_assert(BM * 4 <= NT, "A loader: one row quarter per thread");
    __shared__ __attribute__((aligned(16))) float smem[T::lds_floats];
    float* const As0 = smem;
    float* const Bs0 = smem + 2 * BM * LDK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WVN, wn = wave % WVN;
    const int Kp = p.gh * p.gw, Nw = p.Cg * p.ntaps;
    const int grp = (int)blockIdx.y / p.mtiles, my = (int)blockIdx.y - grp * p.mtiles;      // grouped launch (ag_groups.h)
    const float* __restrict__ xin_g = p.xin + (size_t)grp * p.xin_gs;
    float* __restrict__ c_g = p.c_t.p[0] ? const_cast<float*>(p.c_t.p[grp]) : p.c + (size_t)grp * p.c_gs;
    const int m0 = my * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * p.ksplit_len, kend = min(Kp, kbeg + p.ksplit_len);
    if (kbeg >= kend) return;

    // A loader: row am, 4 consecutive pixels aq..aq+3 (A is pixel-contiguous) -> one dwordx4 load, one b128 LDS write
    const int am = tid >> 2, aq = (tid & 3) * 4;
    const bool a_thread = am < BM && (m0 + am) < p.Mw;
    const float* a_row = p.a + (size_t)grp * p.a_gs + (size_t)min(m0 + am, p.Mw - 1) * Kp;
    // B loader: pixel bk of the tile (lanes run along pixels), columns bn, bn + BSTEP, ...; a column = (channel, tap) is fixed per
    // thread for the whole kernel: its plane + tap offset and its tap displacement live in registers
    const int bk = tid & 15, bn = tid >> 4;
    const int plane = p.Hg * p.Wg;
    int col_off[BC], col_dy[BC], col_dx[BC];
#pragma unroll
    for (int j = 0; j < BC; j++) {
        const int nn = n0 + bn + BSTEP * j;
        const int c = min(nn / p.ntaps, p.Cg - 1), t = nn % p.ntaps;
        col_dy[j] = (nn < Nw) ? p.dy[t] : (1 << 28);               // out-of-range columns fail the bounds test below
        col_dx[j] = p.dx[t];
        col_off[j] = c * plane + p.dy[t] * p.Wg + p.dx[t];
    }
    // this thread's pixel of the NEXT tile to load, advanced by BK pixels per tile without a division or a loop:
    // BK = qstep * gw + rstep with rstep < gw, so one conditional wrap is exact for every grid width
    int kpix = kbeg + bk;
    int gy = kpix / p.gw, gx = kpix - gy * p.gw;
    const int qstep = BK / p.gw, rstep = BK - qstep * p.gw;

    f32x16 acc[WMB][WNB];
#pragma unroll
    for (int i = 0; i < WMB; i++)
#pragma unroll
        for (int j = 0; j < WNB; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    struct Stage {
        f32x4 ra;
        float rb[BC];
        uint32_t ok;        // bit j: rb[j] is a real sample; bit 31: ra is real
    };
    Stage S[2];
    Operands<WMB, WNB> O[2];
    auto gload = [&](int k0, Stage& st) {
        const int ka = k0 + aq;
        const bool a_ok = a_thread && ka < kend;      // AVEC: every 4-group lies entirely inside or outside [kbeg, kend)
        if constexpr (AVEC) {
            st.ra = *reinterpret_cast<const f32x4*>(a_row + (a_ok ? ka : 0));
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const bool in = a_ok && ka + q < kend;
                const float v = a_row[in ? ka + q : 0];
                st.ra[q] = in ? v : 0.f;
            }
        }
        uint32_t ok = a_ok ? 0x80000000u : 0u;
        const bool k_ok = kpix < kend;
        const int iy0 = gy * p.sy, ix0 = gx * p.sx;
        const int pixoff = iy0 * p.Wg + ix0;
#pragma unroll
        for (int j = 0; j < BC; j++) {
            const bool in = k_ok & ((unsigned)(iy0 + col_dy[j]) < (unsigned)p.Hg) & ((unsigned)(ix0 + col_dx[j]) < (unsigned)p.Wg);
            st.rb[j] = xin_g[in ? col_off[j] + pixoff : 0];
            ok |= in ? (1u << j) : 0u;
        }
        st.ok = ok;
        kpix += BK;
        gx += rstep;
        gy += qstep;
        const bool wrap = gx >= p.gw;
        gx -= wrap ? p.gw : 0;
        gy += wrap ? 1 : 0;
    };
    auto lstore = [&](int buf, const Stage& st) {
        float* As = As0 + buf * BM * LDK;
        float* Bs = Bs0 + buf * BN * LDK;
        if (am < BM) {
            f32x4 v = st.ra;
            if (!(st.ok >> 31)) v = f32x4{ 0.f, 0.f, 0.f, 0.f };
            *reinterpret_cast<f32x4*>(As + am * LDK + aq) = v;
        }
#pragma unroll
        for (int j = 0; j < BC; j++) Bs[(bn + BSTEP * j) * LDK + bk] = ((st.ok >> j) & 1u) ? st.rb[j] : 0.f;
    };
    auto lread = [&](int buf, Operands<WMB, WNB>& o) {
        read_operands<WMB, WNB>(As0 + buf * BM * LDK, Bs0 + buf * BN * LDK, wm, wn, lane, o);
    };

    // same three-stage pipeline as gather_conv_kernel
    const int nkt = (kend - kbeg + BK - 1) / BK;
    gload(kbeg, S[0]);
    if (nkt > 1) gload(kbeg + BK, S[1]);
    lstore(0, S[0]);
    lds_barrier();
    lread(0, O[0]);
    auto step_full = [&](int kt, Stage& s_same, Stage& s_next, Operands<WMB, WNB>& o_cur, Operands<WMB, WNB>& o_next) {
        gload(kbeg + (kt + 2) * BK, s_same);
        mma_half<WMB, WNB>(o_cur, 0, acc);
        lstore((kt + 1) & 1, s_next);
        lds_barrier();
        lread((kt + 1) & 1, o_next);
        mma_half<WMB, WNB>(o_cur, 1, acc);
    };
    auto step_tail = [&](int kt, Stage& s_next, Operands<WMB, WNB>& o_cur, Operands<WMB, WNB>& o_next, bool has_next) {
        mma_half<WMB, WNB>(o_cur, 0, acc);
        if (has_next) {
            lstore((kt + 1) & 1, s_next);
            lds_barrier();
            lread((kt + 1) & 1, o_next);
        }
        mma_half<WMB, WNB>(o_cur, 1, acc);
    };
    int kt = 0;
    for (; kt + 3 < nkt; kt += 2) {
        step_full(kt, S[0], S[1], O[0], O[1]);
        step_full(kt + 1, S[1], S[0], O[1], O[0]);
    }
    const int rem = nkt - kt;
    if (rem == 3) {
        step_full(kt, S[0], S[1], O[0], O[1]);
        step_tail(kt + 1, S[0], O[1], O[0], true);
        step_tail(kt + 2, S[1], O[0], O[1], false);
    } else if (rem == 2) {
        step_tail(kt, S[1], O[0], O[1], true);
        step_tail(kt + 1, S[0], O[1], O[0], false);
    } else {
        step_tail(kt, S[1], O[0], O[1], false);
    }

    wgrad_store<WMB, WNB>(p, acc, 1.f, grp, c_g, m0, n0, Nw, wm, wn, lane);
}

// wgrad_kernel on the split engine: both operands are activations, so both are split by their loader threads.  A: four consecutive
// pixels of a row (one 8-byte LDS write per plane); B: one pixel of BC columns -- neighbouring lanes hold neighbouring pixels, so
// lane pairs swap one value per column pair (DPP quad_perm) and every lane ends up with TWO adjacent pixels of one column of the pair:
// a 4-byte LDS write per plane and column pair instead of four 2-byte ones.
// BVEC (round 4): the gathered operand loaded along the K axis as well.  For a stride-1 "same" convolution whose row length is a multiple of
// 16, the 16 pixels of a K tile are one run of one image row, so a column (channel, tap) of the tile is 16 CONSECUTIVE input values (shifted by
// the tap): a thread takes 4 of them with one 16-byte load (unaligned 16-byte loads are free on this part: profiles/ub/load_rate.hip) instead
// of one value of 4 / 8 columns with as many 4-byte gathers, holds both values of every bf16 pair itself (no DPP exchange with the neighbour
// lane) and writes 8 bytes per plane.  Zero padding: a row outside the image masks the whole quad; a tap displacement of -1 / +1 reaches
// outside at the first / last quad of a row -- that quad is loaded UN-shifted and moved by one register with a zero at the open end, so
// every load stays inside the tensor and unconditional.  PMC of the scalar form (r04_pmc_wgrad_256_256_128.txt): 140 VALU instructions per wave
// and K tile for 12 MFMAs -- the weight gradient was bound by its loader's instruction count, not by the matrix pipe.
template <int WMB, int WNB, int WVM, int WVN, bool AVEC, int NTERMS, bool BVEC = false>
__global__ void __launch_bounds__(64 * WVM * WVN, AG_CONV_WAVES_PER_SIMD) wgrad_split_kernel(WgradProblem p)
{
    constexpr int NPL = planes_of(NTERMS);
    constexpr bool F16 = is_f16_form(NTERMS);
    using T = SplitTile<WMB, WNB, WVM, WVN, NPL>;
    constexpr int BM = T::BM, BN = T::BN, NT = T::NT;
    constexpr int BSTEP = NT / 16;
    constexpr int BC = BN / BSTEP;
    constexpr int BCV = BN / (NT / 4);                 // BVEC: columns per thread (a thread = 4 consecutive k of a column)
    static_assert(BM * 4 <= NT && BC % 2 == 0 && BCV >= 1, "loader shapes");
    __shared__ __attribute__((aligned(16))) char smem[T::lds_bytes];
    char* const As0 = smem;
    char* const Bs0 = smem + 2 * T::a_bytes;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WVN, wn = wave % WVN;
    const int Kp = p.gh * p.gw, Nw = p.Cg * p.ntaps;
    const int grp = (int)blockIdx.y / p.mtiles, my = (int)blockIdx.y - grp * p.mtiles;      // grouped launch (ag_groups.h)
    const float* __restrict__ xin_g = p.xin + (size_t)grp * p.xin_gs;
    float* __restrict__ c_g = p.c_t.p[0] ? const_cast<float*>(p.c_t.p[grp]) : p.c + (size_t)grp * p.c_gs;
    const int m0 = my * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * p.ksplit_len, kend = min(Kp, kbeg + p.ksplit_len);
    if (kbeg >= kend) return;

    const int am = tid >> 2, aq = (tid & 3) * 4;
    const bool a_thread = am < BM && (m0 + am) < p.Mw;
    const float* a_row = p.a + (size_t)grp * p.a_gs + (size_t)min(m0 + am, p.Mw - 1) * Kp;
    const int a_woff = chunk_off(min(am, BM - 1), aq >> 3) + (aq & 7) * 2;
    const int bk = tid & 15, bn = tid >> 4;
    const int plane = p.Hg * p.Wg;
    int col_off[BC], col_dy[BC], col_dx[BC];
#pragma unroll
    for (int j = 0; j < BC; j++) {
        const int nn = n0 + bn + BSTEP * j;
        const int c = min(nn / p.ntaps, p.Cg - 1), t = nn % p.ntaps;
        col_dy[j] = (nn < Nw) ? p.dy[t] : (1 << 28);
        col_dx[j] = p.dx[t];
        col_off[j] = c * plane + p.dy[t] * p.Wg + p.dx[t];
    }
    // after the pair swap an even lane writes pixels (bk, bk + 1) of the first column of each pair, an odd lane pixels (bk - 1, bk) of
    // the second: LDS word of row (bn + BSTEP * j), k = bk & ~1
    const int odd = bk & 1;
    int b_woff[BC / 2];
#pragma unroll
    for (int h = 0; h < BC / 2; h++) {
        const int row = bn + BSTEP * (2 * h + odd), k = bk & ~1;
        b_woff[h] = chunk_off(row, k >> 3) + (k & 7) * 2;
    }
    int kpix = kbeg + bk;
    int gy = kpix / p.gw, gx = kpix - gy * p.gw;
    const int qstep = BK / p.gw, rstep = BK - qstep * p.gw;
    // BVEC state: this thread's k quad, its columns' channel base / tap displacement, LDS word, and the tile's first pixel (row gy0, column gx0)
    const int vq = tid & 3, vn = tid >> 2;
    int vcol_base[BCV], vcol_dy[BCV], vcol_dx[BCV], vb_woff[BCV];
    if constexpr (BVEC) {
#pragma unroll
        for (int j = 0; j < BCV; j++) {
            const int nn = n0 + vn + (NT / 4) * j;
            const int c = min(nn / p.ntaps, p.Cg - 1), t = nn % p.ntaps;
            vcol_dy[j] = (nn < Nw) ? p.dy[t] : (1 << 28);
            vcol_dx[j] = p.dx[t];
            vcol_base[j] = c * plane;
            vb_woff[j] = chunk_off(vn + (NT / 4) * j, (4 * vq) >> 3) + ((4 * vq) & 7) * 2;
        }
    }
    int gy0 = kbeg / p.gw, gx0 = kbeg - gy0 * p.gw;

    f32x16 acc[WMB][WNB];
#pragma unroll
    for (int i = 0; i < WMB; i++)
#pragma unroll
        for (int j = 0; j < WNB; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    TensorScale sa{ 1.f, 1.f }, sb{ 1.f, 1.f };
    if constexpr (F16) {
        sa = tensor_scale(p.amax_a + (p.a_gs ? (size_t)grp * kAmaxParts : 0), 1.f, lane);
        sb = tensor_scale(p.amax_b + (p.xin_gs ? (size_t)grp * kAmaxParts : 0), 1.f, lane);
    }
    // one pair of fp32 values -> its words in the planes (w2 unused by the two-plane forms)
    auto split2 = [&](float x0, float x1, float sc, uint32_t& a, uint32_t& b, uint32_t& c) {
        b = 0; c = 0;
        if constexpr (NTERMS == kF16S) a = pack_f16(x0 * sc, x1 * sc);
        else if constexpr (F16)        split_pair_h(x0 * sc, x1 * sc, a, b);
        else                           split_pair(x0, x1, a, b, c);
    };

    struct Stage {
        f32x4 ra;
        float rb[BVEC ? 1 : BC];
        f32x4 rv[BVEC ? BCV : 1];
        uint32_t ok;           // bit 31: the A quad is real; scalar form: bit j = column j's value is real; BVEC: 2 bits per column (row ok, edge code)
    };
    Stage S[2];
    SplitOperands<WMB, WNB> O;
    auto gload = [&](int k0, Stage& st) {
        const int ka = k0 + aq;
        const bool a_ok = a_thread && ka < kend;
        if constexpr (AVEC) {
            st.ra = *reinterpret_cast<const f32x4*>(a_row + (a_ok ? ka : 0));
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const bool in = a_ok && ka + q < kend;
                const float v = a_row[in ? ka + q : 0];
                st.ra[q] = in ? v : 0.f;
            }
        }
        uint32_t ok = a_ok ? 0x80000000u : 0u;
        if constexpr (BVEC) {
            // whole tiles only (host: pixel count and slice length are multiples of 16): row gy0, columns gx0 .. gx0 + 15
            const int x0 = gx0 + 4 * vq;
#pragma unroll
            for (int j = 0; j < BCV; j++) {
                const int iy = gy0 + vcol_dy[j], ix = x0 + vcol_dx[j];
                const bool row_ok = (unsigned)iy < (unsigned)p.Hg;
                const int edge = ix < 0 ? 1 : (ix + 4 > p.Wg ? 2 : 0);           // the quad sticks out by one element on the left / right
                const int lx = ix + (edge == 1 ? 1 : 0) - (edge == 2 ? 1 : 0);
                st.rv[j] = *reinterpret_cast<const f32x4*>(xin_g + vcol_base[j] + (row_ok ? iy : gy0) * p.Wg + lx);
                ok |= ((row_ok ? 1u : 0u) | ((uint32_t)edge << 1)) << (3 * j);
            }
            st.ok = ok;
            gx0 += BK;
            const bool wrap0 = gx0 >= p.gw;
            gx0 -= wrap0 ? p.gw : 0;
            gy0 += wrap0 ? 1 : 0;
            return;
        }
        const bool k_ok = kpix < kend;
        const int iy0 = gy * p.sy, ix0 = gx * p.sx;
        const int pixoff = iy0 * p.Wg + ix0;
#pragma unroll
        for (int j = 0; j < BC; j++) {
            const bool in = k_ok & ((unsigned)(iy0 + col_dy[j]) < (unsigned)p.Hg) & ((unsigned)(ix0 + col_dx[j]) < (unsigned)p.Wg);
            st.rb[j] = xin_g[in ? col_off[j] + pixoff : 0];
            ok |= in ? (1u << j) : 0u;
        }
        st.ok = ok;
        kpix += BK;
        gx += rstep;
        gy += qstep;
        const bool wrap = gx >= p.gw;
        gx -= wrap ? p.gw : 0;
        gy += wrap ? 1 : 0;
    };
    auto lstore = [&](int buf, const Stage& st) {
        char* As = As0 + buf * T::a_bytes;
        char* Bs = Bs0 + buf * T::b_bytes;
        if (am < BM) {
            f32x4 v = st.ra;
            if (!(st.ok >> 31)) v = f32x4{ 0.f, 0.f, 0.f, 0.f };
            u32x2 w0, w1, w2;
#pragma unroll
            for (int e = 0; e < 2; e++) {
                uint32_t a, b, c;
                split2(v[2 * e], v[2 * e + 1], sa.s, a, b, c);
                w0[e] = a; w1[e] = b; w2[e] = c;
            }
            *reinterpret_cast<u32x2*>(As + 0 * BM * kRowB + a_woff) = w0;
            if constexpr (NPL >= 2) *reinterpret_cast<u32x2*>(As + 1 * BM * kRowB + a_woff) = w1;
            if constexpr (NPL == 3) *reinterpret_cast<u32x2*>(As + 2 * BM * kRowB + a_woff) = w2;
        }
        if constexpr (BVEC) {
#pragma unroll
            for (int j = 0; j < BCV; j++) {
                const uint32_t f = (st.ok >> (3 * j)) & 7u;
                const f32x4 L = st.rv[j];
                f32x4 v = L;
                if (f & 2u) v = f32x4{ 0.f, L[0], L[1], L[2] };           // left edge: loaded un-shifted, element -1 is padding
                if (f & 4u) v = f32x4{ L[1], L[2], L[3], 0.f };           // right edge
                if (!(f & 1u)) v = f32x4{ 0.f, 0.f, 0.f, 0.f };           // row outside the image (or a column past the last one)
                u32x2 w0, w1, w2;
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    uint32_t a, b, c;
                    split2(v[2 * e], v[2 * e + 1], sb.s, a, b, c);
                    w0[e] = a; w1[e] = b; w2[e] = c;
                }
                *reinterpret_cast<u32x2*>(Bs + 0 * BN * kRowB + vb_woff[j]) = w0;
                if constexpr (NPL >= 2) *reinterpret_cast<u32x2*>(Bs + 1 * BN * kRowB + vb_woff[j]) = w1;
                if constexpr (NPL == 3) *reinterpret_cast<u32x2*>(Bs + 2 * BN * kRowB + vb_woff[j]) = w2;
            }
            return;
        }
#pragma unroll
        for (int h = 0; h < BC / 2; h++) {
            const float v0 = ((st.ok >> (2 * h)) & 1u) ? st.rb[2 * h] : 0.f, v1 = ((st.ok >> (2 * h + 1)) & 1u) ? st.rb[2 * h + 1] : 0.f;
            // even lane gives away its second-column value and keeps the first; odd lane the other way round
            const float give = odd ? v0 : v1, keep = odd ? v1 : v0;
            const float got = __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(give), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
            uint32_t a, b, c;
            split2(odd ? got : keep, odd ? keep : got, sb.s, a, b, c);        // (pixel k & ~1, pixel (k & ~1) + 1)
            *reinterpret_cast<uint32_t*>(Bs + 0 * BN * kRowB + b_woff[h]) = a;
            if constexpr (NPL >= 2) *reinterpret_cast<uint32_t*>(Bs + 1 * BN * kRowB + b_woff[h]) = b;
            if constexpr (NPL == 3) *reinterpret_cast<uint32_t*>(Bs + 2 * BN * kRowB + b_woff[h]) = c;
        }
    };
    auto lread = [&](int buf) {
        read_split_operands<WMB, WNB, BM, BN, NTERMS>(As0 + buf * T::a_bytes, Bs0 + buf * T::b_bytes, wm, wn, lane, O);
    };

    auto mma = [&]() {
        if constexpr (F16) mma_split_h<WMB, WNB, NTERMS>(O, acc);
        else               mma_split<WMB, WNB, NTERMS>(O, acc);
    };

    const int nkt = (kend - kbeg + BK - 1) / BK;
    gload(kbeg, S[0]);
    if (nkt > 1) gload(kbeg + BK, S[1]);
    lstore(0, S[0]);
    lds_barrier();
    auto step_full = [&](int kt, Stage& s_same, Stage& s_next) {
        lread(kt & 1);
        gload(kbeg + (kt + 2) * BK, s_same);
        lstore((kt + 1) & 1, s_next);
        mma();
        lds_barrier();
    };
    auto step_tail = [&](int kt, Stage& s_next, bool has_next) {
        lread(kt & 1);
        if (has_next) lstore((kt + 1) & 1, s_next);
        mma();
        if (has_next) lds_barrier();
    };
    int kt = 0;
    for (; kt + 3 < nkt; kt += 2) {
        step_full(kt, S[0], S[1]);
        step_full(kt + 1, S[1], S[0]);
    }
    const int rem = nkt - kt;
    if (rem == 3) {
        step_full(kt, S[0], S[1]);
        step_tail(kt + 1, S[0], true);
        step_tail(kt + 2, S[1], false);
    } else if (rem == 2) {
        step_tail(kt, S[1], true);
        step_tail(kt + 1, S[0], false);
    } else {
        step_tail(kt, S[1], false);
    }

    wgrad_store<WMB, WNB>(p, acc, F16 ? sa.inv * sb.inv : 1.f, grp, c_g, m0, n0, Nw, wm, wn, lane);
}

// Matrix-pipe calibration: every wave issues `iters` x 4 independent v_mfma_f32_32x32x2_f32 back to back, no memory.
// Used by profiles/mfma_peak.py to measure the attainable fp32 MFMA rate of the box the convolutions are priced against.
__global__ void __launch_bounds__(256) mfma_rate_kernel(int iters, float* out)
{
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) v += acc[i][r];
    if (v == 12345.678f) out[0] = v;     // keep the chain alive
}

// The same for v_mfma_f32_32x32x16_bf16 (fp32 accumulation): the rate a three-term bf16 split of the fp32 operands
// (a_hi b_hi + a_hi b_lo + a_lo b_hi) would run on -- calibration only, nothing in the product uses it.
__global__ void __launch_bounds__(256) mfma_rate_bf16_kernel(int iters, float* out)
{
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)(1.0f + (threadIdx.x & 7) * 0.125f); b[i] = (__bf16)0.5f; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) v += acc[i][r];
    if (v == 12345.678f) out[0] = v;
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
static int round_up(int v, int a) { return (v + a - 1) / a * a; }

// arithmetic of the MFMA convolutions (ag_conv_set_math): process-wide, read at every call
static std::atomic<int> g_conv_math{ AG_CONV_MATH_SPLIT_F16 };
static int split_terms()        // 0: fp32 MFMA engine; 6 / 3: bf16 products per fp32 product; kF16 (2): the two-part fp16 form; kF16S (1): one fp16 part
{
    const int m = g_conv_math.load(std::memory_order_relaxed);
    return m == AG_CONV_MATH_SPLIT_BF16 ? 6 : m == AG_CONV_MATH_SPLIT_BF16X3 ? 3 : m == AG_CONV_MATH_SPLIT_F16 ? kF16 : m == AG_CONV_MATH_F16 ? kF16S : 0;
}
static bool split_math() { return split_terms() != 0; }

// fp16 form: the partial maxima of a call's two operands live behind the split-K partial sums in the call's workspace
constexpr size_t kAmaxBytes = 2 * (size_t)kMaxGroups * kAmaxParts * sizeof(float);
// one launch for the partial maxima of up to three operand tensors (instances: `table` entries, or `ptr + g * gs`, or ONE shared instance
// when gs == 0 and there is no table); tensor i's instance g goes to out[(i * kMaxGroups + g) * kAmaxParts ...]
int conv_absmax(const AmaxTensor* t, int n, int G, float* out, hipStream_t s, float* zero, int zero_inst)
{
    AmaxJobs J;
    int jobs = 0;
    static const bool trace = [] { const char* e = getenv("AG_AMAX_TRACE"); return e && e[0] == '1'; }();     // diagnostic: what is swept
    auto flush = [&]() {
        bool small = true;
        for (int j = 0; j < jobs; j++) small = small && J.len[j] * J.rows[j] <= 4096;
        if (trace) {
            long long tot = 0, mx = 0;
            for (int j = 0; j < jobs; j++) { const long long e = J.len[j] * J.rows[j]; tot += e; mx = e > mx ? e : mx; }
            fprintf(stderr, "AMAX jobs %d total %lld largest %lld rows %d %s\n", jobs, tot, mx, J.rows[0], small ? "small" : "full");
        }
        if (small) hipLaunchKernelGGL(absmax_small_kernel, dim3(jobs), dim3(256), 0, s, J);
        else {
            // workgroups per job: one pass of the four-loads-in-flight loop (16 K floats per workgroup) over the LARGEST job, at most
            // kAmaxParts -- 256 x jobs workgroups of 1024 threads for a stack of 36 K-float weight tensors were mostly dispatch
            long long largest = 0;
            for (int j = 0; j < jobs; j++) largest = std::max(largest, J.len[j] * J.rows[j]);
            static const bool sized = [] { const char* e = getenv("AG_AMAX_SIZED_GRID"); return !(e && e[0] == '0'); }();
            const int parts = sized ? (int)std::min<long long>(kAmaxParts, std::max<long long>(1, (largest + 16383) / 16384)) : kAmaxParts;
            hipLaunchKernelGGL(absmax_kernel, dim3(parts, jobs), dim3(kAmaxThreads), 0, s, J);
        }
        jobs = 0;
    };
    for (int i = 0; i < n; i++) {
        if (!t[i].ptr && !t[i].table) continue;
        const int n_i = t[i].inst ? t[i].inst : G;
        const int inst = t[i].table ? n_i : (t[i].gs ? n_i : 1);
        for (int g = 0; g < inst; g++) {
            if (jobs == 2 * kMaxGroups) flush();          // three full stacks of more than 10 instances: a second launch
            J.ptr[jobs] = t[i].table ? t[i].table->p[g] : t[i].ptr + (size_t)g * t[i].gs;
            J.len[jobs] = t[i].len; J.stride[jobs] = t[i].stride; J.rows[jobs] = t[i].rows;
            J.outp[jobs] = out + (size_t)(i * kMaxGroups + g) * kAmaxParts;
            jobs++;
        }
    }
    for (int g = 0; zero && g < zero_inst; g++) {          // slots to be zeroed for a producer's atomic maxima
        if (jobs == 2 * kMaxGroups) flush();
        J.ptr[jobs] = zero; J.len[jobs] = 0; J.stride[jobs] = 0; J.rows[jobs] = 1;
        J.outp[jobs] = zero + (size_t)g * kAmaxParts;
        jobs++;
    }
    if (jobs == 0) return AG_OK;
    flush();
    return check_hip(hipGetLastError(), "absmax_kernel");
}
size_t conv_absmax_floats(int tensors) { return (size_t)tensors * kMaxGroups * kAmaxParts; }
bool conv_math_needs_absmax() { return is_f16_form(split_terms()); }

// ag_conv_status: one pinned, mapped word per process (portable across devices); kernels store 1 into it, the host reads it without synchronising
static int* status_word()
{
    static int* w = [] {
        int* p = nullptr;
        if (hipHostMalloc(reinterpret_cast<void**>(&p), 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) return (int*)nullptr;
        *p = 0;
        return p;
    }();
    return w;
}
// The guard is armed in the SCALED fp16 forms only (round 6): there a non-finite accumulator means the scale was wrong and the outputs are
// garbage; in the fp32 / bf16 forms an inf or NaN operand simply propagates, as it does through the reference's cuDNN convolutions.
static int* armed_status_word() { return is_f16_form(split_terms()) ? status_word() : nullptr; }
static int status_tag_of(int kind, int rows, int chans) { return (kind << 28) | ((rows & 0x3fff) << 14) | (chans & 0x3fff); }
static int take_status(bool clear)
{
    int* w = status_word();
    if (!w) return AG_OK;
    const int v = __atomic_load_n(w, __ATOMIC_RELAXED);
    if (v && clear) __atomic_store_n(w, 0, __ATOMIC_RELAXED);
    if (!v) return AG_OK;
    static const char* kinds[] = {"?", "forward / input-gradient gather", "weight gradient", "?"};
    set_error("a convolution launched EARLIER (%s kernel, %d output rows x %d gathered channels; the last one to raise the flag) produced a non-finite "
              "accumulator: an operand exceeded the maximum its fp16 scale was taken from (a stale handed-over maximum?) or was not finite",
              kinds[(v >> 28) & 3], (v >> 14) & 0x3fff, v & 0x3fff);
    return AG_ERR_RANGE;
}

static int validate(const AgConvDesc* d)
{
    if (!d) { set_error("null conv descriptor"); return AG_ERR_INVALID_ARGUMENT; }
    if (d->Cin <= 0 || d->Cout <= 0 || d->H <= 0 || d->W <= 0 || d->k <= 0 || d->k * d->k > kMaxTaps) {
        set_error("bad conv sizes");
        return AG_ERR_INVALID_ARGUMENT;
    }
    if (d->kind == AG_CONV) {
        if ((d->stride != 1 && d->stride != 2) || d->padding < 0) { set_error("conv2d: stride must be 1 or 2, padding >= 0"); return AG_ERR_UNSUPPORTED; }
        if ((d->H + 2 * d->padding - d->k) < 0 || (d->W + 2 * d->padding - d->k) < 0) { set_error("conv2d: kernel larger than padded input"); return AG_ERR_INVALID_ARGUMENT; }
    } else if (d->kind == AG_CONV_TRANSPOSE) {
        if (d->stride != 2 || d->padding != 0) { set_error("conv_transpose2d: only stride 2, padding 0"); return AG_ERR_UNSUPPORTED; }
    } else {
        set_error("unknown conv kind");
        return AG_ERR_INVALID_ARGUMENT;
    }
    return AG_OK;
}

static void out_size(const AgConvDesc* d, int& OH, int& OW)
{
    if (d->kind == AG_CONV) {
        OH = (d->H + 2 * d->padding - d->k) / d->stride + 1;
        OW = (d->W + 2 * d->padding - d->k) / d->stride + 1;
    } else {
        OH = (d->H - 1) * 2 + d->k;
        OW = (d->W - 1) * 2 + d->k;
    }
}

// Tap subset of one class.  `taps` lists (ky, kx) of the subset.
struct TapSet { int n; int ky[kMaxTaps], kx[kMaxTaps]; };

// Tile height: 64 rows when that wastes fewer padded rows than 128 (the 64-channel layers at 512^2, the 12-channel ToRGB)
static int pick_bm(int M) { return round_up(M, 64) < round_up(M, 128) ? 64 : 128; }
static int bn_of(int bm) { return bm == 64 ? 256 : 128; }

static float wscale_of(const AgConvDesc* d) { return d->weight_scale == 0.f ? 1.f : d->weight_scale; }

// Split-K policy.  A 128 x 128 tile per workgroup leaves the chip idle when M * N is small (the 512-channel layers at
// 8^2 .. 64^2 have 4 .. 128 tiles for 256 CUs) and the K loop (up to 576 tiles) becomes the critical path, so slices of K go to
// blockIdx.z.  Model fitted to profiles/conv_split_sweep.py (r2_split_sweep*.log): a CU works through the workgroups it is dealt;
// two co-resident ones finish in ~1.8x the time of one, so with W workgroups a CU's share is r = ceil(W / 256) of them and the
// launch takes  (K tiles per slice) * (r == 1 ? 1 : 0.9 r)  + per-workgroup fixed cost * r  (+ the partial-sum pass, growing with
// the split count).  It lands on W = 256 or 512 instead of 384 (half the CUs with two workgroups, the rest with one: +20 %).
// (Dealing the slices to XCDs -- all tiles of a slice on one L2 -- was measured too: no gain, r2_split_sweep2.log.)
constexpr size_t kMaxPartialBytes = size_t(96) << 20;
constexpr int kCUs = 256;
static double lanes_cost(long long W, int per, double fixed)
{
    const long long r = (W + kCUs - 1) / kCUs;
    return per * (r <= 1 ? 1.0 : 0.9 * r) + fixed * r;
}
static int choose_splits(long long tiles, int Mpad, int Ncols, int nkt, int G = 1)
{
    if (nkt < 8) return 1;
    const long long cap = (long long)(kMaxPartialBytes / ((size_t)Mpad * Ncols * sizeof(float) * G));
    const int smax = (int)std::min<long long>(std::min<long long>(nkt / 4, cap), 96);
    // unit = the time of one K tile in a workgroup (~0.45 us).  Splitting adds the finish launch (~6 units) and, per split, one
    // write + one read of the fp32 output at ~6 TB/s (L2 / Infinity Cache resident for these sizes)
    const double per_split = 8.0 * (double)Mpad * Ncols * G / 6e12 / 0.45e-6;
    int best = 1;
    double best_cost = 1e30;
    for (int sp = 1; sp <= std::max(1, smax); sp++) {
        const int per = (nkt + sp - 1) / sp;
        const double cost = lanes_cost(tiles * sp, per, 10.0) + (sp > 1 ? 6.0 + per_split * sp : 0.0);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = sp; }
    }
    return best;
}

// Round 6 EXPERIMENT (opt-in, AG_CONV_DMA=1: big tiles only, =2: every eligible launch): the LDS-DMA loader over pre-split planes
// (gather_conv_dma_kernel) for the fp16 forms on whole 16-channel blocks.  Correct (tests/test_conv_gpu.py green in this mode) and NOT faster on the
// training step: profiles/r06_conv_dma.md.
static bool conv_dma_possible(int Cg)
{
    static const bool dma_on = [] { const char* e = getenv("AG_CONV_DMA"); return e && (e[0] == '1' || e[0] == '2'); }();      // OPT-IN: measured no net gain (below)
    return dma_on && is_f16_form(split_terms()) && Cg % BK == 0;
}
// Workgroup tile.  A 128 x 128 tile moves 16 KB of operands per 16-deep K tile from L2 for 1.57 M products: 96 products per byte, i.e. ~26 TB/s of
// L2 -> LDS traffic at the matrix pipe's peak -- the 128-wide kernels are L2-bandwidth-bound at about half of it (profiles/r06_conv_dma.md).  With the
// DMA loader's registers free for accumulators: 256 x 256 (192 products per byte) where the output has whole 256-row tiles and enough of them to
// fill the chip, 128 x 256 (131) for the 128-channel layers; AG_CONV_BIG_TILES=0 keeps the 128-wide tiles.
static void pick_tile(int M, long long N, int G, bool dma, int& bm, int& bn)
{
    bm = pick_bm(M); bn = bn_of(bm);
    static const bool big = [] { const char* e = getenv("AG_CONV_BIG_TILES"); return !(e && e[0] == '0'); }();
    if (!dma || !big) return;
    const int terms = split_terms();
    const long long nt256 = (N + 255) / 256;
    if (M % 256 == 0 && nt256 * (M / 256) * G >= 192) { bm = 256; bn = 256; }
    else if (M % 128 == 0 && terms == kF16 && nt256 * (M / 128) * G >= 384) { bm = 128; bn = 256; }
}

// Fills tile_begin / col_begin / at_off / nkt of the classes (dy, dx, gh, gw, y0, x0, ntaps set by the caller), packs the
// weights of all classes with one launch and runs them with one launch (+ one split-K finish).
static int pack_and_launch(GatherProblem& gp, const TapSet* taps, int bm, const PtrTable& w, long long stride_c, long long stride_m,
                           float wscale, int k, float* At, float* partial, float* amax, const AmaxTensor& w_shape, const float* pre_w,
                           const float* pre_in, hipStream_t s, bool skip_pack = false, char* xs_buf = nullptr, int bn = 0)
{
    const bool split = split_math();
    const int terms = split_terms();
    const bool f16 = is_f16_form(terms);
    const int BN = bn ? bn : bn_of(bm);
    const int G = gp.G;
    PackProblem pp;
    pp.w_t = w; pp.At = At; pp.C = gp.Cg; pp.Cpad = gp.Cpad; pp.M = gp.M; pp.Mpad = gp.Mpad; pp.BM = bm; pp.nclasses = gp.nclasses;
    pp.stride_c = stride_c; pp.stride_m = stride_m; pp.wscale = wscale; pp.has_wscale = wscale != 1.f;
    long long tiles = 0, at = 0;
    int cols = 0, nkt_max = 0;
    for (int c = 0; c < gp.nclasses; c++) {
        GatherClass& cl = gp.cls[c];
        cl.nkt = cl.ntaps * (gp.Cpad / BK);
        cl.tile_begin = (int)tiles; cl.col_begin = cols; cl.at_off = at; cl.pad_ = 0;
        const int N = cl.gh * cl.gw;
        tiles += (N + BN - 1) / BN;
        cols += N;
        pp.begin[c] = at; pp.ntaps[c] = cl.ntaps; pp.zero[c] = cl.zero_weights;
        for (int t = 0; t < cl.ntaps; t++) pp.tapoff[c][t] = taps[c].ky[t] * k + taps[c].kx[t];
        at += (long long)cl.nkt * BK * gp.Mpad;
        if (cl.nkt > nkt_max) nkt_max = cl.nkt;
    }
    pp.begin[gp.nclasses] = at;
    pp.at_gs = at;
    gp.at_gs = at;
    gp.mtiles = gp.Mpad / bm;
    gp.Ncols = cols;
    if (tiles == 0) return AG_OK;
    pp.k2 = k * k;
    if ((stride_c != pp.k2 && stride_m != pp.k2) || pp.k2 > kMaxTaps) { set_error("pack: unexpected weight strides"); return AG_ERR_INVALID_ARGUMENT; }
    pp.planes = planes_of(terms);
    pp.amax = nullptr;
    gp.amax_a = gp.amax_b = nullptr;
    gp.amax_a_mult = 1.f;
    if (f16 && skip_pack && !pre_w) { set_error("conv: frozen packed weights need their maxima (ConvOpts::amax_w)"); return AG_ERR_INVALID_ARGUMENT; }
    if (f16) {
        AmaxTensor t[2] = { w_shape, AmaxTensor{ gp.xin, nullptr, gp.x_gs, (long long)gp.Cg * gp.Hg * gp.Wg, 0, 1 } };
        t[0].table = &w;
        if (pre_w) t[0] = AmaxTensor{};           // the caller already has this operand's partial maxima (ConvOpts)
        if (pre_in) t[1] = AmaxTensor{};
        int rc0 = conv_absmax(t, 2, G, amax, s);
        if (rc0) return rc0;
        pp.amax = pre_w ? pre_w : amax;
        gp.amax_a = pp.amax;
        gp.amax_b = pre_in ? pre_in : amax + (size_t)kMaxGroups * kAmaxParts;
        gp.amax_a_mult = wscale != 1.f ? fabsf(wscale) : 1.f;
    }
    int rc = AG_OK;
    if (!skip_pack) {                             // (frozen weights: the image a previous call left in ConvOpts::packed)
        if (split) hipLaunchKernelGGL(pack_weights_split_kernel, dim3(gp.Cpad / BK, gp.Mpad / 16, G), dim3(256), 0, s, pp);
        else       hipLaunchKernelGGL(pack_weights_kernel, dim3(gp.Cpad / BK, gp.Mpad / 16, G), dim3(256), 0, s, pp);
        rc = check_hip(hipGetLastError(), "pack_weights_kernel");
        if (rc) return rc;
    }

    int splits = choose_splits(tiles * (gp.Mpad / bm) * G, gp.Mpad, cols, nkt_max, G);
    if (const char* forced = getenv("AG_CONV_SPLITS")) {       // measurement hook (profiles/conv_split_sweep.py)
        const int f = atoi(forced);
        const long long cap = (long long)(kMaxPartialBytes / ((size_t)gp.Mpad * cols * sizeof(float) * G));
        if (f >= 1) splits = (int)std::min<long long>(std::min(f, std::max(1, nkt_max)), std::max<long long>(1, cap));
    }
    gp.kt_per_split = (nkt_max + splits - 1) / splits;
    splits = (nkt_max + gp.kt_per_split - 1) / gp.kt_per_split;
    gp.partial = splits > 1 ? partial : nullptr;
    gp.part_gs = (long long)splits * gp.Mpad * cols;
    gp.At = At;
    dim3 grid((unsigned)tiles, (gp.Mpad / bm) * G, splits);
    // XCD-aware tile order (tile_of_workgroup), opt-in with AG_CONV_XCD=1: 4.2x less L2-miss traffic (PMC: 524 -> 124 MB per launch on a
    // 256 -> 256 layer at 256^2) and no time gained (ABBA same-box: 33.37 vs 33.45 ms per step) -- the kernel does not wait for those misses
    // (profiles/r04_conv_xcd_order_ab.txt).  Needs whole eighths.
    static const bool xcd_on = [] { const char* e = getenv("AG_CONV_XCD"); return e && e[0] == '1'; }();
    gp.xcd_order = (xcd_on && ((long long)grid.x * grid.y) % 8 == 0 && (long long)grid.x * grid.y >= 64) ? 1 : 0;
    double flops = 0.0;
    for (int c = 0; c < gp.nclasses; c++)
        if (!gp.cls[c].zero_weights) flops += 2.0 * G * gp.M * (double)gp.cls[c].gh * gp.cls[c].gw * gp.cls[c].ntaps * gp.Cg;
    char tag[64];
    snprintf(tag, sizeof(tag), "g bm%d G%d M%d C%d N%d cls%d nkt%d sp%d wg%lld", bm, G, gp.M, gp.Cg, cols, gp.nclasses, nkt_max, splits,
             (long long)grid.x * grid.y * grid.z);
    // Round 6: the LDS-DMA loader over pre-split planes (gather_conv_dma_kernel) for the fp16 forms on whole 16-channel blocks; AG_CONV_DMA=0: the
    // register-staged loader (same-box A/B)
    // the DMA loader pays for its pre-split pass (8 bytes per input element at ~5 TB/s) only where it buys the big tiles: on 128-wide tiles it is
    // 2-4 % faster than the register-staged loader before that pass, 5-10 % slower after it (profiles/r06_conv_dma.md); AG_CONV_DMA=2 forces it everywhere
    static const bool dma_all = [] { const char* e = getenv("AG_CONV_DMA"); return e && e[0] == '2'; }();
    const bool dma = conv_dma_possible(gp.Cg) && xs_buf && !(terms == kF16S && bm == 64) && (dma_all || BN * bm >= 128 * 256);
    if (dma) {
        PresplitProblem ps_;
        const int inst = gp.x_gs ? G : 1;
        ps_.x = gp.xin; ps_.xs = xs_buf; ps_.amax = gp.amax_b; ps_.x_gs = gp.x_gs;
        ps_.C = gp.Cg; ps_.Cpad = gp.Cpad; ps_.HW = gp.Hg * gp.Wg; ps_.planes = planes_of(terms);
        ps_.xs_gs = (long long)ps_.planes * (gp.Cpad / 8) * ps_.HW * 16;
        const int bx = std::max(1, std::min((ps_.HW + 255) / 256, 2048 / std::max(1, (gp.Cpad / 8) * inst / 8)));
        hipLaunchKernelGGL(presplit_kernel, dim3(bx, gp.Cpad / 8, inst), dim3(256), 0, s, ps_);
        rc = check_hip(hipGetLastError(), "presplit_kernel");
        if (rc) return rc;
        ProfScope ps(AG_K_GATHER_CONV, s, flops, tag);
#define AG_LAUNCH_DMA(WMB, WNB, WVM, WVN, NTM) hipLaunchKernelGGL((gather_conv_dma_kernel<WMB, WNB, WVM, WVN, NTM>), grid, dim3(64 * WVM * WVN), 0, s, gp, (const char*)xs_buf, ps_.xs_gs)
        if (bm == 256 && BN == 256) { if (terms == kF16S) AG_LAUNCH_DMA(4, 2, 2, 4, kF16S); else AG_LAUNCH_DMA(4, 2, 2, 4, kF16); }
        else if (bm == 128 && BN == 256) AG_LAUNCH_DMA(2, 2, 2, 4, kF16);
        else if (bm == 64) AG_LAUNCH_DMA(2, 2, 1, 4, kF16);
        else if (terms == kF16S) AG_LAUNCH_DMA(2, 2, 2, 2, kF16S);
        else AG_LAUNCH_DMA(2, 2, 2, 2, kF16);
#undef AG_LAUNCH_DMA
        rc = check_hip(hipGetLastError(), "gather_conv_dma_kernel");
        if (rc || splits == 1) return rc;
        const long long total = (long long)gp.M * cols;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        if (gp.act && gp.out_amax && blocks * G > 2048) blocks = std::max(1, 2048 / G);
        hipLaunchKernelGGL(reduce_splits_kernel, dim3(blocks, G), dim3(256), 0, s, gp, splits);
        return check_hip(hipGetLastError(), "reduce_splits_kernel");
    }
    ProfScope ps(AG_K_GATHER_CONV, s, flops, tag);      // covers the split-K finish too
    if (split) {
        const bool cexact = gp.Cg % BK == 0 && (size_t)gp.Cg * gp.Hg * gp.Wg * sizeof(float) < (size_t(1) << 32);
#define AG_LAUNCH_SPLIT_T(WMB, WNB, CE, NTM) hipLaunchKernelGGL((gather_conv_split_kernel<WMB, WNB, 2, 4, CE, NTM>), grid, dim3(512), 0, s, gp)
#define AG_LAUNCH_SPLIT(WMB, WNB, CE) do { if (terms == 6) AG_LAUNCH_SPLIT_T(WMB, WNB, CE, 6); else if (terms == 3) AG_LAUNCH_SPLIT_T(WMB, WNB, CE, 3); \
                                           else if (terms == kF16S) AG_LAUNCH_SPLIT_T(WMB, WNB, CE, kF16S); else AG_LAUNCH_SPLIT_T(WMB, WNB, CE, kF16); } while (0)
        if (bm == 64) {
            if (cexact) AG_LAUNCH_SPLIT(1, 2, true); else AG_LAUNCH_SPLIT(1, 2, false);
        } else {
            if (cexact) AG_LAUNCH_SPLIT(2, 1, true); else AG_LAUNCH_SPLIT(2, 1, false);
        }
#undef AG_LAUNCH_SPLIT
#undef AG_LAUNCH_SPLIT_T
    } else {
        if (bm == 64) hipLaunchKernelGGL((gather_conv_kernel<1, 2, 2, 4>), grid, dim3(512), 0, s, gp);
        else          hipLaunchKernelGGL((gather_conv_kernel<2, 1, 2, 4>), grid, dim3(512), 0, s, gp);
    }
    rc = check_hip(hipGetLastError(), "gather_conv_kernel");
    if (rc || splits == 1) return rc;
    const long long total = (long long)gp.M * cols;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (gp.act && gp.out_amax && blocks * G > 2048) blocks = std::max(1, 2048 / G);      // each workgroup ends with an atomic into its instance's 16 lines of slots
    hipLaunchKernelGGL(reduce_splits_kernel, dim3(blocks, G), dim3(256), 0, s, gp, splits);
    return check_hip(hipGetLastError(), "reduce_splits_kernel");
}

}  // namespace ag

using namespace ag;

extern "C" {

int ag_conv_output_size(const AgConvDesc* d, int32_t* OH, int32_t* OW)
{
    int rc = validate(d);
    if (rc) return rc;
    int oh, ow;
    out_size(d, oh, ow);
    if (OH) *OH = oh;
    if (OW) *OW = ow;
    return AG_OK;
}

static size_t packed_bytes(const AgConvDesc* d)
{
    const int Cmax = d->Cin > d->Cout ? d->Cin : d->Cout;
    const size_t kk = (size_t)round_up(Cmax, BK) * (d->k * d->k + 4);   // all tap subsets together (+ degenerate classes)
    return align_up(kk * (size_t)round_up(Cmax, 128) * 6, 256);      // 6 bytes per element: three bf16 planes (fp32 engine: 4)
}

size_t ag_conv_workspace_bytes(const AgConvDesc* d)
{
    if (validate(d)) return 0;
    // packed weights of all tap subsets together + split-K partial sums (choose_splits keeps them under the cap)
    return conv_workspace_bytes_g(d, 1);
}

}  // extern "C"

namespace ag {
// the pre-split activation planes of the DMA loader (round 6): 4 bytes per element of the larger of the two tensors a gather launch may read
static size_t presplit_bytes(const AgConvDesc* d, int G)
{
    int OH, OW;
    out_size(d, OH, OW);
    const size_t a = (size_t)round_up(d->Cin, BK) * d->H * d->W, b = (size_t)round_up(d->Cout, BK) * OH * OW;
    return align_up((size_t)G * std::max(a, b) * 4, 256) + 256;
}
size_t conv_workspace_bytes_g(const AgConvDesc* d, int G)
{
    if (validate(d) || G < 1 || G > kMaxGroups) return 0;
    static const bool dma_on = [] { const char* e = getenv("AG_CONV_DMA"); return e && (e[0] == '1' || e[0] == '2'); }();
    return (size_t)G * packed_bytes(d) + kMaxPartialBytes + kAmaxBytes + 512 + (dma_on ? presplit_bytes(d, G) : 0);
}
size_t conv_packed_bytes_g(const AgConvDesc* d, int G)
{
    if (validate(d) || G < 1 || G > kMaxGroups) return 0;
    return (size_t)G * packed_bytes(d) + 512;          // (+ the alignment slack of aligned_base)
}
}  // namespace ag

extern "C" {

// Shared by forward / backward-input: which GEMM to run.
//   plain gather with all taps (sy = sx = stride_in), tap offset = sign * k - pad_off, one class
//   or the 4 output-parity classes of a stride-2 "scatter" (transposed conv forward, input gradient of a stride-2 conv)
static int run_gather_family(const AgConvDesc* d, bool backward_input, int G, const float* xin, long long x_gs, const PtrTable& w,
                             const float* out_scale, const PtrTable& bias, float* yout, long long y_gs, void* workspace, size_t workspace_bytes,
                             hipStream_t s, const ConvOpts& opt = ConvOpts())
{
    int OH, OW;
    out_size(d, OH, OW);
    const int k = d->k, k2 = k * k;
    if (workspace_bytes < conv_workspace_bytes_g(d, G) || !workspace) { set_error("conv workspace too small"); return AG_ERR_SCRATCH_TOO_SMALL; }
    if (G > 1 && out_scale) { set_error("grouped convolution: out_scale is a single-instance option"); return AG_ERR_UNSUPPORTED; }
    float* At = opt.packed ? reinterpret_cast<float*>(aligned_base(opt.packed)) : reinterpret_cast<float*>(aligned_base(workspace));
    float* partial = reinterpret_cast<float*>(aligned_base(workspace) + (size_t)G * packed_bytes(d));
    float* amax = reinterpret_cast<float*>(aligned_base(workspace) + (size_t)G * packed_bytes(d) + kMaxPartialBytes);
    char* xs_buf = aligned_base(workspace) + align_up((size_t)G * packed_bytes(d) + kMaxPartialBytes + kAmaxBytes, 256);

    GatherProblem gp;
    gp.status = armed_status_word();
    gp.out_scale = out_scale; gp.bias_t = bias; gp.yout = yout; gp.xin = xin;
    gp.G = G; gp.x_gs = x_gs; gp.y_gs = y_gs;
    gp.act = opt.act ? opt.act->kind : 0;
    gp.slope = opt.act ? opt.act->slope : 0.f; gp.act_scale = opt.act ? opt.act->scale : 1.f;
    gp.noise_t = opt.act ? opt.act->noise : PtrTable{}; gp.nw_t = opt.act ? opt.act->nw : PtrTable{};
    gp.out_amax = opt.act ? opt.act->out_amax : nullptr;
    if (gp.act && (out_scale || backward_input)) { set_error("conv: the fused activation is a forward option without out_scale"); return AG_ERR_INVALID_ARGUMENT; }
    const bool conv = d->kind == AG_CONV;
    // input of the GEMM / output of the GEMM in tensor terms
    int Cg, Hg, Wg, M, OHf, OWf;
    long long stride_c, stride_m;
    if (!backward_input) { Cg = d->Cin; Hg = d->H; Wg = d->W; M = d->Cout; OHf = OH; OWf = OW; }
    else                 { Cg = d->Cout; Hg = OH; Wg = OW; M = d->Cin; OHf = d->H; OWf = d->W; }
    // weight layout: conv [Cout][Cin][k][k], transposed conv [Cin][Cout][k][k]
    // wt_oihw: a transposed convolution whose weight is kept [Cout][Cin][k][k] like a convolution's (the layer calls: the modulated weight
    // and its gradient then have ONE layout and the modulation kernels read and write them coalesced)
    const bool oihw = conv || opt.wt_oihw;
    if (opt.w_cin_total && (!conv || opt.w_cin_total < d->Cin)) { set_error("conv: a weight slice needs AG_CONV and w_cin_total >= Cin"); return AG_ERR_INVALID_ARGUMENT; }
    const long long cin_rows = opt.w_cin_total ? opt.w_cin_total : d->Cin;       // channels of the weight tensor the rows belong to
    const long long s_co = oihw ? cin_rows * k2 : k2, s_ci = oihw ? k2 : (long long)d->Cout * k2;
    stride_c = backward_input ? s_co : s_ci;
    stride_m = backward_input ? s_ci : s_co;
    // the weight tensor of an instance as runs of floats (absmax of the fp16 form): whole and contiguous, or Cout rows of a channel slice
    AmaxTensor w_shape{ nullptr, nullptr, 0, (long long)d->Cout * d->Cin * k2, 0, 1 };
    const float* const pre_in = backward_input ? opt.amax_dy : opt.amax_x;
    if (opt.w_cin_total && opt.w_cin_total != d->Cin) { w_shape.len = (long long)d->Cin * k2; w_shape.stride = cin_rows * k2; w_shape.rows = d->Cout; }
    gp.status_tag = status_tag_of(1, M, Cg);
    const bool scatter_ = (d->kind == AG_CONV && backward_input && d->stride == 2) || (d->kind != AG_CONV && !backward_input);
    int bm, bn;
    pick_tile(M, (long long)OHf * OWf / (scatter_ ? 4 : 1), G, conv_dma_possible(Cg), bm, bn);
    gp.Cg = Cg; gp.Hg = Hg; gp.Wg = Wg; gp.M = M; gp.Mpad = round_up(M, bm); gp.OHf = OHf; gp.OWf = OWf;
    gp.Cpad = round_up(Cg, BK);
    TapSet taps[kMaxClasses];

    // "gather" cases: forward conv, input gradient of a transposed conv (a stride-2 conv over dy), input gradient of a
    // stride-1 conv (taps mirrored).  "scatter" cases (stride 2): transposed conv forward, input gradient of a stride-2 conv.
    const bool scatter = (conv && backward_input && d->stride == 2) || (!conv && !backward_input);
    if (gp.act && scatter) { set_error("conv: the fused activation needs a plain gather (no transposed convolution)"); return AG_ERR_UNSUPPORTED; }
    if (!scatter) {
        TapSet& ts = taps[0]; ts.n = k2;
        for (int ky = 0; ky < k; ky++) for (int kx = 0; kx < k; kx++) { ts.ky[ky * k + kx] = ky; ts.kx[ky * k + kx] = kx; }
        GatherClass& cl = gp.cls[0];
        gp.nclasses = 1; gp.os = 1;
        cl.ntaps = k2; cl.gh = OHf; cl.gw = OWf; cl.y0 = 0; cl.x0 = 0; cl.zero_weights = 0;
        if (conv && !backward_input) {                 // y[oy] <- x[oy*s - p + ky]
            gp.sy = gp.sx = d->stride;
            for (int t = 0; t < k2; t++) { cl.dy[t] = ts.ky[t] - d->padding; cl.dx[t] = ts.kx[t] - d->padding; }
        } else if (conv) {                             // stride-1 conv, dx[iy] <- dy[iy + p - ky]
            gp.sy = gp.sx = 1;
            for (int t = 0; t < k2; t++) { cl.dy[t] = d->padding - ts.ky[t]; cl.dx[t] = d->padding - ts.kx[t]; }
        } else {                                       // transposed conv, dx[iy] <- dy[2 iy + ky]
            gp.sy = gp.sx = 2;
            for (int t = 0; t < k2; t++) { cl.dy[t] = ts.ky[t]; cl.dx[t] = ts.kx[t]; }
        }
        for (int t = k2; t < kMaxTaps; t++) cl.dy[t] = cl.dx[t] = 0;
        return pack_and_launch(gp, taps, bm, w, stride_c, stride_m, wscale_of(d), k, At, partial, amax, w_shape, opt.amax_w, pre_in, s, opt.packed && opt.packed_valid, xs_buf, bn);
    }
    // scatter with stride 2: output coordinate o = 2*i + ky - poff  (poff = padding for the conv gradient, 0 for convT).
    // Class (qy, qx) = parity of the output coordinate; it receives only taps with ky = (o + poff) mod 2, from
    // input i = (o + poff - ky) / 2 = g + (q + poff - ky) / 2 with o = q + 2 g.
    const int poff = conv ? d->padding : 0;
    gp.nclasses = 0; gp.os = 2; gp.sy = gp.sx = 1;
    for (int qy = 0; qy < 2; qy++)
        for (int qx = 0; qx < 2; qx++) {
            TapSet ts; ts.n = 0;
            for (int ky = 0; ky < k; ky++) {
                if (((qy + poff - ky) & 1) != 0) continue;
                for (int kx = 0; kx < k; kx++) {
                    if (((qx + poff - kx) & 1) != 0) continue;
                    ts.ky[ts.n] = ky; ts.kx[ts.n] = kx; ts.n++;
                }
            }
            GatherClass cl;
            cl.gh = (OHf - qy + 1) / 2; cl.gw = (OWf - qx + 1) / 2;
            cl.y0 = qy; cl.x0 = qx; cl.zero_weights = 0;
            if (cl.gh <= 0 || cl.gw <= 0) continue;
            if (ts.n == 0) {
                // no tap reaches this class (possible for k = 1): the outputs are out_scale*0 + bias; handled as a class over
                // one tap with all-zero packed weights
                ts.n = 1; ts.ky[0] = 0; ts.kx[0] = 0;
                cl.zero_weights = 1;
            }
            cl.ntaps = ts.n;
            for (int t = 0; t < kMaxTaps; t++) { cl.dy[t] = 0; cl.dx[t] = 0; }
            // floor division for negative odd numerators never happens: numerators are even by construction
            for (int t = 0; t < ts.n; t++) { cl.dy[t] = (qy + poff - ts.ky[t]) / 2; cl.dx[t] = (qx + poff - ts.kx[t]) / 2; }
            if (cl.zero_weights) { cl.dy[0] = 0; cl.dx[0] = 0; }
            // heaviest class first: insertion by descending tap count
            int pos = gp.nclasses;
            while (pos > 0 && gp.cls[pos - 1].ntaps < cl.ntaps) { gp.cls[pos] = gp.cls[pos - 1]; taps[pos] = taps[pos - 1]; pos--; }
            gp.cls[pos] = cl; taps[pos] = ts;
            gp.nclasses++;
        }
    return pack_and_launch(gp, taps, bm, w, stride_c, stride_m, wscale_of(d), k, At, partial, amax, w_shape, opt.amax_w, pre_in, s, opt.packed && opt.packed_valid, xs_buf, bn);
}

}  // extern "C"

namespace ag {

int conv_forward_g(const AgConvDesc* d, int G, const float* x, long long x_gs, const PtrTable& w, const float* out_scale, const PtrTable& bias,
                   float* y, long long y_gs, void* workspace, size_t workspace_bytes, hipStream_t s, const ConvOpts& o)
{
    int rc = validate(d);
    if (rc || (rc = take_status(true))) return rc;
    if (G < 1 || G > kMaxGroups || !x || !table_complete(w, G) || !y) { set_error("null conv tensor / bad group count"); return AG_ERR_INVALID_ARGUMENT; }
    if (!o.w_cin_total && !o.act && (rc = pointwise_forward(d, G, x, x_gs, w, out_scale, bias, y, y_gs, s)) != 0) return rc < 0 ? rc : AG_OK;
    return run_gather_family(d, false, G, x, x_gs, w, out_scale, bias, y, y_gs, workspace, workspace_bytes, s, o);
}

int conv_backward_input_g(const AgConvDesc* d, int G, const float* dy, long long dy_gs, const PtrTable& w, float* dx, long long dx_gs,
                          void* workspace, size_t workspace_bytes, hipStream_t s, const ConvOpts& o)
{
    int rc = validate(d);
    if (rc || (rc = take_status(true))) return rc;
    if (G < 1 || G > kMaxGroups || !dy || !table_complete(w, G) || !dx) { set_error("null conv tensor / bad group count"); return AG_ERR_INVALID_ARGUMENT; }
    if (!o.w_cin_total && (rc = pointwise_backward_input(d, G, dy, dy_gs, w, dx, dx_gs, s)) != 0) return rc < 0 ? rc : AG_OK;
    return run_gather_family(d, true, G, dy, dy_gs, w, nullptr, PtrTable{}, dx, dx_gs, workspace, workspace_bytes, s, o);
}

// dw: G gradients stacked at dw_gs floats (each the shape of one weight); overwritten
int conv_backward_weight_g(const AgConvDesc* d, int G, const float* x, long long x_gs, const float* dy, long long dy_gs, float* dw, long long dw_gs,
                           void* workspace, size_t workspace_bytes, hipStream_t s, const ConvOpts& o)
{
    int rc = validate(d);
    if (rc || (rc = take_status(true))) return rc;
    if (G < 1 || G > kMaxGroups || !x || !dy || (!dw && !o.dw_table)) { set_error("null conv tensor / bad group count"); return AG_ERR_INVALID_ARGUMENT; }
    if (o.dw_table && (!table_complete(*o.dw_table, G) || o.dw_row_stride <= 0 || d->k == 1)) { set_error("conv: bad weight-gradient table"); return AG_ERR_INVALID_ARGUMENT; }
    if (!o.dw_table && (rc = pointwise_backward_weight(d, G, x, x_gs, dy, dy_gs, dw, dw_gs, workspace, workspace_bytes, s)) != 0) return rc < 0 ? rc : AG_OK;
    int OH, OW;
    out_size(d, OH, OW);
    const int k = d->k, k2 = k * k;
    WgradProblem wp;
    if (d->kind == AG_CONV) {      // dw[co][(ci,t)] = sum_o dy[co][o] * x[ci][o*s - p + k]
        wp.a = dy; wp.xin = x; wp.a_gs = dy_gs; wp.xin_gs = x_gs; wp.Mw = d->Cout; wp.Cg = d->Cin; wp.Hg = d->H; wp.Wg = d->W; wp.gh = OH; wp.gw = OW;
        wp.sy = wp.sx = d->stride;
        for (int t = 0; t < k2; t++) { wp.dy[t] = t / k - d->padding; wp.dx[t] = t % k - d->padding; }
    } else {                       // dw[ci][(co,t)] = sum_i x[ci][i] * dy[co][2 i + k]
        wp.a = x; wp.xin = dy; wp.a_gs = x_gs; wp.xin_gs = dy_gs; wp.Mw = d->Cin; wp.Cg = d->Cout; wp.Hg = OH; wp.Wg = OW; wp.gh = d->H; wp.gw = d->W;
        wp.sy = wp.sx = 2;
        for (int t = 0; t < k2; t++) { wp.dy[t] = t / k; wp.dx[t] = t % k; }
    }
    wp.status = armed_status_word(); wp.status_tag = status_tag_of(2, wp.Mw, wp.Cg);
    wp.c = dw; wp.c_gs = dw_gs; wp.G = G; wp.ntaps = k2; wp.wscale = wscale_of(d);
    wp.c_row_stride = wp.Cg * k2; wp.c_chan_stride = k2;                         // [Mw][Cg][taps]
    if (o.wt_oihw && d->kind == AG_CONV_TRANSPOSE) { wp.c_row_stride = k2; wp.c_chan_stride = wp.Mw * k2; }   // rows = Cin: [Cout][Cin][taps]
    wp.c_t = PtrTable{};
    if (o.dw_table) { wp.c_t = *o.dw_table; wp.c_row_stride = (int)o.dw_row_stride; }
    const int Kp = wp.gh * wp.gw, Nw = wp.Cg * k2;
    const int bm = pick_bm(wp.Mw), BN = bn_of(bm);
    const int tiles = ((Nw + BN - 1) / BN) * ((wp.Mw + bm - 1) / bm) * G;
    wp.mtiles = (wp.Mw + bm - 1) / bm;
    // pixel slices: same CU-share model as the gather kernel; a workgroup's fixed cost is its prologue plus the 16 K float atomics of
    // its epilogue (~6 K-tile-times); at least 8 K tiles per slice
    const int nkt_all = (Kp + BK - 1) / BK;
    int splits = 1;
    {
        // Round 5 (second session): since the slices are stored and added by wgrad_reduce_kernel (no atomics), a split costs one more write and one
        // more read of the whole [G][Mpad][Npad] tile set, at ~3 TB/s, in units of a K tile's time (0.45 us) -- 84 units per slice for the
        // 512 x 512 x 9, G = 6 layers, whose K loop is 256 tiles: the model now knows (AG_WGRAD_SPLIT_MODEL=0: the round-2 model, for the A/B)
        static const bool traffic_aware = [] { const char* e = getenv("AG_WGRAD_SPLIT_MODEL"); return !(e && e[0] == '0'); }();
        const double set_bytes = (double)G * round_up(wp.Mw, bm) * round_up(Nw, BN) * sizeof(float);
        const double per_split = traffic_aware ? 2.0 * set_bytes / 3e12 / 0.45e-6 : 0.0;
        double best_cost = 1e30;
        const int smax = std::max(1, std::min(nkt_all / 8, 4096));
        for (int sp = 1; sp <= smax; sp++) {
            const double cost = lanes_cost((long long)tiles * sp, (nkt_all + sp - 1) / sp, 6.0) + (sp > 1 ? 6.0 + per_split * sp : 0.0);
            if (cost < best_cost - 1e-9) { best_cost = cost; splits = sp; }
        }
    }
    if (const char* forced = getenv("AG_WGRAD_SPLITS")) {      // measurement hook (profiles/conv_split_sweep.py)
        const int f = atoi(forced);
        if (f >= 1) splits = std::min(f, std::max(1, nkt_all));
    }
    // destinations: instance g -> its own gradient, or (dw_table) possibly the same tensor as its neighbours: those instances are added up
    WgradReduce wr;
    wr.begin[0] = 0;
    int nruns = 0, longest_run = 1;
    for (int g = 0; g < G; g++) {
        const float* dst = o.dw_table ? o.dw_table->p[g] : dw + (size_t)g * dw_gs;
        if (nruns == 0 || dst != wr.dst.p[nruns - 1]) { wr.dst.p[nruns] = dst; wr.begin[nruns] = g; nruns++; }
        wr.begin[nruns] = g + 1;
        longest_run = std::max(longest_run, wr.begin[nruns] - wr.begin[nruns - 1]);
    }
    wp.Mpad = wp.mtiles * bm;
    wp.Npad = round_up(Nw, BN);
    // the slices' tiles live in the call's workspace (nothing else of it is in use by a weight gradient except the operand maxima behind it)
    const size_t tile_set = (size_t)G * wp.Mpad * wp.Npad * sizeof(float);
    const size_t part_room = workspace ? (size_t)G * packed_bytes(d) + kMaxPartialBytes : 0;
    if (workspace && workspace_bytes < conv_workspace_bytes_g(d, G)) { set_error("conv workspace too small"); return AG_ERR_SCRATCH_TOO_SMALL; }
    const int max_splits = (int)std::min<size_t>(4096, part_room / tile_set);
    if (splits > max_splits) splits = std::max(1, max_splits);
    wp.ksplit_len = round_up((Kp + splits - 1) / splits, BK);
    splits = (Kp + wp.ksplit_len - 1) / wp.ksplit_len;
    const bool via_partial = splits > 1 || longest_run > 1;
    if (via_partial && max_splits < 1) { set_error("conv workspace too small for the weight gradient's slices"); return AG_ERR_SCRATCH_TOO_SMALL; }
    wp.partial = via_partial ? reinterpret_cast<float*>(aligned_base(workspace)) : nullptr;
    dim3 grid((Nw + BN - 1) / BN, wp.mtiles * G, splits);
    char tag[64];
    snprintf(tag, sizeof(tag), "w bm%d G%d M%d C%d Kp%d t%d sp%d wg%lld %s", bm, G, wp.Mw, wp.Cg, Kp, k2, splits, (long long)grid.x * grid.y * grid.z,
             d->kind == AG_CONV ? (d->stride == 1 ? "s1" : "s2") : "T");
    ProfScope ps(AG_K_WGRAD, s, 2.0 * G * wp.Mw * (double)Kp * Nw, tag);
    const bool avec = (Kp & 3) == 0;      // rows of A 16-byte aligned
    wp.amax_a = wp.amax_b = nullptr;
    if (split_math()) {
        const int terms = split_terms();
        if (is_f16_form(terms)) {
            if (!workspace || workspace_bytes < conv_workspace_bytes_g(d, G)) { set_error("conv workspace too small"); return AG_ERR_SCRATCH_TOO_SMALL; }
            float* amax = reinterpret_cast<float*>(aligned_base(workspace) + (size_t)G * packed_bytes(d) + kMaxPartialBytes);
            const bool conv = d->kind == AG_CONV;
            const float* pre_a = conv ? o.amax_dy : o.amax_x;
            const float* pre_b = conv ? o.amax_x : o.amax_dy;
            AmaxTensor t[2] = { AmaxTensor{ wp.a, nullptr, wp.a_gs, (long long)wp.Mw * Kp, 0, 1 }, AmaxTensor{ wp.xin, nullptr, wp.xin_gs, (long long)wp.Cg * wp.Hg * wp.Wg, 0, 1 } };
            if (pre_a) t[0] = AmaxTensor{};
            if (pre_b) t[1] = AmaxTensor{};
            if ((rc = conv_absmax(t, 2, G, amax, s))) return rc;
            wp.amax_a = pre_a ? pre_a : amax;
            wp.amax_b = pre_b ? pre_b : amax + (size_t)kMaxGroups * kAmaxParts;
        }
#define AG_LAUNCH_WSPLIT_T(WMB, WNB, AV, NTM) hipLaunchKernelGGL((wgrad_split_kernel<WMB, WNB, 2, 4, AV, NTM>), grid, dim3(512), 0, s, wp)
#define AG_LAUNCH_WSPLIT_VT(WMB, WNB, NTM) hipLaunchKernelGGL((wgrad_split_kernel<WMB, WNB, 2, 4, true, NTM, true>), grid, dim3(512), 0, s, wp)
#define AG_LAUNCH_WSPLIT(WMB, WNB, AV) do { if (terms == 6) AG_LAUNCH_WSPLIT_T(WMB, WNB, AV, 6); else if (terms == 3) AG_LAUNCH_WSPLIT_T(WMB, WNB, AV, 3); \
                                            else if (terms == kF16S) AG_LAUNCH_WSPLIT_T(WMB, WNB, AV, kF16S); else AG_LAUNCH_WSPLIT_T(WMB, WNB, AV, kF16); } while (0)
#define AG_LAUNCH_WSPLIT_V(WMB, WNB) do { if (terms == 6) AG_LAUNCH_WSPLIT_VT(WMB, WNB, 6); else if (terms == 3) AG_LAUNCH_WSPLIT_VT(WMB, WNB, 3); \
                                          else if (terms == kF16S) AG_LAUNCH_WSPLIT_VT(WMB, WNB, kF16S); else AG_LAUNCH_WSPLIT_VT(WMB, WNB, kF16); } while (0)
        // the K-vectorised loader of the gathered operand: stride-1 "same" convolutions with rows of a multiple of 16 pixels (every 3 x 3 stride-1
        // layer of the product), whole K tiles per slice; AG_WGRAD_BVEC=0 keeps the scalar gathers (A/B)
        static const bool bvec_on = [] { const char* e = getenv("AG_WGRAD_BVEC"); return !(e && e[0] == '0'); }();
        const bool bvec = bvec_on && avec && d->kind == AG_CONV && d->stride == 1 && wp.gw == wp.Wg && wp.gh == wp.Hg && (wp.gw & 15) == 0 && wp.gw >= 16 &&
                          (Kp & 15) == 0 && (wp.ksplit_len & 15) == 0 && 2 * d->padding + 1 == k && k <= 3;
        if (bvec) {
            if (bm == 64) AG_LAUNCH_WSPLIT_V(1, 2); else AG_LAUNCH_WSPLIT_V(2, 1);
        } else
        if (bm == 64) {
            if (avec) AG_LAUNCH_WSPLIT(1, 2, true); else AG_LAUNCH_WSPLIT(1, 2, false);
        } else {
            if (avec) AG_LAUNCH_WSPLIT(2, 1, true); else AG_LAUNCH_WSPLIT(2, 1, false);
        }
#undef AG_LAUNCH_WSPLIT
#undef AG_LAUNCH_WSPLIT_V
#undef AG_LAUNCH_WSPLIT_T
#undef AG_LAUNCH_WSPLIT_VT
    } else if (bm == 64) {
        if (avec) hipLaunchKernelGGL((wgrad_kernel<1, 2, 2, 4, true>), grid, dim3(512), 0, s, wp);
        else      hipLaunchKernelGGL((wgrad_kernel<1, 2, 2, 4, false>), grid, dim3(512), 0, s, wp);
    } else {
        if (avec) hipLaunchKernelGGL((wgrad_kernel<2, 1, 2, 4, true>), grid, dim3(512), 0, s, wp);
        else      hipLaunchKernelGGL((wgrad_kernel<2, 1, 2, 4, false>), grid, dim3(512), 0, s, wp);
    }
    if ((rc = check_hip(hipGetLastError(), "wgrad_kernel")) || !via_partial) return rc;
    wr.partial = wp.partial; wr.G = G; wr.splits = splits; wr.Mw = wp.Mw; wr.Nw = Nw; wr.Mpad = wp.Mpad; wr.Npad = wp.Npad; wr.ntaps = k2;
    wr.c_row_stride = wp.c_row_stride; wr.c_chan_stride = wp.c_chan_stride;
    const long long total = (long long)wp.Mw * Nw;
    // many terms per element (the few-tile layers at large images): the terms of an element over 8 or 16 threads; AG_WGRAD_REDUCE_PAR=0: always
    // the one-thread-per-element kernel (the A/B)
    static const bool par_on = [] { const char* e = getenv("AG_WGRAD_REDUCE_PAR"); return !(e && e[0] == '0'); }();
    const int terms_per_elem = splits * longest_run;
    if (par_on && terms_per_elem >= 8) {
        if (terms_per_elem >= 48)
            hipLaunchKernelGGL(wgrad_reduce_par_kernel<16>, dim3((unsigned)std::min<long long>((total + 15) / 16, 8192), nruns), dim3(256), 0, s, wr);
        else
            hipLaunchKernelGGL(wgrad_reduce_par_kernel<32>, dim3((unsigned)std::min<long long>((total + 31) / 32, 8192), nruns), dim3(256), 0, s, wr);
        return check_hip(hipGetLastError(), "wgrad_reduce_par_kernel");
    }
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 2048), nruns), dim3(256), 0, s, wr);
    return check_hip(hipGetLastError(), "wgrad_reduce_kernel");
}

}  // namespace ag

extern "C" {

int ag_conv_forward(const AgConvDesc* d, const float* x, const float* w, const float* out_scale, const float* bias, float* y,
                    void* workspace, size_t workspace_bytes, void* stream)
{
    return conv_forward_g(d, 1, x, 0, table_of(w), out_scale, table_of(bias), y, 0, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream));
}

int ag_conv_backward_input(const AgConvDesc* d, const float* dy, const float* w, float* dx, void* workspace,
                           size_t workspace_bytes, void* stream)
{
    return conv_backward_input_g(d, 1, dy, 0, table_of(w), dx, 0, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream));
}

int ag_conv_backward_weight(const AgConvDesc* d, const float* x, const float* dy, float* dw, void* workspace,
                            size_t workspace_bytes, void* stream)
{
    return conv_backward_weight_g(d, 1, x, 0, dy, 0, dw, 0, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream));
}

int ag_conv_set_math(int mode)
{
    if (mode != AG_CONV_MATH_FP32_MFMA && mode != AG_CONV_MATH_SPLIT_BF16 && mode != AG_CONV_MATH_SPLIT_BF16X3 && mode != AG_CONV_MATH_SPLIT_F16 && mode != AG_CONV_MATH_F16) { set_error("unknown conv math mode"); return AG_ERR_INVALID_ARGUMENT; }
    g_conv_math.store(mode, std::memory_order_relaxed);
    return AG_OK;
}

int ag_conv_get_math(void) { return g_conv_math.load(std::memory_order_relaxed); }

int ag_conv_status(int clear) { return take_status(clear != 0); }



int ag_debug_mfma_rate_bf16(int blocks, int iters, float* out, void* stream)
{
    hipLaunchKernelGGL(mfma_rate_bf16_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), iters, out);
    return check_hip(hipGetLastError(), "mfma_rate_bf16_kernel");
}

int ag_debug_mfma_rate(int blocks, int iters, float* out, void* stream)
{
    hipLaunchKernelGGL(mfma_rate_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), iters, out);
    return check_hip(hipGetLastError(), "mfma_rate_kernel");
}

}  // extern "C"
