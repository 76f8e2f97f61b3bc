// Tile binning: scan of per-tile counts, instance scatter, per-tile depth sort (gfx950).
//
// Replaces cub::DeviceScan::InclusiveSum + duplicateWithKeys + cub::DeviceRadixSort::SortPairs +
// identifyTileRanges (reference cuda_rasterizer/rasterizer_impl.cu:70-138, 278-320).
//
// The reference builds 64-bit (tile | depth) keys in Gaussian order and runs a device-wide STABLE radix sort
// over 32+bit key bits, so inside a tile the order is (depth bits ascending, ties by Gaussian index, because
// duplicateWithKeys emits in index order).  Each Gaussian occurs at most once per tile, so that order is the
// unique ascending order of the 64-bit value (depth_bits << 32 | gaussian_index).  We therefore
//   1. count instances per tile (atomics in the preprocess kernel),
//   2. exclusive-scan the T tile counts in ONE workgroup -> ranges[tile] directly (no identifyTileRanges pass),
//   3. scatter (depth_bits<<32 | index) keys into each tile's segment in arbitrary order (atomic cursor),
//   4. sort every segment independently in LDS: sort-4 / sort-8 network per thread + merge-path rounds on the 64-bit keys, persistent
//      workgroups over the non-empty tiles in two size classes (<= 2048 and <= 8192 entries; a global bitonic network beyond).
// The sorted point_list and ranges are bit-identical to the reference's for every input, and the data moved is
// 8 B + 4 B per instance once, instead of 6+ radix passes over 12 B pairs.
#include <mutex>

#include "ag_common.h"

namespace ag {

// ---------------------------------------------------------------------------------------------------------
// 2. scan of tile counts (T = 4096 at 1024^2, 16384 at 2048^2): one 1024-thread workgroup, four tiles per thread
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) tile_scan_kernel(int T, const uint32_t* __restrict__ tile_count,
                                                        uint32_t* __restrict__ cursor, uint2* __restrict__ ranges,
                                                        uint32_t* __restrict__ num_rendered,
                                                        uint4* __restrict__ tile_order, uint32_t capacity, uint32_t skip_large)
{
    // Four consecutive tiles per thread and pass (one 16-byte load, one 16-byte cursor store, two 16-byte range stores): the
    // kernel is one workgroup deep, so its time is the number of dependent global round trips -- 2 per 4096 tiles this way,
    // 8 with one tile per thread.
    // Empty tiles (3500 of 4096 for an avatar) take no LDS atomics -- same-address LDS atomics serialise lane by lane and were
    // most of this kernel's time: their slot in tile_order is T - 1 - (empty tiles before them), known from the same scan.
    __shared__ uint32_t wave_sums[16], wave_ne[16];
    __shared__ uint32_t carry_s, carry_ne;
    __shared__ uint32_t cls_hist[34], cls_off[34];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { carry_s = 0; carry_ne = 0; }
    if (tid < 34) cls_hist[tid] = 0;
    __syncthreads();
    const uint4* count4 = reinterpret_cast<const uint4*>(tile_count);      // 256-byte aligned sub-array (ImageLayout)
    const bool one_pass = T <= 4096;          // counts and segment starts of this thread's 4 tiles stay in registers for the ordering phase
    uint32_t keep_c[4] = {0u, 0u, 0u, 0u}, keep_st[4] = {0u, 0u, 0u, 0u};
    for (int base = 0; base < T; base += 4096) {
        const int t0 = base + 4 * tid;
        uint32_t c[4] = {0u, 0u, 0u, 0u};
        if (t0 + 3 < T) {
            const uint4 v4 = count4[t0 >> 2];
            c[0] = v4.x; c[1] = v4.y; c[2] = v4.z; c[3] = v4.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) if (t0 + k < T) c[k] = tile_count[t0 + k];
        }
        const uint32_t mine = c[0] + c[1] + c[2] + c[3];
        const uint32_t ne_mine = (c[0] != 0u) + (c[1] != 0u) + (c[2] != 0u) + (c[3] != 0u);
        uint32_t v = mine, vn = ne_mine;                     // inclusive scans of the thread totals inside the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t n = __shfl_up(v, d, 64), nn = __shfl_up(vn, d, 64);
            if (lane >= d) { v += n; vn += nn; }
        }
        if (lane == 63) { wave_sums[wave] = v; wave_ne[wave] = vn; }
        __syncthreads();
        uint32_t woff = 0, woff_ne = 0;
        for (int w = 0; w < wave; w++) { woff += wave_sums[w]; woff_ne += wave_ne[w]; }
        const uint32_t carry = carry_s;
        uint32_t ne_before = carry_ne + woff_ne + vn - ne_mine;      // non-empty tiles before tile t0
        uint32_t start[5];
        start[0] = carry + woff + v - mine;
#pragma unroll
        for (int k = 0; k < 4; k++) start[k + 1] = start[k] + c[k];
        if (t0 + 3 < T) {
            reinterpret_cast<uint4*>(cursor)[t0 >> 2] = make_uint4(start[0], start[1], start[2], start[3]);
            uint4* r4 = reinterpret_cast<uint4*>(ranges + t0);   // empty tiles keep (0, 0), the reference's memset value
            r4[0] = make_uint4(c[0] ? start[0] : 0u, c[0] ? start[1] : 0u, c[1] ? start[1] : 0u, c[1] ? start[2] : 0u);
            r4[1] = make_uint4(c[2] ? start[2] : 0u, c[2] ? start[3] : 0u, c[3] ? start[3] : 0u, c[3] ? start[4] : 0u);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (t0 + k < T) {
                    cursor[t0 + k] = start[k];
                    ranges[t0 + k] = c[k] ? make_uint2(start[k], start[k + 1]) : make_uint2(0u, 0u);
                }
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (t0 + k < T) {
                if (c[k]) {
                    atomicAdd(&cls_hist[32 - __clz(c[k])], 1u);              // size class = bit length of the count
                    ne_before++;
                } else {
                    tile_order[(uint32_t)(T - 1) - ((uint32_t)(t0 + k) - ne_before)] = make_uint4((uint32_t)(t0 + k), 0u, 0u, 0u);
                }
            }
#pragma unroll
        for (int k = 0; k < 4; k++) { keep_c[k] = c[k]; keep_st[k] = start[k]; }
        __syncthreads();
        if (tid == 1023) { carry_s = start[4]; carry_ne = ne_before; }
        __syncthreads();
    }
    // instances, non-empty tiles, overflow flag.  `capacity` = instances the binning buffer was sized for when the later stages
    // were enqueued before this count was known (ag_raster_forward_optimistic): on overflow they must not touch it -- the
    // scatter returns on the flag, and with zero active tiles the sort and blend kernels have nothing to walk.
    // Round 5: num_rendered[3] = tiles of the large size classes (>= 2048 instances: bit length >= 12).  `skip_large`: the caller did NOT
    // enqueue the large-class sort launch behind this frame (no frame of the process has had such a tile so far: launch_bin_sort) -- if this
    // frame has one after all, it is refused exactly like an overflow (nothing downstream touches it; the host sees the flag and redoes the
    // frame, from then on with the launch).
    if (tid == 0) {
        uint32_t n_large = 0;
        for (int cls = 12; cls < 34; cls++) n_large += cls_hist[cls];
        const bool over = carry_s > capacity || (skip_large && n_large > 0u);
        num_rendered[0] = carry_s;
        num_rendered[1] = over ? 0u : carry_ne;
        num_rendered[2] = over ? 1u : 0u;
        num_rendered[3] = n_large;
    }
    // Work order for the persistent blend kernels: tiles by descending size class (longest-processing-time first),
    // empty tiles last.  Order inside a class is arbitrary.
    if (tid == 0) {
        uint32_t acc = 0;
        for (int cls = 33; cls >= 1; cls--) { cls_off[cls] = acc; acc += cls_hist[cls]; }
    }
    __syncthreads();
    for (int base = 0; base < T; base += 4096) {
        const int t0 = base + 4 * tid;
        if (t0 >= T) break;
        uint32_t c[4] = {0u, 0u, 0u, 0u}, st[4] = {0u, 0u, 0u, 0u};
        if (one_pass) {        // 1024^2: no second round trip to memory (the kernel is two dependent round trips deep otherwise)
#pragma unroll
            for (int k = 0; k < 4; k++) { c[k] = keep_c[k]; st[k] = keep_st[k]; }
        } else if (t0 + 3 < T) {      // written above by this workgroup (barriers in between): one round trip for both
            const uint4 v4 = count4[t0 >> 2], s4 = reinterpret_cast<const uint4*>(cursor)[t0 >> 2];
            c[0] = v4.x; c[1] = v4.y; c[2] = v4.z; c[3] = v4.w;
            st[0] = s4.x; st[1] = s4.y; st[2] = s4.z; st[3] = s4.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) if (t0 + k < T) { c[k] = tile_count[t0 + k]; st[k] = cursor[t0 + k]; }
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (t0 + k < T) {
                if (c[k]) tile_order[atomicAdd(&cls_off[32 - __clz(c[k])], 1u)] = make_uint4((uint32_t)(t0 + k), st[k], st[k] + c[k], 0u);
            }
    }
}

int launch_tile_scan(const AgRasterForwardArgs& a, hipStream_t s, uint32_t capacity, bool skip_large)
{
    const int gx = (a.W + kTileX - 1) / kTileX, gy = (a.H + kTileY - 1) / kTileY;
    char* ib = aligned_base(a.image_buffer);
    ImageLayout il((size_t)a.W, (size_t)a.H);
    { ProfScope ps(AG_K_TILE_SCAN, s); hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, s, gx * gy,
                       reinterpret_cast<const uint32_t*>(ib + il.tile_count),
                       reinterpret_cast<uint32_t*>(ib + il.cursor),
                       reinterpret_cast<uint2*>(ib + il.ranges),
                       reinterpret_cast<uint32_t*>(ib + il.num_rendered),
                       reinterpret_cast<uint4*>(ib + il.tile_order), capacity, skip_large ? 1u : 0u); }
    return check_hip(hipGetLastError(), "tile_scan_kernel");
}

// ---------------------------------------------------------------------------------------------------------
// 3. scatter: one thread per Gaussian, one (depth|index) key per touched tile
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void get_rect_i(float px, float py, int max_radius, int gx, int gy,
                                           uint32_t& x0o, uint32_t& y0o, uint32_t& x1o, uint32_t& y1o)
{
    // identical to the preprocess kernel (auxiliary.h:46-56); only exact fp32 adds and a division by 16
    const float r = (float)max_radius;
    int x0 = (int)((px - r) / (float)kTileX);
    int y0 = (int)((py - r) / (float)kTileY);
    int x1 = (int)((px + r + (float)kTileX - 1.0f) / (float)kTileX);
    int y1 = (int)((py + r + (float)kTileY - 1.0f) / (float)kTileY);
    x0 = max(0, x0); y0 = max(0, y0); x1 = max(0, x1); y1 = max(0, y1);
    x0o = (uint32_t)min(gx, x0); y0o = (uint32_t)min(gy, y0);
    x1o = (uint32_t)min(gx, x1); y1o = (uint32_t)min(gy, y1);
}

constexpr int kWinBins = 2048;   // same workgroup tile window as the preprocess kernel

__global__ void __launch_bounds__(256) scatter_kernel(int P, int gx, int gy, const int* __restrict__ radii,
                                                     const GaussRec* __restrict__ rec, uint32_t* __restrict__ cursor,
                                                     uint64_t* __restrict__ keys, const uint32_t* __restrict__ counts)
{
    if (counts[2]) return;   // more instances than the binning buffer holds (optimistic forward): the caller redoes the frame
    // Slots are handed out per (workgroup, tile): instances are counted in an LDS histogram over the workgroup's
    // tile window, one global atomic per touched tile reserves the workgroup's block of the tile segment, and the
    // rank inside the block comes from a second pass of LDS atomics.  Order inside a segment is arbitrary (the sort
    // fixes it), so this is equivalent to one global atomic per instance at a fraction of the contention.
    __shared__ int s_win[4];
    __shared__ uint32_t s_hist[kWinBins];
    __shared__ uint32_t s_base[kWinBins];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x == 0) { s_win[0] = 0x7fffffff; s_win[1] = 0x7fffffff; s_win[2] = 0; s_win[3] = 0; }
    __syncthreads();
    uint32_t x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    uint64_t key = 0;
    const int rad = (idx < P) ? radii[idx] : 0;
    if (rad > 0) {
        const float px = rec[idx].x, py = rec[idx].y;
        const uint32_t dbits = __float_as_uint(rec[idx].depth);
        get_rect_i(px, py, rad, gx, gy, x0, y0, x1, y1);
        key = ((uint64_t)dbits << 32) | (uint32_t)idx;
    }
    window_accumulate(s_win, x1 > x0 && y1 > y0, (int)x0, (int)y0, (int)x1, (int)y1);
    __syncthreads();
    const int wx0 = s_win[0], wy0 = s_win[1];
    const int bw = s_win[2] - wx0, bh = s_win[3] - wy0;
    if (bw <= 0 || bh <= 0) return;
    const int nb = bw * bh;
    if (nb <= kWinBins) {
        for (int i = threadIdx.x; i < nb; i += 256) s_hist[i] = 0u;
        __syncthreads();
        for (uint32_t y = y0; y < y1; y++)
            for (uint32_t x = x0; x < x1; x++) atomicAdd(&s_hist[((int)y - wy0) * bw + ((int)x - wx0)], 1u);
        __syncthreads();
        for (int i = threadIdx.x; i < nb; i += 256) {
            const uint32_t c = s_hist[i];
            if (c) s_base[i] = atomicAdd(&cursor[(uint32_t)(wy0 + i / bw) * (uint32_t)gx + (uint32_t)(wx0 + i % bw)], c);
            s_hist[i] = 0u;
        }
        __syncthreads();
        for (uint32_t y = y0; y < y1; y++)
            for (uint32_t x = x0; x < x1; x++) {
                const int li = ((int)y - wy0) * bw + ((int)x - wx0);
                const uint32_t r = atomicAdd(&s_hist[li], 1u);
                keys[s_base[li] + r] = key;
            }
    } else {
        for (uint32_t y = y0; y < y1; y++)
            for (uint32_t x = x0; x < x1; x++) {
                const uint32_t pos = atomicAdd(&cursor[y * (uint32_t)gx + x], 1u);
                keys[pos] = key;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------
// 4. per-tile sort of the u64 keys (depth_bits << 32 | gaussian_index; unique, so any correct sort gives THE order).
//    Merge sort in LDS: each thread sorts 8 keys in registers (19-comparator network), then log2(n/8) rounds of pairwise
//    run merges between two LDS buffers, every thread producing 8 consecutive outputs of its pair after a merge-path
//    binary search.  10 rounds with ~30 dependent LDS reads each for an 8192-entry tile, against 91 barrier-separated
//    compare-exchange stages of a bitonic network.  Two size classes: <= 2048 entries (256 threads, 32 KiB LDS) and
//    <= 8192 (1024 threads, 128 KiB LDS).  Larger tiles (never reached by avatar-sized splats) fall back to an
//    all-ascending bitonic network run directly on the global segment by one workgroup.
//    Round 2 tried the opposite trade: a bitonic network held in registers (K consecutive keys per thread, stages at distance < K in
//    registers, < 64 K between lanes with DPP / ds_swizzle / one bpermute, only the 3-10 widest stages through LDS + barriers).  Bit-
//    identical (all raster tests), but 22 / 59 / 66 us on the front / oblique / side views against 21 / 36 / 48 here: the tiles of an
//    avatar view are long (886 entries on average, up to 6015), so the network's n log^2 n exchanges cost more than this sort's chains
//    of dependent LDS reads.  Not kept.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cswap(uint64_t& a, uint64_t& b)
{
    const uint64_t lo = a < b ? a : b, hi = a < b ? b : a;
    a = lo; b = hi;
}

__device__ __forceinline__ void sort8(uint64_t (&v)[8])
{
    cswap(v[0], v[1]); cswap(v[2], v[3]); cswap(v[4], v[5]); cswap(v[6], v[7]);
    cswap(v[0], v[2]); cswap(v[1], v[3]); cswap(v[4], v[6]); cswap(v[5], v[7]);
    cswap(v[1], v[2]); cswap(v[5], v[6]); cswap(v[0], v[4]); cswap(v[3], v[7]);
    cswap(v[1], v[5]); cswap(v[2], v[6]);
    cswap(v[1], v[4]); cswap(v[3], v[6]);
    cswap(v[2], v[4]); cswap(v[3], v[5]);
    cswap(v[3], v[4]);
}

__device__ __forceinline__ void sort4(uint64_t (&v)[4])
{
    cswap(v[0], v[1]); cswap(v[2], v[3]);
    cswap(v[0], v[2]); cswap(v[1], v[3]);
    cswap(v[1], v[2]);
}

template <int K> __device__ __forceinline__ void sort_regs(uint64_t (&v)[K]);
template <> __device__ __forceinline__ void sort_regs<8>(uint64_t (&v)[8]) { sort8(v); }
template <> __device__ __forceinline__ void sort_regs<4>(uint64_t (&v)[4]) { sort4(v); }

constexpr uint64_t kKeyInf = ~0ull;

// ---- wave-level bitonic network: 64 lanes x 4 keys in registers -> one ascending run of 256 (round 3) ----
// The first six merge rounds of a tile's sort (runs of 4 -> 256) used to go through LDS like the later ones: a merge-path binary search and
// four dependent LDS reads per thread and round, ~1.6 us each on the bench view's ~1000-entry tiles whatever the run length.  Inside a
// wave the same merges are 33 compare-exchange stages on registers: element i = lane * 4 + r, partner i ^ j; j = 1, 2 are in the lane's own
// registers, j = 4 .. 128 are lane ^ 1, 2 (DPP quad_perm), ^ 4, 8, 16 (ds_swizzle bit mode: the LDS crossbar, no memory) and ^ 32 (bpermute).
// (Round 2 measured a FULL register-held bitonic sort of long tiles slower than the merge sort: its n log^2 n exchanges lose beyond a few
// hundred keys.  Here the network stops at the wave's 256 keys and the merge-path rounds take over at L = 256.)
template <int D>
__device__ __forceinline__ uint32_t lane_xor32(uint32_t v)
{
    if constexpr (D == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);        // quad_perm [1,0,3,2]
    else if constexpr (D == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    else if constexpr (D == 4) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x101f);                     // and 0x1f, xor 4
    else if constexpr (D == 8) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x201f);
    else if constexpr (D == 16) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401f);
    else return (uint32_t)__shfl_xor((int)v, 32, 64);
}

template <int D>
__device__ __forceinline__ uint64_t lane_xor64(uint64_t v)
{
    return ((uint64_t)lane_xor32<D>((uint32_t)(v >> 32)) << 32) | lane_xor32<D>((uint32_t)v);
}

__device__ __forceinline__ void cswap_dir(uint64_t& a, uint64_t& b, bool asc)   // asc: a <= b afterwards, else a >= b
{
    const bool sw = (a > b) == asc;
    const uint64_t x = sw ? b : a, y = sw ? a : b;
    a = x; b = y;
}

template <int K, int J>
__device__ __forceinline__ void bitonic_stage(uint64_t (&v)[4], int lane)
{
    const bool asc = (lane & (K / 4)) == 0;        // direction of the run of K this element's merge builds (K = 256: always ascending)
    if constexpr (J >= 4) {
        const bool want_min = ((lane & (J / 4)) == 0) == asc;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint64_t o = lane_xor64<J / 4>(v[r]);
            v[r] = ((v[r] < o) == want_min) ? v[r] : o;
        }
    } else if constexpr (J == 2) {
        cswap_dir(v[0], v[2], asc); cswap_dir(v[1], v[3], asc);
    } else {
        cswap_dir(v[0], v[1], asc); cswap_dir(v[2], v[3], asc);
    }
    if constexpr (J > 1) bitonic_stage<K, J / 2>(v, lane);
}

// v[0..3] of every lane: any keys -> lane l holds elements 4l .. 4l+3 of the wave's 256 keys in ascending order
__device__ __forceinline__ void wave_sort256(uint64_t (&v)[4], int lane)
{
    sort4(v);
    if (lane & 1) { uint64_t t = v[0]; v[0] = v[3]; v[3] = t; t = v[1]; v[1] = v[2]; v[2] = t; }   // runs of 4: ascending in even lanes, descending in odd ones
    bitonic_stage<8, 4>(v, lane);
    bitonic_stage<16, 8>(v, lane);
    bitonic_stage<32, 16>(v, lane);
    bitonic_stage<64, 32>(v, lane);
    bitonic_stage<128, 64>(v, lane);
    bitonic_stage<256, 128>(v, lane);
}
#ifndef AG_SORT_WAVE_NETWORK
#define AG_SORT_WAVE_NETWORK 1
#endif
constexpr int kSortSmallCap = 2048;   // 512 threads x 4 keys (half the sequential merge steps per round of 256 x 8)
constexpr int kSortLargeCap = 8192;   // 1024 threads x 8 keys

template <typename KeyPtr>
__device__ __forceinline__ void bitonic_sort_global(KeyPtr sk, uint32_t n, uint32_t m, int tid, int nthreads)
{
    const uint32_t half = m >> 1;
    for (uint32_t k = 2; k <= m; k <<= 1) {
        const uint32_t hk = k >> 1, lk = __builtin_ctz(hk);
        for (uint32_t t = tid; t < half; t += nthreads) {
            const uint32_t blk = t >> lk, off = t & (hk - 1);
            const uint32_t i = (blk << (lk + 1)) + off, p = (blk << (lk + 1)) + k - 1 - off;
            if (p < n) { const uint64_t a = sk[i], b = sk[p]; if (a > b) { sk[i] = b; sk[p] = a; } }
        }
        __syncthreads();
        for (uint32_t j = k >> 2; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < half; t += nthreads) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i + j;
                if (p < n) { const uint64_t a = sk[i], b = sk[p]; if (a > b) { sk[i] = b; sk[p] = a; } }
            }
            __syncthreads();
        }
    }
}

// LDS merge sort of one segment of n <= NT * KPT keys by the whole workgroup.  PRESORTED: the segment already consists of sorted runs
// of `run0` keys (the chunk pass below), so the register sort and the merge rounds below run0 are skipped.  The result goes to
// `out_list` (the low 32 bits: Gaussian indices) or, for a chunk, back to `out_keys`.
template <int NT, int KPT, bool PRESORTED>
__device__ __forceinline__ void sort_segment_lds(uint64_t* __restrict__ sk, const uint64_t* __restrict__ seg, uint32_t n, uint32_t run0,
                                                 uint32_t* __restrict__ out_list, uint64_t* __restrict__ out_keys, int tid)
{
    constexpr uint32_t CAP = NT * KPT;
    uint64_t* bufA = sk;
    uint64_t* bufB = sk + CAP;
    // 1. KPT keys per thread, sorted in registers; slots past n hold +inf and simply stay at the top
    uint64_t v[KPT];
    const uint32_t base = (uint32_t)tid * (uint32_t)KPT;
#pragma unroll
    for (int i = 0; i < KPT; i++) v[i] = (base + i < n) ? seg[base + i] : kKeyInf;
    constexpr bool WAVE_RUNS = !PRESORTED && KPT == 4 && AG_SORT_WAVE_NETWORK;   // runs of 256 from the wave network instead of runs of KPT
    if constexpr (WAVE_RUNS) {
        if constexpr (KPT == 4) wave_sort256(v, tid & 63);
    } else if (!PRESORTED) sort_regs<KPT>(v);
    if (base < n) {
#pragma unroll
        for (int i = 0; i < KPT; i++) bufA[base + i] = v[i];
    }
    __syncthreads();
    // 2. merge rounds over the first n8 = ceil(n / KPT) * KPT slots
    const uint32_t n8 = (n + (uint32_t)KPT - 1u) & ~((uint32_t)KPT - 1u);
    for (uint32_t L = PRESORTED ? run0 : (WAVE_RUNS ? 256u : (uint32_t)KPT); L < n8; L <<= 1) {
        if (base < n8) {
            const uint32_t ps = base & ~(2u * L - 1u);          // start of this thread's run pair
            const uint32_t o = base - ps;                        // first output index inside the merged pair
            const uint32_t lx = min(L, n8 - ps);                 // |X|
            const uint32_t ly = (ps + L < n8) ? min(L, n8 - ps - L) : 0u;   // |Y|
            const uint64_t* X = bufA + ps;
            const uint64_t* Y = bufA + ps + L;
            // merge path: smallest i with X[i] > Y[o - i - 1] (keys are unique)
            uint32_t lo = (o > ly) ? o - ly : 0u, hi = min(o, lx);
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (X[mid] < Y[o - mid - 1]) lo = mid + 1; else hi = mid;
            }
            uint32_t i = lo, j = o - lo;
            uint64_t xv = (i < lx) ? X[i] : kKeyInf, yv = (j < ly) ? Y[j] : kKeyInf;
#pragma unroll
            for (int r = 0; r < KPT; r++) {
                const bool takex = xv <= yv;
                bufB[base + r] = takex ? xv : yv;
                if (takex) { i++; xv = (i < lx) ? X[i] : kKeyInf; }
                else       { j++; yv = (j < ly) ? Y[j] : kKeyInf; }
            }
        }
        __syncthreads();
        uint64_t* t = bufA; bufA = bufB; bufB = t;
    }
    if (out_list) { for (uint32_t i = tid; i < n; i += NT) out_list[i] = (uint32_t)bufA[i]; }
    else          { for (uint32_t i = tid; i < n; i += NT) out_keys[i] = bufA[i]; }
}

// Workgroups walk the non-empty tiles in tile_scan_kernel's longest-first order with a grid stride, so only as many
// workgroups are launched as can be resident (the 128 KiB class would otherwise queue 4096 one-per-CU launches).
// Round 2: a tile of 2049 .. 8192 entries used to be sorted start to finish by ONE 1024-thread workgroup (36-48 us on the oblique and
// side views, whose longest tiles have 3000-6000 entries, against 20 us for the small class).  Now the SMALL kernel also sorts the
// 2048-entry chunks of those tiles (chunk item c of tile rank r goes to workgroup grid - 1 - (4 r + c) % grid, before its own tiles) and the
// LARGE kernel only merges the <= 4 sorted runs (two merge rounds instead of ten).
template <int NT, bool LARGE, int KPT>
__global__ void __launch_bounds__(NT) tile_sort_kernel(const uint4* __restrict__ tile_order, const uint32_t* __restrict__ counts,
                                                      uint64_t* __restrict__ keys, uint32_t* __restrict__ point_list)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t sk[];   // two buffers of NT * KPT keys
    constexpr uint32_t CAP = NT * KPT;
    const uint32_t n_active = counts[1];
    const int tid = threadIdx.x;
    if (!LARGE) {
        // chunk items of the large-class tiles (they lead tile_order: it is sorted by descending bit length of the count)
        // dealt from the END of the grid: with fewer active tiles than workgroups (640 against 1280 on the avatar views) the chunks go to
        // workgroups that have no tile of their own
        for (uint32_t item = gridDim.x - 1u - blockIdx.x; item < 4u * n_active; item += gridDim.x) {
            const uint4 wd = tile_order[item >> 2];
            const uint32_t n = wd.z - wd.y;
            if (n < 2048u) break;                                        // no large tile at this rank or behind it
            if (n <= (uint32_t)kSortSmallCap || n > (uint32_t)kSortLargeCap) continue;
            const uint32_t c0 = (item & 3u) * (uint32_t)kSortSmallCap;
            if (c0 >= n) continue;
            __syncthreads();   // LDS buffers of the previous item are free
            uint64_t* seg = keys + wd.y + c0;
            sort_segment_lds<NT, KPT, false>(sk, seg, min((uint32_t)kSortSmallCap, n - c0), 0u, nullptr, seg, tid);
        }
    }
    for (uint32_t rank = blockIdx.x; rank < n_active; rank += gridDim.x) {
    const uint4 wd = tile_order[rank];
    const uint2 rg = make_uint2(wd.y, wd.z);
    const uint32_t n = rg.y - rg.x;
    // size classes: the small kernel takes (0, kSortSmallCap], the large one everything above.  The order is by
    // descending bit length of n, so once the large kernel meets a tile below 2048 entries it is done.
    if (LARGE) { if (n < 2048u) break; if (n <= (uint32_t)kSortSmallCap) continue; }
    else if (n > (uint32_t)kSortSmallCap) continue;
    uint64_t* seg = keys + rg.x;
    __syncthreads();   // LDS buffers of the previous tile are free
    if (n > CAP) {
        uint32_t m = 2;
        while (m < n) m <<= 1;
        bitonic_sort_global(seg, n, m, tid, NT);   // workgroup barriers order the exchanges (one CU, one L1)
        for (uint32_t i = tid; i < n; i += NT) point_list[rg.x + i] = (uint32_t)seg[i];
        continue;
    }
    if (LARGE) sort_segment_lds<NT, KPT, true>(sk, seg, n, (uint32_t)kSortSmallCap, point_list + rg.x, nullptr, tid);
    else       sort_segment_lds<NT, KPT, false>(sk, seg, n, 0u, point_list + rg.x, nullptr, tid);
    }
}

int launch_bin_sort(const AgRasterForwardArgs& a, int R, hipStream_t s, bool skip_large)
{
    const int gx = (a.W + kTileX - 1) / kTileX, gy = (a.H + kTileY - 1) / kTileY;
    if (R <= 0) return AG_OK;
    char* gb = aligned_base(a.geom_buffer);
    char* ib = aligned_base(a.image_buffer);
    char* bb = aligned_base(a.binning_buffer);
    GeomLayout gl((size_t)a.P);
    ImageLayout il((size_t)a.W, (size_t)a.H);
    BinLayout bl((size_t)R);
    uint64_t* keys = reinterpret_cast<uint64_t*>(bb + bl.keys);
    uint32_t* point_list = reinterpret_cast<uint32_t*>(bb + bl.point_list);
    { ProfScope ps(AG_K_SCATTER, s); hipLaunchKernelGGL(scatter_kernel, dim3((a.P + 255) / 256), dim3(256), 0, s, a.P, gx, gy, a.radii,
                       reinterpret_cast<const GaussRec*>(gb + gl.rec), reinterpret_cast<uint32_t*>(ib + il.cursor), keys,
                       reinterpret_cast<const uint32_t*>(ib + il.num_rendered)); }
    if (check_hip(hipGetLastError(), "scatter_kernel")) return AG_ERR_HIP;
    constexpr size_t kSmallLds = 2ull * kSortSmallCap * sizeof(uint64_t), kLargeLds = 2ull * kSortLargeCap * sizeof(uint64_t);
    {
        // function attributes are per device: remember which devices have it (a bit per ordinal), under a lock
        static std::mutex mu;
        static uint64_t done = 0;
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lk(mu);
        if (dev < 0 || dev >= 64 || !((done >> dev) & 1)) {
            if (check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_sort_kernel<1024, true, 8>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLargeLds), "sort LDS attr"))
                return AG_ERR_HIP;
            if (dev >= 0 && dev < 64) done |= 1ull << dev;
        }
    }
    {
        ProfScope ps(AG_K_TILE_SORT, s);
        const uint4* order = reinterpret_cast<const uint4*>(ib + il.tile_order);
        const uint32_t* counts = reinterpret_cast<const uint32_t*>(ib + il.num_rendered);
        const int T = gx * gy;
        hipLaunchKernelGGL((tile_sort_kernel<512, false, 4>), dim3(T < 1280 ? T : 1280), dim3(512), kSmallLds, s, order, counts, keys, point_list);
        // the large class (tiles of 2049 .. 8192 instances merged from their sorted chunks, longer ones sorted in global memory): 256 workgroups
        // that each need a whole CU's LDS.  On avatar views no tile is that long and the launch was 4 us of the one-stream view and a 27-us slot
        // in the overlapped pipeline for nothing (round-4 review): skipped until a frame of the process has had such a tile (tile_scan_kernel
        // refuses a frame that needs it when it was skipped; ag_abi.hip keeps the sticky flag)
        if (!skip_large)
            hipLaunchKernelGGL((tile_sort_kernel<1024, true, 8>), dim3(T < 256 ? T : 256), dim3(1024), kLargeLds, s, order, counts, keys, point_list);
    }
    return check_hip(hipGetLastError(), "tile_sort_kernel");
}

}  // namespace ag
