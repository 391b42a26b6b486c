// Tile binning + depth ordering for gfx950.
//
// Replaces gsplat's isect_tiles + cub::DeviceRadixSort over 64-bit (tile | depth) keys +
// isect_offset_encode [UPSTREAM gsplat >= 1.5, not vendored; SURVEY.md App. A item 3].
// The OUTPUT is bit-identical to that pipeline (sorted isect_ids / flatten_ids, tile offsets),
// the ROUTE is different and chosen for HBM traffic:
//
//   upstream : emit I (key64,val32) pairs, 6 LSD radix passes over I items of 12 B
//              (~144 B of traffic per intersection);
//   here     : (1) LSD radix sort of the N per-Gaussian (depth bits, id) pairs -- 4 passes over
//              N items of 8 B; (2) exclusive scan of tiles-per-Gaussian in depth order;
//              (3) emit (tile id, Gaussian id) in depth order; (4) a STABLE radix sort on the
//              tile id alone -- 2 passes over I items of 8 B.  Stability makes the final order
//              (tile, depth, id) == the order of a stable sort on (tile<<32 | depth) with ties in
//              emit order, i.e. exactly upstream's.  ~48 B of traffic per intersection at I ~ 3.5 N.
//
// Radix pass = 3 launches (per-block digit histogram, per-digit scan over blocks, stable
// scatter).  The scatter ranks keys inside a wavefront with 8 ballots (wave64 match-any),
// so a round of 256 keys costs ~3 barriers and no LDS atomics, and preserves input order.
#include "adk_common.hpp"
#include "radix_sort.hpp"

namespace adk {

// ------------------------------------------------------------------ tiles-per-Gaussian scan (depth order)
__global__ __launch_bounds__(256) void count_block_sums_kernel(const uint32_t* __restrict__ sorted_ids,
                                                               const int32_t* __restrict__ tiles_per_gauss, int N,
                                                               uint32_t* __restrict__ block_sums)
{
    __shared__ uint32_t ws[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    uint32_t c = (i < N) ? (uint32_t)tiles_per_gauss[sorted_ids[i]] : 0u;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// Single block: exclusive scan of block_sums (in place) + grand total.
__global__ __launch_bounds__(1024) void scan_block_sums_kernel(uint32_t* __restrict__ block_sums, int nb,
                                                               int64_t* __restrict__ total_out)
{
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nb; b0 += 1024) {
        const int i = b0 + threadIdx.x;
        const uint32_t v = (i < nb) ? block_sums[i] : 0u;
        uint32_t s = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(s, o, 64); if (lane >= o) s += t; }
        if (lane == 63) wsum[wv] = s;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wv; ++w) woff += wsum[w];
        const uint32_t carry = carry_s;
        if (i < nb) block_sums[i] = carry + woff + s - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + s;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = (int64_t)carry_s;
}

__device__ __forceinline__ void tile_range_bin(float mx, float my, float rx, float ry, int tile_w, int tile_h,
                                               int& x0, int& x1, int& y0, int& y1)
{
    const float ts = 16.0f;
    const float tx = mx / ts, ty = my / ts, trx = rx / ts, try_ = ry / ts;
    x0 = (int)fminf(fmaxf(floorf(tx - trx), 0.f), (float)tile_w);
    x1 = (int)fminf(fmaxf(ceilf(tx + trx), 0.f), (float)tile_w);
    y0 = (int)fminf(fmaxf(floorf(ty - try_), 0.f), (float)tile_h);
    y1 = (int)fminf(fmaxf(ceilf(ty + try_), 0.f), (float)tile_h);
}

// Thread i handles the i-th Gaussian in depth order and writes its (tile id, Gaussian id) pairs
// at the exclusive-scan offset, tiles in row-major order (y outer) like upstream's emit loop.
__global__ __launch_bounds__(256) void emit_kernel(const uint32_t* __restrict__ sorted_ids,
                                                   const int32_t* __restrict__ tiles_per_gauss,
                                                   const float* __restrict__ rec, int N, int tile_w, int tile_h,
                                                   const uint32_t* __restrict__ block_offs, int64_t capacity,
                                                   uint32_t* __restrict__ tile_ids, uint32_t* __restrict__ gauss_ids)
{
    __shared__ uint32_t ws[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t g = 0, c = 0;
    if (i < N) { g = sorted_ids[i]; c = (uint32_t)tiles_per_gauss[g]; }
    uint32_t s = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(s, o, 64); if (lane >= o) s += t; }
    if (lane == 63) ws[wv] = s;
    __syncthreads();
    uint32_t off = block_offs[blockIdx.x] + s - c;
    for (int w = 0; w < wv; ++w) off += ws[w];
    int x0 = 0, x1 = 0, y0 = 0, y1 = 0;
    if (c != 0) {
        const float4* r4 = reinterpret_cast<const float4*>(rec) + 3 * (int64_t)g;
        const float4 a = r4[0], b = r4[1];
        tile_range_bin(a.x, a.y, a.w, b.w, tile_w, tile_h, x0, x1, y0, y1);
    }
    // Round 6 (finding 62): a Gaussian with more than BIN_BIG_TILES tiles is emitted by its whole WAVE (lane l writes entries l, l + 64, ...), not by
    // the one lane that owns it: a screen-filling splat is 8 160 serial store pairs at 1080p otherwise.  Same entries at the same offsets.
    const bool big = c > 64u; // = BIN_BIG_TILES (defined below)
    unsigned long long bm = __ballot(big);
    while (bm) {
        const int src = __builtin_ctzll(bm);
        bm &= bm - 1;
        const uint32_t gg = (uint32_t)__shfl((int)g, src, 64), oo = (uint32_t)__shfl((int)off, src, 64), cc = (uint32_t)__shfl((int)c, src, 64);
        const int bx0 = __shfl(x0, src, 64), bx1 = __shfl(x1, src, 64), by0 = __shfl(y0, src, 64);
        const int w = bx1 - bx0;
        for (uint32_t i = (uint32_t)lane; i < cc; i += 64u) {
            const int r = (int)i / w;
            const int64_t o = (int64_t)oo + i;
            if (o < capacity) { tile_ids[o] = (uint32_t)((by0 + r) * tile_w + bx0 + ((int)i - r * w)); gauss_ids[o] = gg; }
        }
    }
    if (c == 0 || big) return;
    for (int ty = y0; ty < y1; ++ty)
        for (int tx = x0; tx < x1; ++tx) {
            if ((int64_t)off < capacity) { tile_ids[off] = (uint32_t)(ty * tile_w + tx); gauss_ids[off] = g; }
            ++off;
        }
}

// offsets[t] = first index in the tile-sorted list whose tile id >= t (isect_offset_encode).
__global__ __launch_bounds__(256) void tile_offsets_kernel(const uint32_t* __restrict__ tile_sorted, int64_t n_isects,
                                                           int n_tiles, int32_t* __restrict__ offsets)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n_isects == 0) { if (i < n_tiles) offsets[i] = 0; return; }
    if (i >= n_isects) return;
    const int cur = (int)tile_sorted[i];
    const int prev = (i == 0) ? -1 : (int)tile_sorted[i - 1];
    for (int t = prev + 1; t <= cur; ++t) offsets[t] = (int32_t)i;
    if (i == n_isects - 1)
        for (int t = cur + 1; t < n_tiles; ++t) offsets[t] = (int32_t)n_isects;
}

__global__ __launch_bounds__(256) void make_isect_ids_kernel(const uint32_t* __restrict__ tile_sorted,
                                                             const int32_t* __restrict__ flatten_ids,
                                                             const uint32_t* __restrict__ depth_keys, int64_t n,
                                                             int64_t* __restrict__ isect_ids)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    isect_ids[i] = ((int64_t)tile_sorted[i] << 32) | (int64_t)depth_keys[flatten_ids[i]];
}

// Total number of (Gaussian, tile) intersections.  It only depends on the projection, so the host can
// fetch it while the depth sort is still running instead of stalling the stream after it.
__global__ __launch_bounds__(256) void count_isects_kernel(const int32_t* __restrict__ tiles_per_gauss, int N,
                                                           unsigned long long* __restrict__ total)
{
    __shared__ unsigned long long red[4];
    unsigned long long s = 0;
    const int stride = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) s += (unsigned)tiles_per_gauss[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { // one same-address atomic per workgroup, <= 256 in total
        const unsigned long long t = (red[0] + red[1]) + (red[2] + red[3]);
        if (t) atomicAdd(total, t);
    }
}


// ================================================================================================================
// Round 2: TILE-LOCAL binning.  The global route above sorts N depth keys (4 radix passes) and then I tile ids (2 passes)
// in 18 dependent launches of 5-35 us on lists far too small to hide them (0.27 ms at 1 M / 1080p, ~5 % of HBM speed).
// The order upstream asks for -- (tile, depth bits, Gaussian id) -- does not need a global sort at all:
//   1. counting sort by tile: every workgroup owns a contiguous slice of the Gaussians and a histogram over ALL tiles in LDS
//      (4 B x tiles: 32 KB at 1080p, 79 KB at 2592x1944), counts each covered tile, and the [slices][tiles] table is
//      scanned (column scan per tile, then one small scan over the tiles = the `offsets` output and n_isects);
//   2. the same slices scatter their (depth bits << 32 | id) keys to LDS cursors initialised from the table: every tile's
//      segment is now complete, in arbitrary order;
//   3. one workgroup per tile sorts its segment IN LDS (bitonic on the 64-bit keys; unique keys, so the result is exactly
//      the stable (tile, depth, id) order) and writes the ids back in place.
// 5 launches, one read of the records per step 1/2, 8 B per intersection written and read twice.  Bit-identical output.
// Falls back to the global route when the tile histogram does not fit LDS or a tile holds more than BIN_SORT_BIG entries
// (the host learns the largest tile together with n_isects, the one value it waits for anyway).
#ifndef BIN_SLICES
#define BIN_SLICES 256
#endif
#ifndef BIN_THREADS
#define BIN_THREADS 1024
#endif
// per slice: 16 waves share one LDS histogram (256 slices x 4 waves left most of the chip idle: 89 us -> see DESIGN)
#define BIN_SORT_BIG 8192
// Round 6 (finding 62): a Gaussian whose tile rectangle has more than BIN_BIG_TILES tiles is not walked by the one thread that owns it but
// by its whole WAVE (lane l takes tiles l, l + 64, ...; the owner's rectangle and key are read with v_readlane).  The per-thread walk is
// serial -- one LDS atomic (count) or one returning LDS atomic + an 8 B store (scatter) per tile -- and the kernel ends with its slowest
// thread: over a 1 000-frame sequence the optimiser grows a handful of Gaussians until they cover most of the frame (up to 8 160 tiles at
// 1080p), and those alone took bin_count from 0.03 to 0.26 ms and bin_scatter from 0.07 to 0.68 ms per optimisation step by frame 500
// (profiles/r06_sequence_drift.txt) -- +48 % on the whole step, invisible on the benchmark clouds (radius 6-7 px).  Wave-level,
// not workgroup-level: it needs no list, no barrier and no cap, and a frame whose TYPICAL rectangle is above the threshold (a close-up)
// costs what the per-thread walk costs (64 lanes x one pass each) instead of 1 024 threads idling behind 64.  Same counts, and the order
// inside a tile's segment was already arbitrary (the per-tile sort fixes it): bit-identical lists.
#ifndef BIN_BIG_TILES
#define BIN_BIG_TILES 64
#endif

// Internal tiles may be WIDER than gsplat's 16x16 (round 3: one wave rasterises a 32x16 tile, so a splat costs one list entry, one
// LDS record and -- in the backward -- one 64-lane reduction and one flush per 32x16 tile instead of per 16x16 tile).  A wide tile
// lists exactly the Gaussians that gsplat lists for at least one of the 16x16 tiles inside it: the 16-pixel tile range is computed
// as upstream does and then divided by the width / height factor (sx, sy = log2 of it).
struct WideGrid { int tile_w16, tile_h16, sx, sy, wide_w, wide_h; };
static inline WideGrid wide_grid(int width, int height, int tile_px_w, int tile_px_h) {
    WideGrid g;
    g.tile_w16 = (width + 15) / 16; g.tile_h16 = (height + 15) / 16;
    g.sx = tile_px_w == 32 ? 1 : 0; g.sy = tile_px_h == 32 ? 1 : 0;
    g.wide_w = (g.tile_w16 + (1 << g.sx) - 1) >> g.sx; g.wide_h = (g.tile_h16 + (1 << g.sy) - 1) >> g.sy;
    return g;
}
__device__ __forceinline__ void tile_range_wide(float mx, float my, float rx, float ry, const WideGrid& g, int& x0, int& x1, int& y0, int& y1)
{
    tile_range_bin(mx, my, rx, ry, g.tile_w16, g.tile_h16, x0, x1, y0, y1);
    const int ax = (1 << g.sx) - 1, ay = (1 << g.sy) - 1;
    x0 >>= g.sx; x1 = (x1 + ax) >> g.sx; y0 >>= g.sy; y1 = (y1 + ay) >> g.sy;
}

__global__ __launch_bounds__(BIN_THREADS) void bin_count_kernel(const int32_t* __restrict__ tiles_per_gauss, const float* __restrict__ rec, int N,
                                                        WideGrid grid, uint32_t* __restrict__ table /* [BIN_SLICES][n_tiles] */)
{
    extern __shared__ uint32_t hist[];
    const int tile_w = grid.wide_w;
    const int n_tiles = grid.wide_w * grid.wide_h;
    for (int t = threadIdx.x; t < n_tiles; t += BIN_THREADS) hist[t] = 0u;
    __syncthreads();
    const int chunk = (int)ceil_div(N, BIN_SLICES);
    const int g0 = blockIdx.x * chunk, g1 = min(N, g0 + chunk);
    const int lane = threadIdx.x & 63;
    for (int base = g0; base < g1; base += BIN_THREADS) { // uniform trip count: the ballot below needs the whole wave
        const int g = base + (int)threadIdx.x;
        int x0 = 0, x1 = 0, y0 = 0, y1 = 0;
        if (g < g1 && tiles_per_gauss[g] != 0) {
            const float4* r4 = reinterpret_cast<const float4*>(rec) + 3 * (int64_t)g;
            const float4 a = r4[0], b = r4[1];
            tile_range_wide(a.x, a.y, a.w, b.w, grid, x0, x1, y0, y1);
        }
        const bool big = (x1 - x0) * (y1 - y0) > BIN_BIG_TILES;
        if (!big)
            for (int ty = y0; ty < y1; ++ty)
                for (int tx = x0; tx < x1; ++tx) atomicAdd(&hist[ty * tile_w + tx], 1u);
        unsigned long long bm = __ballot(big);
        while (bm) { // a large rectangle: walked by the wave
            const int src = __builtin_ctzll(bm);
            bm &= bm - 1;
            const int bx0 = __builtin_amdgcn_readlane(x0, src), bx1 = __builtin_amdgcn_readlane(x1, src);
            const int by0 = __builtin_amdgcn_readlane(y0, src), by1 = __builtin_amdgcn_readlane(y1, src);
            const int w = bx1 - bx0, total = w * (by1 - by0);
            for (int i = lane; i < total; i += 64) {
                const int r = i / w;
                atomicAdd(&hist[(by0 + r) * tile_w + bx0 + (i - r * w)], 1u);
            }
        }
    }
    __syncthreads();
    uint32_t* row = table + (int64_t)blockIdx.x * n_tiles;
    for (int t = threadIdx.x; t < n_tiles; t += BIN_THREADS) row[t] = hist[t];
}

// Exclusive scan down the slices (in place) + tile totals.  A workgroup owns 64 tiles x 16 groups of BIN_SLICES / 16 slices: every
// thread loads its group's counts back to back (the first version walked all 256 slices per thread with 32 workgroups in flight: 39 us
// of dependent latency), the 16 group sums are scanned through LDS, then the prefixes are written.
#define BIN_CS_PARTS 16
__global__ __launch_bounds__(1024) void bin_colscan_kernel(uint32_t* __restrict__ table, int n_tiles, uint32_t* __restrict__ tile_count)
{
    constexpr int PER = BIN_SLICES / BIN_CS_PARTS;
    __shared__ uint32_t part[BIN_CS_PARTS][64];
    const int tl = threadIdx.x & 63, p = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + tl;
    uint32_t c[PER];
    uint32_t sum = 0;
    uint32_t* col = table + (int64_t)p * PER * n_tiles + t;
    if (t < n_tiles) {
#pragma unroll
        for (int i = 0; i < PER; ++i) c[i] = col[(int64_t)i * n_tiles];
#pragma unroll
        for (int i = 0; i < PER; ++i) sum += c[i];
    }
    part[p][tl] = sum;
    __syncthreads();
    if (t >= n_tiles) return;
    uint32_t run = 0;
    for (int q = 0; q < p; ++q) run += part[q][tl];
#pragma unroll
    for (int i = 0; i < PER; ++i) { col[(int64_t)i * n_tiles] = run; run += c[i]; }
    if (p == BIN_CS_PARTS - 1) tile_count[t] = run;
}

// one workgroup: offsets[t] = exclusive scan of tile_count; stats[0] = n_isects, stats[1] = largest tile (int64, device).
// Every thread owns a contiguous run of tiles (two passes over it) and the 1024 run sums are scanned once.
__global__ __launch_bounds__(1024) void bin_tilescan_kernel(const uint32_t* __restrict__ tile_count, int n_tiles, int32_t* __restrict__ offsets,
                                                            int64_t* __restrict__ stats)
{
    __shared__ uint32_t wsum[16], wmax[16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int per = (int)ceil_div(n_tiles, 1024);
    const int i0 = min(n_tiles, (int)threadIdx.x * per), i1 = min(n_tiles, i0 + per);
    uint32_t sum = 0, mx = 0;
    for (int i = i0; i < i1; ++i) { const uint32_t v = tile_count[i]; sum += v; mx = max(mx, v); }
    uint32_t s = sum, m = mx;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(s, o, 64); if (lane >= o) s += u; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    if (lane == 63) { wsum[wv] = s; wmax[wv] = m; }
    __syncthreads();
    uint32_t run = s - sum;
    for (int w = 0; w < wv; ++w) run += wsum[w];
    for (int i = i0; i < i1; ++i) { offsets[i] = (int32_t)run; run += tile_count[i]; }
    if (threadIdx.x == 1023) {
        uint32_t mm = 0;
        for (int w = 0; w < 16; ++w) mm = max(mm, wmax[w]);
        stats[0] = (int64_t)run; stats[1] = (int64_t)mm;
    }
}

__global__ __launch_bounds__(BIN_THREADS) void bin_scatter_kernel(const int32_t* __restrict__ tiles_per_gauss, const float* __restrict__ rec,
                                                          const uint32_t* __restrict__ depth_keys, int N, WideGrid grid,
                                                          const uint32_t* __restrict__ table, const int32_t* __restrict__ offsets,
                                                          int64_t capacity, unsigned long long* __restrict__ pairs)
{
    extern __shared__ uint32_t cur[];
    const int tile_w = grid.wide_w;
    const int n_tiles = grid.wide_w * grid.wide_h;
    const uint32_t* row = table + (int64_t)blockIdx.x * n_tiles;
    for (int t = threadIdx.x; t < n_tiles; t += BIN_THREADS) cur[t] = (uint32_t)offsets[t] + row[t];
    __syncthreads();
    const int chunk = (int)ceil_div(N, BIN_SLICES);
    const int g0 = blockIdx.x * chunk, g1 = min(N, g0 + chunk);
#ifdef BIN_LAB
    uint32_t lab_acc = 0; int lab_k = 0; (void)lab_acc; (void)lab_k;
#endif
    const int lane = threadIdx.x & 63;
    for (int base = g0; base < g1; base += BIN_THREADS) { // uniform trip count: the ballot below needs the whole wave
        const int g = base + (int)threadIdx.x;
        int x0 = 0, x1 = 0, y0 = 0, y1 = 0;
        uint32_t dkey = 0u;
        if (g < g1 && tiles_per_gauss[g] != 0) {
            const float4* r4 = reinterpret_cast<const float4*>(rec) + 3 * (int64_t)g;
            const float4 a = r4[0], b = r4[1];
            tile_range_wide(a.x, a.y, a.w, b.w, grid, x0, x1, y0, y1);
            dkey = depth_keys[g];
        }
        const unsigned long long key = ((unsigned long long)dkey << 32) | (unsigned long long)(uint32_t)g;
#if defined(BIN_LAB)
        const bool big = false;
#else
        const bool big = (x1 - x0) * (y1 - y0) > BIN_BIG_TILES; // the same test as bin_count_kernel's (any split gives the same lists)
#endif
        if (!big)
            for (int ty = y0; ty < y1; ++ty)
                for (int tx = x0; tx < x1; ++tx) {
                    const uint32_t pos = atomicAdd(&cur[ty * tile_w + tx], 1u);
#if defined(BIN_LAB) && BIN_LAB == 1   // lab only: atomics without the scattered stores
                    lab_acc += pos;
#elif defined(BIN_LAB) && BIN_LAB == 2 // lab only: the same number of 8-byte stores, to consecutive addresses per thread
                    if ((int64_t)pos < capacity) pairs[((int64_t)g * 4 + (lab_k++ & 3)) % capacity] = key + pos;
#else
                    if ((int64_t)pos < capacity) pairs[pos] = key;
#endif
                }
        unsigned long long bm = __ballot(big);
        while (bm) { // a large rectangle: walked by the wave
            const int src = __builtin_ctzll(bm);
            bm &= bm - 1;
            const int bx0 = __builtin_amdgcn_readlane(x0, src), bx1 = __builtin_amdgcn_readlane(x1, src);
            const int by0 = __builtin_amdgcn_readlane(y0, src), by1 = __builtin_amdgcn_readlane(y1, src);
            const unsigned long long bkey = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)dkey, src) << 32) |
                                            (unsigned long long)(uint32_t)__builtin_amdgcn_readlane(g, src);
            const int w = bx1 - bx0, total = w * (by1 - by0);
            for (int i = lane; i < total; i += 64) {
                const int r = i / w;
                const uint32_t pos = atomicAdd(&cur[(by0 + r) * tile_w + bx0 + (i - r * w)], 1u);
                if ((int64_t)pos < capacity) pairs[pos] = bkey;
            }
        }
    }
#if defined(BIN_LAB) && BIN_LAB == 1
    if (lab_acc == 0xFFFFFFFFu) pairs[0] = lab_acc;
#endif
}

// ---- the same sort for lists of up to 1024 entries (all of them at the benchmark densities): ONE WAVE per tile, keys in REGISTERS ----
// The LDS version above pays a workgroup barrier per compare-exchange stage (45 for 512 keys): 74 us for the 8 100 tiles of a 1080p
// frame.  Here lane l holds keys [l E, (l + 1) E) of a bitonic network over 64 E slots: compare distances below E never leave the lane
// (plain register compare-exchanges), distances of E .. 32 E are an exchange with lane l ^ d -- quad-perm DPP for d = 1, 2, ds_swizzle for
// d = 4, 8 (no address register, no memory), v_permlane16/32_swap for d = 16, 32.  No LDS, no barriers, 4 tiles per workgroup.
template <int D> __device__ __forceinline__ uint32_t lane_xor(uint32_t x, int lane);
template <> __device__ __forceinline__ uint32_t lane_xor<1>(uint32_t x, int) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, false); }
template <> __device__ __forceinline__ uint32_t lane_xor<2>(uint32_t x, int) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, false); }
template <> __device__ __forceinline__ uint32_t lane_xor<4>(uint32_t x, int) { return (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, (4 << 10) | 0x1f); }
template <> __device__ __forceinline__ uint32_t lane_xor<8>(uint32_t x, int) { return (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, (8 << 10) | 0x1f); }
template <> __device__ __forceinline__ uint32_t lane_xor<16>(uint32_t x, int lane) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 sw = __builtin_amdgcn_permlane16_swap(x, x, false, false); // x' odd rows <- y even rows; y' even rows <- x odd rows
    return (lane & 16) ? sw.x : sw.y;
}
template <> __device__ __forceinline__ uint32_t lane_xor<32>(uint32_t x, int lane) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 sw = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return (lane & 32) ? sw.x : sw.y;
}

template <int E, int K, int J>
__device__ __forceinline__ void bitonic_substage(unsigned long long (&v)[E], int lane)
{
    constexpr int M = 64 * E;
    if constexpr (J >= E) {
        constexpr int D = J / E;
        const bool lower = (lane & D) == 0;
        const bool asc = (K >= M) ? true : ((lane & (K / E)) == 0);
        const bool keep_min = lower == asc;
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const uint32_t plo = lane_xor<D>((uint32_t)v[r], lane), phi = lane_xor<D>((uint32_t)(v[r] >> 32), lane);
            const unsigned long long p = ((unsigned long long)phi << 32) | plo;
            const bool lt = v[r] < p;
            v[r] = (lt != keep_min) ? p : v[r];
        }
    } else {
#pragma unroll
        for (int r = 0; r < E; ++r) {
            if ((r & J) != 0) continue;
            const unsigned long long a = v[r], b = v[r | J];
            const bool asc = (K < E) ? ((r & K) == 0) : ((K >= M) ? true : ((lane & (K / E)) == 0));
            const bool sw = (a > b) == asc;
            v[r] = sw ? b : a;
            v[r | J] = sw ? a : b;
        }
    }
}
template <int E, int K, int J> struct BitonicSub {
    static __device__ __forceinline__ void run(unsigned long long (&v)[E], int lane) {
        bitonic_substage<E, K, J>(v, lane);
        if constexpr (J > 1) BitonicSub<E, K, J / 2>::run(v, lane);
    }
};
template <int E, int K> struct BitonicStage {
    static __device__ __forceinline__ void run(unsigned long long (&v)[E], int lane) {
        BitonicSub<E, K, K / 2>::run(v, lane);
        if constexpr (K < 64 * E) BitonicStage<E, 2 * K>::run(v, lane);
    }
};

template <int E>
__device__ __forceinline__ void wave_sort_tile(const unsigned long long* __restrict__ pairs, int64_t s, int n, int lane, uint32_t t,
                                               int32_t* __restrict__ flatten_ids, uint32_t* __restrict__ tile_ids)
{
    unsigned long long v[E];
#pragma unroll
    for (int r = 0; r < E; ++r) { const int i = lane * E + r; v[r] = i < n ? pairs[s + i] : ~0ull; }
    BitonicStage<E, 2>::run(v, lane);
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int i = lane * E + r;
        if (i < n) { flatten_ids[s + i] = (int32_t)(uint32_t)v[r]; if (tile_ids) tile_ids[s + i] = t; }
    }
}

#define BIN_SORT_WAVE 1024
__global__ __launch_bounds__(256) void bin_tile_sort_wave_kernel(const unsigned long long* __restrict__ pairs, const int32_t* __restrict__ offsets,
                                                                 int n_tiles, int64_t n_isects, int32_t* __restrict__ flatten_ids,
                                                                 uint32_t* __restrict__ tile_ids)
{
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= n_tiles) return;
    const int64_t s = offsets[t];
    const int64_t e = (t == n_tiles - 1) ? n_isects : (int64_t)offsets[t + 1];
    const int n = (int)(e - s);
    if (n <= 0 || n > BIN_SORT_WAVE) return;
    if (n <= 128) wave_sort_tile<2>(pairs, s, n, lane, (uint32_t)t, flatten_ids, tile_ids);
    else if (n <= 256) wave_sort_tile<4>(pairs, s, n, lane, (uint32_t)t, flatten_ids, tile_ids);
    else if (n <= 512) wave_sort_tile<8>(pairs, s, n, lane, (uint32_t)t, flatten_ids, tile_ids);
    else wave_sort_tile<16>(pairs, s, n, lane, (uint32_t)t, flatten_ids, tile_ids);
}

// ---- lists of 1025 .. 8192 entries, first choice (round 3): split the list into 16 KEY-ORDERED buckets, two per wave, each sorted in
// registers.  The merge below pads such a list to the next power of two (4 700 keys -> 8 192 slots), sorts every 1 024-slot chunk and
// then pays three more merge stages with LDS exchanges: ~100 compare-exchange stages per wave, 96 us for the 768 tiles of the 1 M /
// 512x384 frame -- VALU-bound (61 M wave instructions).  Buckets that partition the KEY RANGE make the buckets' sorted runs concatenate
// into the sorted list with no merge at all (a sample sort): wave 0 sorts a strided sample of 256 keys in registers, every 16th of them
// is a splitter, and a key's bucket is the number of splitters not above it -- balanced whatever the depth distribution is
// (equal-width depth buckets were tried first: a cloud that fills a frustum has twice the tile's average in its farthest eighth, and
// every tile of the bench frame overflowed).  What goes to LDS is the key's POSITION in the tile's segment (4 B, 16 regions of 1 000:
// 64 KB hold any list up to 8 192 keys whose buckets stay within twice their mean); slots are handed out per wave -- one ballot per
// bucket, lanes 0 .. 15 add the counts to the buckets' counters with ONE conflict-free LDS atomic per key slot, a lane's position is
// the returned base plus its rank among the wave's lanes of that bucket (one returning atomic per key, 64 lanes on 8 addresses,
// serialises: 0.10 -> 0.16 ms for the frame).  Wave w then sorts buckets 2 w and 2 w + 1 with the smallest network that holds each
// (128 .. 1 024 slots) and writes them behind the buckets before them.  A tile with a bucket over its region writes
// BIN_BUCKET_GAVE_UP into its first output slot and is left to the merge kernel, which runs after this one on exactly those tiles.
// Unique keys => the same list, bit for bit.
#define BIN_BUCKETS 16               // at most; lists of up to BIN_BUCKET_FEW_N keys take 8 (one per wave: fewer ballots, one network per wave)
#define BIN_BUCKET_FEW_N 3584
#define BIN_BUCKET_CAP 1000          // 16 x 1000 positions x 4 B + splitters + counters stay below the 64 KB a workgroup may declare statically
#define BIN_BUCKET_GAVE_UP (-1)
template <int E>
__device__ __forceinline__ void wave_sort_bucket(const unsigned long long* __restrict__ seg, const uint32_t* __restrict__ region, int n, int64_t out,
                                                 int lane, uint32_t t, int32_t* __restrict__ flatten_ids, uint32_t* __restrict__ tile_ids)
{
    unsigned long long v[E];
#pragma unroll
    for (int r = 0; r < E; ++r) { const int i = lane * E + r; v[r] = i < n ? seg[region[i]] : ~0ull; }
    BitonicStage<E, 2>::run(v, lane);
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int i = lane * E + r;
        if (i < n) { flatten_ids[out + i] = (int32_t)(uint32_t)v[r]; if (tile_ids) tile_ids[out + i] = t; }
    }
}

template <int NB>
__device__ __forceinline__ void sort_tile_by_buckets(const unsigned long long* __restrict__ seg, int n, int64_t s, int t, uint32_t* sk,
                                                     unsigned long long* split, uint32_t* cnt, int32_t* __restrict__ flatten_ids,
                                                     uint32_t* __restrict__ tile_ids)
{
    const int tid = threadIdx.x, c = tid >> 6, lane = tid & 63;
    if (tid < NB) cnt[tid] = 0u;
    constexpr int PER = BIN_SORT_BIG / 512;
    unsigned long long k[PER];
#pragma unroll
    for (int r = 0; r < PER; ++r) { const int i = r * 512 + tid; k[r] = i < n ? seg[i] : ~0ull; }
    if (c == 0) { // splitters: 256 keys taken at equal strides through the list, sorted by this wave; every 16th one splits
        unsigned long long v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = seg[(int)(((int64_t)(lane * 4 + r) * n) >> 8)];
        BitonicStage<4, 2>::run(v, lane);
        constexpr int EVERY = 64 / NB; // lanes per splitter: sorted position 4 lane = (256 / NB) (lane / EVERY)
        if ((lane % EVERY) == 0 && lane > 0) split[lane / EVERY] = v[0];
    }
    __syncthreads();
    bool lost = false;
#pragma unroll
    for (int r = 0; r < PER; ++r) {
        const int i = r * 512 + tid;
        if (r * 512 + (c << 6) >= n) break; // wave-uniform: nothing of this wave's slot r is inside the list
        const bool have = i < n;
        int b = 0; // number of splitters <= key, by bisection over the sorted splitters (LDS broadcast-ish reads)
#pragma unroll
        for (int step = NB / 2; step >= 1; step >>= 1) b += (k[r] >= split[b + step]) ? step : 0;
        uint32_t mine = 0u, rank = 0u;
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const unsigned long long m = __ballot(have && b == q);
            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            rank = (b == q) ? below : rank;
            mine = (lane == q) ? (uint32_t)__popcll(m) : mine;
        }
        uint32_t base = 0u;
        if (lane < NB && mine) base = atomicAdd(&cnt[lane], mine);
        const uint32_t pos = (uint32_t)__shfl((int)base, b, 64) + rank;
        if (have) { if (pos < BIN_BUCKET_CAP) sk[b * BIN_BUCKET_CAP + pos] = (uint32_t)i; else lost = true; }
    }
    if (__syncthreads_or(lost)) { // a bucket is over its region: the merge kernel takes this tile
        if (tid == 0) flatten_ids[s] = BIN_BUCKET_GAVE_UP;
        return;
    }
    constexpr int PW = NB / 8; // buckets per wave
    int before = 0;
    for (int b = 0; b < PW * c; ++b) before += (int)cnt[b];
#pragma unroll
    for (int h = 0; h < PW; ++h) {
        const int bk = PW * c + h;
        const int nb = (int)cnt[bk];
        const uint32_t* region = sk + bk * BIN_BUCKET_CAP;
        if (nb <= 128) wave_sort_bucket<2>(seg, region, nb, s + before, lane, (uint32_t)t, flatten_ids, tile_ids);
        else if (nb <= 256) wave_sort_bucket<4>(seg, region, nb, s + before, lane, (uint32_t)t, flatten_ids, tile_ids);
        else if (nb <= 512) wave_sort_bucket<8>(seg, region, nb, s + before, lane, (uint32_t)t, flatten_ids, tile_ids);
        else wave_sort_bucket<16>(seg, region, nb, s + before, lane, (uint32_t)t, flatten_ids, tile_ids);
        before += nb;
    }
}

__global__ __launch_bounds__(512) void bin_tile_sort_bucket_kernel(const unsigned long long* __restrict__ pairs, const int32_t* __restrict__ offsets,
                                                                   int n_tiles, int64_t n_isects, int32_t* __restrict__ flatten_ids,
                                                                   uint32_t* __restrict__ tile_ids, int max_n)
{
    __shared__ uint32_t sk[BIN_BUCKETS * BIN_BUCKET_CAP];
    __shared__ unsigned long long split[BIN_BUCKETS];   // split[1 .. NB - 1]: bucket b holds the keys in [split[b], split[b + 1])
    __shared__ uint32_t cnt[BIN_BUCKETS];
    const int t = blockIdx.x;
    const int64_t s = offsets[t];
    const int64_t e = (t == n_tiles - 1) ? n_isects : (int64_t)offsets[t + 1];
    const int n = (int)(e - s);
    if (n <= BIN_SORT_WAVE || n > BIN_SORT_BIG) return; // workgroup-uniform
    if (n > max_n) { if (threadIdx.x == 0) flatten_ids[s] = BIN_BUCKET_GAVE_UP; return; }
    if (n <= BIN_BUCKET_FEW_N) sort_tile_by_buckets<8>(pairs + s, n, s, t, sk, split, cnt, flatten_ids, tile_ids);
    else sort_tile_by_buckets<16>(pairs + s, n, s, t, sk, split, cnt, flatten_ids, tile_ids);
}

// ---- lists of 1025 .. 8192 entries (1 M Gaussians on a 512x384 frame: every tile): the SAME register network, 1024 keys per wave,
// up to 8 waves per tile.  Wave c sorts chunk c in registers -- ascending for even c, descending for odd c (the ascending network run
// on the bitwise complements), i.e. exactly the state a 64 x 16 x C bitonic network is in after its K = 1024 stage; the remaining
// stages K = 2048 .. m then need, per stage, log2(K / 1024) compare-exchanges BETWEEN chunks (same position, partner chunk c ^ J /
// 1024: one round trip through LDS each) followed by the in-register merge J = 512 .. 1.  6 LDS exchanges + 12 workgroup barriers for
// 8192 keys, where a plain LDS bitonic sort (256 threads, one barrier per compare-exchange stage: 91 of them) took 332 us for the
// 768 tiles of that frame -- 20 % of the mapper step.
__global__ __launch_bounds__(512) void bin_tile_sort_merge_kernel(const unsigned long long* __restrict__ pairs, const int32_t* __restrict__ offsets,
                                                                  int n_tiles, int64_t n_isects, int32_t* __restrict__ flatten_ids,
                                                                  uint32_t* __restrict__ tile_ids, int only_flagged)
{
    constexpr int E = 16, CH = 64 * E; // keys per lane, keys per wave
    __shared__ unsigned long long sk[BIN_SORT_BIG];
    const int t = blockIdx.x;
    const int64_t s = offsets[t];
    const int64_t e = (t == n_tiles - 1) ? n_isects : (int64_t)offsets[t + 1];
    const int n = (int)(e - s);
    if (n <= BIN_SORT_WAVE || n > BIN_SORT_BIG) return; // workgroup-uniform
    if (only_flagged && flatten_ids[s] != BIN_BUCKET_GAVE_UP) return; // bin_tile_sort_bucket_kernel sorted this tile (ids are >= 0)
    int m = 2 * CH;
    while (m < n) m <<= 1;
    const int c = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool active = c * CH < m;
    unsigned long long v[E];
    if (active) {
#pragma unroll
        for (int r = 0; r < E; ++r) { const int i = c * CH + lane * E + r; v[r] = i < n ? pairs[s + i] : ~0ull; }
        const unsigned long long flip = (c & 1) ? ~0ull : 0ull;
#pragma unroll
        for (int r = 0; r < E; ++r) v[r] ^= flip;
        BitonicStage<E, 2>::run(v, lane);
#pragma unroll
        for (int r = 0; r < E; ++r) v[r] ^= flip;
    }
    for (int K = 2 * CH; K <= m; K <<= 1) {
        const bool asc = (K == m) || (((c * CH) & K) == 0);
        for (int J = K >> 1; J >= CH; J >>= 1) {
            const int dc = J / CH;
            if (active) {
#pragma unroll
                for (int r = 0; r < E; ++r) sk[c * CH + r * 64 + lane] = v[r];
            }
            __syncthreads();
            if (active) {
                const bool keep_min = ((c & dc) == 0) == asc;
#pragma unroll
                for (int r = 0; r < E; ++r) {
                    const unsigned long long p = sk[(c ^ dc) * CH + r * 64 + lane];
                    v[r] = ((v[r] < p) != keep_min) ? p : v[r];
                }
            }
            __syncthreads();
        }
        if (active) {
            const unsigned long long flip = asc ? 0ull : ~0ull;
#pragma unroll
            for (int r = 0; r < E; ++r) v[r] ^= flip;
            BitonicSub<E, CH, CH / 2>::run(v, lane); // K >= 64 E: every compare-exchange ascending
#pragma unroll
            for (int r = 0; r < E; ++r) v[r] ^= flip;
        }
    }
    if (active) {
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const int i = c * CH + lane * E + r;
            if (i < n) { flatten_ids[s + i] = (int32_t)(uint32_t)v[r]; if (tile_ids) tile_ids[s + i] = (uint32_t)t; }
        }
    }
}

// ---- lists ABOVE 8192 entries (round 5): no cliff.  gsplat's radix sort has no list-length limit (h3dgsv3.py:664-680 -> isect_tiles + sort),
// and a trained map seen from close by, or any frame whose splats are several pixels wide, puts tens of thousands of entries on a tile; until
// round 4 ONE such tile sent the whole frame to the global radix route and the step back to the per-stage Python chain.  Here one workgroup
// per long tile sorts its segment by RECURSIVE MSD PARTITION ON THE KEY RANGE, ping-ponging between `pairs` and a scratch buffer of the
// same size: a segment of more than 1 024 keys finds its smallest / largest 64-bit key, picks NB = the power of two >= count / 512 buckets
// (4 .. 1 024) that split [min, max] evenly -- bucket = (key - min) >> shift --, counts (LDS atomics), scans, and scatters every key to its
// bucket's region in the OTHER buffer; buckets of <= 1 024 keys are then sorted by one wave each in registers (the same bitonic networks as
// every other list) straight into `flatten_ids`, larger ones go on an LDS stack and are partitioned again.  Deterministic termination: the
// keys are unique, so max > min, the shift leaves at least log2(NB) - 1 significant bits of the range and the buckets of `min` and `max`
// differ -- every child is strictly smaller than its parent, whatever the depth distribution (equal depths included: the ids split them).
// Unique keys also make the result THE (tile, depth, id) order, bit for bit.  Writes by one wave are read by others of the same workgroup
// through global memory, across a workgroup barrier (workgroup-scope release / acquire: same CU, same vector L1).  Pending segments hold > 1 024 keys each, so a list of n keys never has more
// than n / 1 024 of them on the stack: 4 096 slots = lists of up to 4 194 304 entries (BIN_SORT_LONG_MAX; the host knows the fullest tile).
#define BIN_LONG_NB 1024
#define BIN_LONG_REG 32              // keys a thread holds in registers while its workgroup partitions a segment of up to 16 384 keys
#define BIN_LONG_STACK 4096
#define BIN_SORT_LONG_MAX (BIN_LONG_STACK * 1024)

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned long long u = ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)v, o, 64);
        v = u < v ? u : v;
    }
    return v;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned long long u = ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)v, o, 64);
        v = u > v ? u : v;
    }
    return v;
}

__device__ __forceinline__ void wave_sort_any(const unsigned long long* buf, int64_t abs, int n, int lane, uint32_t t,
                                              int32_t* __restrict__ flatten_ids, uint32_t* __restrict__ tile_ids)
{
    if (n <= 128) wave_sort_tile<2>(buf, abs, n, lane, t, flatten_ids, tile_ids);
    else if (n <= 256) wave_sort_tile<4>(buf, abs, n, lane, t, flatten_ids, tile_ids);
    else if (n <= 512) wave_sort_tile<8>(buf, abs, n, lane, t, flatten_ids, tile_ids);
    else wave_sort_tile<16>(buf, abs, n, lane, t, flatten_ids, tile_ids);
}

#if defined(__HIP_DEVICE_COMPILE__) && !(defined(__gfx950__) && defined(__AMDGCN_CUMODE__))
#error "bin_tile_sort_long_kernel hands keys between the waves of a workgroup through global memory under workgroup-scope fences: built and measured for gfx950 in CU mode"
#endif
__global__ __launch_bounds__(512) void bin_tile_sort_long_kernel(unsigned long long* pairs, unsigned long long* scratch, const int32_t* __restrict__ offsets,
                                                                 int n_tiles, int64_t n_isects, int32_t* __restrict__ flatten_ids,
                                                                 uint32_t* __restrict__ tile_ids)
{
    __shared__ uint32_t cnt[BIN_LONG_NB];          // bucket counts, then the scatter's cursors
    __shared__ uint32_t off[BIN_LONG_NB + 1];      // bucket starts inside the segment
    __shared__ uint32_t stk_start[BIN_LONG_STACK]; // pending segments: start inside the tile's list ...
    __shared__ uint32_t stk_count[BIN_LONG_STACK]; // ... and count | buffer << 31 (0: pairs, 1: scratch)
    __shared__ unsigned long long red_min[8], red_max[8];
    __shared__ uint32_t wsum[8];
    __shared__ int sp;
    const int t = blockIdx.x;
    const int64_t s = offsets[t];
    const int64_t e = (t == n_tiles - 1) ? n_isects : (int64_t)offsets[t + 1];
    const int64_t n64 = e - s;
    if (n64 <= BIN_SORT_BIG || n64 > BIN_SORT_LONG_MAX) return; // workgroup-uniform; shorter lists belong to the other kernels
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    if (tid == 0) { stk_start[0] = 0u; stk_count[0] = (uint32_t)n64; sp = 1; }
    __syncthreads();
    while (true) {
        const int top = sp;                          // uniform: read between barriers
        if (top == 0) break;
        const uint32_t seg0 = stk_start[top - 1], cw = stk_count[top - 1];
        __syncthreads();                             // everyone has read the entry before it is overwritten
        if (tid == 0) sp = top - 1;
        const int count = (int)(cw & 0x7FFFFFFFu);
        const unsigned long long* src = ((cw >> 31) ? scratch : pairs) + s + seg0;
        unsigned long long* const dst_base = (cw >> 31) ? pairs : scratch;
        unsigned long long* dst = dst_base + s + seg0;
        // -- key range of the segment.  Segments of up to 512 x BIN_LONG_REG keys (every first level but a pathological one) are read ONCE, all
        // loads in flight together, and the three sweeps below run on registers; longer ones sweep the source three times.
        constexpr int REG = BIN_LONG_REG;
        const bool in_regs = count <= 512 * REG;
        unsigned long long kreg[REG];
        unsigned long long lo = ~0ull, hi = 0ull;
        if (in_regs) {
#pragma unroll
            for (int r = 0; r < REG; ++r) { const int i = r * 512 + tid; kreg[r] = i < count ? src[i] : ~0ull; }
#pragma unroll
            for (int r = 0; r < REG; ++r) {
                const unsigned long long k = kreg[r];
                lo = k < lo ? k : lo;                               // the padding ~0 never wins a minimum against a real key
                hi = (r * 512 + tid < count && k > hi) ? k : hi;
            }
        } else {
            for (int i = tid; i < count; i += 512) { const unsigned long long k = src[i]; lo = k < lo ? k : lo; hi = k > hi ? k : hi; }
        }
        lo = wave_min_u64(lo); hi = wave_max_u64(hi);
        if (lane == 0) { red_min[wv] = lo; red_max[wv] = hi; }
        int nb_log = 2;
        while ((1 << nb_log) < BIN_LONG_NB && (512 << nb_log) < count) ++nb_log;
        const int NB = 1 << nb_log;
        for (int b = tid; b < NB; b += 512) cnt[b] = 0u;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 8; ++w) { lo = red_min[w] < lo ? red_min[w] : lo; hi = red_max[w] > hi ? red_max[w] : hi; }
        const unsigned long long range = hi - lo;    // >= 1: the keys are unique and count > 1
        const int bits = 64 - __clzll((long long)range);
        const int shift = bits > nb_log ? bits - nb_log : 0;
        // -- count, scan
        if (in_regs) {
#pragma unroll
            for (int r = 0; r < REG; ++r)
                if (r * 512 + tid < count) atomicAdd(&cnt[(uint32_t)((kreg[r] - lo) >> shift)], 1u);
        } else {
            for (int i = tid; i < count; i += 512) atomicAdd(&cnt[(uint32_t)((src[i] - lo) >> shift)], 1u);
        }
        __syncthreads();
        {   // exclusive scan of cnt[0 .. NB): thread tid owns entries 2 tid, 2 tid + 1 (NB <= 1024 = 2 x 512)
            const uint32_t a = 2 * tid < NB ? cnt[2 * tid] : 0u, b = 2 * tid + 1 < NB ? cnt[2 * tid + 1] : 0u;
            uint32_t x = a + b;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)x, o, 64); if (lane >= o) x += u; }
            if (lane == 63) wsum[wv] = x;
            __syncthreads();
            uint32_t base = x - (a + b);
            for (int w = 0; w < wv; ++w) base += wsum[w];
            if (2 * tid < NB) { off[2 * tid] = base; cnt[2 * tid] = base; }
            if (2 * tid + 1 < NB) { off[2 * tid + 1] = base + a; cnt[2 * tid + 1] = base + a; }
            if (tid == 0) off[NB] = (uint32_t)count;
        }
        __syncthreads();
        // -- scatter to the other buffer (order inside a bucket is arbitrary: it is sorted or partitioned next)
        if (in_regs) {
#pragma unroll
            for (int r = 0; r < REG; ++r)
                if (r * 512 + tid < count) dst[atomicAdd(&cnt[(uint32_t)((kreg[r] - lo) >> shift)], 1u)] = kreg[r];
        } else {
            for (int i = tid; i < count; i += 512) {
                const unsigned long long k = src[i];
                dst[atomicAdd(&cnt[(uint32_t)((k - lo) >> shift)], 1u)] = k;
            }
        }
        // the scattered keys are read back by OTHER waves of this workgroup only: workgroup-scope release / acquire, which is what
        // __syncthreads() carries (the waves of a workgroup share their CU's vector L1, so a line this workgroup rewrote is never stale
        // there).  An agent-scope fence here (`__threadfence()`) writes back and invalidates the XCD's whole L2 on every level of every
        // tile: measured 0.68 ms for the 768 lists of a dense north-star frame against 0.21 without it.
        // (this does not rest on the CU-mode L1 by accident -- HIP's __syncthreads() IS fence(release, "workgroup") + s_barrier +
        // fence(acquire, "workgroup"), and the backend legalises workgroup scope per target mode: under -mtgsplit, where a workgroup's waves
        // may sit on different CUs, it emits the L1 invalidate / write-through that scope then needs.  This library is built for gfx950 in
        // CU mode only, see the #error below.)
        __syncthreads();
        // -- children: small ones are sorted now (bucket b by wave b mod 8), large ones are pushed (thread 0, in bucket order)
        if (tid == 0) {
            int p = sp;
            for (int b = 0; b < NB; ++b) {
                const uint32_t c = off[b + 1] - off[b];
                if (c > BIN_SORT_WAVE) { stk_start[p] = seg0 + off[b]; stk_count[p] = c | ((cw >> 31) ? 0u : 0x80000000u); ++p; }
            }
            sp = p;
        }
        for (int b = wv; b < NB; b += 8) {
            const int c = (int)(off[b + 1] - off[b]);
            if (c > 0 && c <= BIN_SORT_WAVE) wave_sort_any(dst_base, s + seg0 + off[b], c, lane, (uint32_t)t, flatten_ids, tile_ids);
        }
        __syncthreads();
    }
}

} // namespace adk

static inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

// ---- stage 1: depth order of the Gaussians + scan of their tile counts --------------------------
extern "C" int64_t adk_bin_depth_workspace_bytes(int N)
{
    if (N < 0) return ADK_EINVAL;
    const int64_t n = N > 0 ? N : 1;
    return 4 * align256(n * 4) + align256(adk::radix_scratch_bytes(n)) + 256;
}

// In : depth_keys[N] (u32, 0xFFFFFFFF = culled), gauss_ids[N] (= 0..N-1), tiles_per_gauss[N].
// Out: sorted_ids[N] (Gaussian ids, front to back, ties by id), block_offs[ceil(N/256)] (exclusive
//      scan of the per-256 sums of tiles_per_gauss in that order), *n_isects (int64, device).
extern "C" int adk_bin_depth_order(int N, const uint32_t* depth_keys, const uint32_t* gauss_ids,
                                   const int32_t* tiles_per_gauss, uint32_t* sorted_ids, uint32_t* block_offs,
                                   int64_t* n_isects, void* workspace, int64_t workspace_bytes, hipStream_t stream)
{
    if (N < 0) return ADK_EINVAL;
    if (!n_isects) return ADK_EINVAL;
    if (N == 0) { return adk::clear_bytes(n_isects, sizeof(int64_t), stream); }
    if (!depth_keys || !gauss_ids || !tiles_per_gauss || !sorted_ids || !block_offs || !workspace) return ADK_EINVAL;
    if (workspace_bytes < adk_bin_depth_workspace_bytes(N) || ((uintptr_t)workspace & 255)) return ADK_EWORKSPACE;
    char* w = (char*)workspace;
    const int64_t seg = align256((int64_t)N * 4);
    uint32_t* k0 = (uint32_t*)w; uint32_t* v0 = (uint32_t*)(w + seg);
    uint32_t* k1 = (uint32_t*)(w + 2 * seg); uint32_t* v1 = (uint32_t*)(w + 3 * seg);
    uint32_t* scratch = (uint32_t*)(w + 4 * seg);
    // depth bits of a positive float are < 0x7F800000 and the cull sentinel is 0xFFFFFFFF: 32 bits, 4 passes.  An even
    // number of passes ends in the second buffer pair, so sorted_ids itself is handed in as that pair's value
    // buffer: the last scatter writes the result in place (no device-to-device copy).
    (void)v1;
    const int res = adk::radix_sort_pairs(depth_keys, gauss_ids, k0, v0, k1, sorted_ids, N, 0, 32, scratch, stream);
    if (res != 1) return ADK_EUNSUPPORTED; // 4 passes by construction
    const int nb = (int)adk::ceil_div(N, 256);
    hipLaunchKernelGGL(adk::count_block_sums_kernel, dim3(nb), dim3(256), 0, stream, sorted_ids, tiles_per_gauss, N, block_offs);
    hipLaunchKernelGGL(adk::scan_block_sums_kernel, dim3(1), dim3(1024), 0, stream, block_offs, nb, n_isects);
    ADK_RETURN_LAST_ERROR();
}

// ---- stage 2: emit in depth order, stable sort by tile, per-tile offsets -----------------------------
// *n_isects (int64, device) = sum(tiles_per_gauss).  Same value adk_bin_depth_order reports, available
// one sort earlier (integer sum: order-independent).
extern "C" int adk_bin_count_isects(int N, const int32_t* tiles_per_gauss, int64_t* n_isects, hipStream_t stream)
{
    if (N < 0 || !n_isects) return ADK_EINVAL;
    int e = adk::clear_bytes(n_isects, sizeof(int64_t), stream); // a kernel, not hipMemsetAsync: hipGraph-safe
    if (e != 0) return e;
    if (N == 0) return 0;
    if (!tiles_per_gauss) return ADK_EINVAL;
    int nb = (int)adk::ceil_div(N, 2048);
    if (nb > 256) nb = 256;
    hipLaunchKernelGGL(adk::count_isects_kernel, dim3(nb), dim3(256), 0, stream, tiles_per_gauss, N, (unsigned long long*)n_isects);
    ADK_RETURN_LAST_ERROR();
}

extern "C" int64_t adk_bin_tiles_workspace_bytes(int64_t n_isects)
{
    if (n_isects < 0) return ADK_EINVAL;
    const int64_t n = n_isects > 0 ? n_isects : 1;
    return 4 * align256(n * 4) + align256(adk::radix_scratch_bytes(n)) + 256;
}

// n_isects is the HOST copy of the value stage 1 produced (the caller sized flatten_ids/tile_ids
// with it).  Out: flatten_ids[I] (int32 Gaussian ids in (tile, depth, id) order), tile_ids[I]
// (u32 tile of each entry, same order), offsets[tile_h*tile_w].
extern "C" int adk_bin_tiles(int N, int64_t n_isects, const uint32_t* sorted_ids, const uint32_t* block_offs,
                             const int32_t* tiles_per_gauss, const float* rec, int width, int height,
                             int32_t* flatten_ids, uint32_t* tile_ids, int32_t* offsets, void* workspace,
                             int64_t workspace_bytes, hipStream_t stream)
{
    if (N < 0 || n_isects < 0 || width <= 0 || height <= 0 || !offsets) return ADK_EINVAL;
    const int tile_w = (width + 15) / 16, tile_h = (height + 15) / 16, n_tiles = tile_w * tile_h;
    if (n_isects >= (int64_t)1 << 31) return ADK_EUNSUPPORTED;
    if (n_isects == 0 || N == 0) {
        hipLaunchKernelGGL(adk::tile_offsets_kernel, dim3((unsigned)adk::ceil_div(n_tiles, 256)), dim3(256), 0, stream, nullptr, (int64_t)0, n_tiles, offsets);
        ADK_RETURN_LAST_ERROR();
    }
    if (!sorted_ids || !block_offs || !tiles_per_gauss || !rec || !flatten_ids || !tile_ids || !workspace) return ADK_EINVAL;
    if (workspace_bytes < adk_bin_tiles_workspace_bytes(n_isects) || ((uintptr_t)workspace & 255)) return ADK_EWORKSPACE;
    char* w = (char*)workspace;
    const int64_t seg = align256(n_isects * 4);
    uint32_t* k0 = (uint32_t*)w; uint32_t* v0 = (uint32_t*)(w + seg);
    uint32_t* k1 = (uint32_t*)(w + 2 * seg); uint32_t* v1 = (uint32_t*)(w + 3 * seg);
    uint32_t* scratch = (uint32_t*)(w + 4 * seg);
    const int nb = (int)adk::ceil_div(N, 256);
    hipLaunchKernelGGL(adk::emit_kernel, dim3(nb), dim3(256), 0, stream, sorted_ids, tiles_per_gauss, rec, N, tile_w, tile_h, block_offs, n_isects, k0, v0);
    int bits = 0;
    while ((1 << bits) < n_tiles) ++bits;
    const int bit_hi = ((bits + 7) / 8) * 8; // whole 8-bit digits covering the tile id
    // source = (k0,v0), read by the first pass only.  The pass count decides which buffer pair receives the last
    // scatter; the caller's (tile_ids, flatten_ids) are placed there so the sorted list is written in place.
    const int hi = bit_hi > 0 ? bit_hi : 8, passes = hi / 8;
    uint32_t* ko = tile_ids;
    uint32_t* vo = reinterpret_cast<uint32_t*>(flatten_ids);
    int res;
    if (passes & 1) res = adk::radix_sort_pairs(k0, v0, ko, vo, k1, v1, n_isects, 0, hi, scratch, stream) == 0 ? 0 : -1;
    else res = adk::radix_sort_pairs(k0, v0, k1, v1, ko, vo, n_isects, 0, hi, scratch, stream) == 1 ? 0 : -1;
    if (res != 0) return ADK_EUNSUPPORTED;
    hipLaunchKernelGGL(adk::tile_offsets_kernel, dim3((unsigned)adk::ceil_div(n_isects, 256)), dim3(256), 0, stream, tile_ids, n_isects, n_tiles, offsets);
    ADK_RETURN_LAST_ERROR();
}


// ---- tile-local route (round 2): counting sort by tile + per-tile LDS sort -----------------------------------------------
// LDS the per-slice tile histogram may use (bytes): the count / scatter kernels ask for 4 B x tiles of dynamic LDS.
#define ADK_BIN_LDS_LIMIT (128 * 1024)

// the shapes adk_raster_fwd_t / adk_raster_bwd_t consume: gsplat's 16x16 and the wide 32x16 internal tile (a list binned for any other shape has no consumer)
static inline bool tile_shape_ok(int tpw, int tph) { return (tpw == 16 || tpw == 32) && tph == 16; }

// 1 if adk_bin_local_* can handle this image size (tile histogram fits LDS), else the caller uses adk_bin_depth_order / adk_bin_tiles.
extern "C" int adk_bin_local_supported_t(int width, int height, int tile_px_w, int tile_px_h)
{
    if (width <= 0 || height <= 0 || !tile_shape_ok(tile_px_w, tile_px_h)) return 0;
    const adk::WideGrid g = adk::wide_grid(width, height, tile_px_w, tile_px_h);
    return (int64_t)g.wide_w * g.wide_h * 4 <= ADK_BIN_LDS_LIMIT ? 1 : 0;
}
extern "C" int adk_bin_local_supported(int width, int height) { return adk_bin_local_supported_t(width, height, 16, 16); }

extern "C" int64_t adk_bin_local_workspace_bytes_t(int width, int height, int tile_px_w, int tile_px_h)
{
    if (width <= 0 || height <= 0 || !tile_shape_ok(tile_px_w, tile_px_h)) return ADK_EINVAL;
    const adk::WideGrid g = adk::wide_grid(width, height, tile_px_w, tile_px_h);
    const int64_t n_tiles = (int64_t)g.wide_w * g.wide_h;
    return align256((int64_t)BIN_SLICES * n_tiles * 4) + align256(n_tiles * 4) + 256;
}
extern "C" int64_t adk_bin_local_workspace_bytes(int width, int height) { return adk_bin_local_workspace_bytes_t(width, height, 16, 16); }

// Step 1: per-tile counts.  offsets [tiles] (the isect_offset_encode output for 16x16 tiles) and stats [2] int64 (device):
// stats[0] = n_isects, stats[1] = entries of the fullest tile.  The workspace is consumed by adk_bin_local_scatter.
extern "C" int adk_bin_local_count_t(int N, const int32_t* tiles_per_gauss, const float* rec, int width, int height, int tile_px_w, int tile_px_h,
                                     int32_t* offsets, int64_t* stats, void* workspace, int64_t workspace_bytes, hipStream_t stream)
{
    using namespace adk;
    if (N < 0 || width <= 0 || height <= 0 || !offsets || !stats || !workspace) return ADK_EINVAL;
    if (!adk_bin_local_supported_t(width, height, tile_px_w, tile_px_h)) return ADK_EUNSUPPORTED;
    if (workspace_bytes < adk_bin_local_workspace_bytes_t(width, height, tile_px_w, tile_px_h) || ((uintptr_t)workspace & 255)) return ADK_EWORKSPACE;
    if (N > 0 && (!tiles_per_gauss || !rec)) return ADK_EINVAL;
    const WideGrid g = wide_grid(width, height, tile_px_w, tile_px_h);
    const int n_tiles = g.wide_w * g.wide_h;
    uint32_t* table = (uint32_t*)workspace;
    uint32_t* tile_count = (uint32_t*)((char*)workspace + align256((int64_t)BIN_SLICES * n_tiles * 4));
    const size_t lds = (size_t)n_tiles * 4;
    static bool attr_set = false; // > 64 KB of dynamic LDS needs the opt-in (idempotent; a race only repeats it)
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bin_count_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, ADK_BIN_LDS_LIMIT);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bin_scatter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, ADK_BIN_LDS_LIMIT);
        attr_set = true;
    }
    hipLaunchKernelGGL(bin_count_kernel, dim3(BIN_SLICES), dim3(BIN_THREADS), lds, stream, tiles_per_gauss, rec, N, g, table);
    hipLaunchKernelGGL(bin_colscan_kernel, dim3((unsigned)ceil_div(n_tiles, 64)), dim3(1024), 0, stream, table, n_tiles, tile_count);
    hipLaunchKernelGGL(bin_tilescan_kernel, dim3(1), dim3(1024), 0, stream, tile_count, n_tiles, offsets, stats);
    ADK_RETURN_LAST_ERROR();
}
extern "C" int adk_bin_local_count(int N, const int32_t* tiles_per_gauss, const float* rec, int width, int height, int32_t* offsets,
                                   int64_t* stats, void* workspace, int64_t workspace_bytes, hipStream_t stream)
{
    return adk_bin_local_count_t(N, tiles_per_gauss, rec, width, height, 16, 16, offsets, stats, workspace, workspace_bytes, stream);
}

extern "C" int64_t adk_bin_local_pairs_bytes(int64_t n_isects) { return n_isects < 0 ? ADK_EINVAL : align256((n_isects > 0 ? n_isects : 1) * 8); }

// Step 2: every slice scatters its (depth bits << 32 | id) keys into the tiles' segments of `pairs` (capacity entries of 8 B; entries
// beyond the capacity are dropped, so the host may launch this with an ESTIMATED capacity before it has read n_isects and repeat it in
// the rare case the estimate was too small).  Order inside a segment is arbitrary.
extern "C" int adk_bin_local_scatter_t(int N, int64_t capacity, const uint32_t* depth_keys, const int32_t* tiles_per_gauss, const float* rec,
                                       int width, int height, int tile_px_w, int tile_px_h, const int32_t* offsets, const void* workspace,
                                       int64_t workspace_bytes, void* pairs, hipStream_t stream)
{
    using namespace adk;
    if (N < 0 || capacity < 0 || width <= 0 || height <= 0 || !offsets || !workspace || !tile_shape_ok(tile_px_w, tile_px_h)) return ADK_EINVAL;
    if (N == 0 || capacity == 0) return 0;
    if (!depth_keys || !tiles_per_gauss || !rec || !pairs) return ADK_EINVAL;
    if (workspace_bytes < adk_bin_local_workspace_bytes_t(width, height, tile_px_w, tile_px_h) || ((uintptr_t)workspace & 255) || ((uintptr_t)pairs & 7)) return ADK_EWORKSPACE;
    const WideGrid g = wide_grid(width, height, tile_px_w, tile_px_h);
    const int n_tiles = g.wide_w * g.wide_h;
    hipLaunchKernelGGL(bin_scatter_kernel, dim3(BIN_SLICES), dim3(BIN_THREADS), (size_t)n_tiles * 4, stream, tiles_per_gauss, rec, depth_keys, N, g,
                       (const uint32_t*)workspace, offsets, capacity, (unsigned long long*)pairs);
    ADK_RETURN_LAST_ERROR();
}
extern "C" int adk_bin_local_scatter(int N, int64_t capacity, const uint32_t* depth_keys, const int32_t* tiles_per_gauss, const float* rec,
                                     int width, int height, const int32_t* offsets, const void* workspace, int64_t workspace_bytes,
                                     void* pairs, hipStream_t stream)
{
    return adk_bin_local_scatter_t(N, capacity, depth_keys, tiles_per_gauss, rec, width, height, 16, 16, offsets, workspace, workspace_bytes, pairs, stream);
}

// Step 3: n_isects / max_tile = the HOST copies of stats (pairs must have held all n_isects entries).  Out: flatten_ids [I] in
// (tile, depth, id) order, tile_ids [I] (or NULL).  max_tile must not exceed 8192 (otherwise: ADK_EUNSUPPORTED, use the global route).
extern "C" int adk_bin_local_sort_t(int64_t n_isects, int64_t max_tile, int width, int height, int tile_px_w, int tile_px_h, const int32_t* offsets,
                                    const void* pairs, int32_t* flatten_ids, uint32_t* tile_ids, hipStream_t stream)
{
    using namespace adk;
    if (n_isects < 0 || width <= 0 || height <= 0 || !offsets || !tile_shape_ok(tile_px_w, tile_px_h)) return ADK_EINVAL;
    if (n_isects == 0) return 0;
    if (max_tile > BIN_SORT_BIG || n_isects >= ((int64_t)1 << 31)) return ADK_EUNSUPPORTED;
    if (!pairs || !flatten_ids) return ADK_EINVAL;
    const WideGrid g = wide_grid(width, height, tile_px_w, tile_px_h);
    const int n_tiles = g.wide_w * g.wide_h;
    hipLaunchKernelGGL(bin_tile_sort_wave_kernel, dim3((unsigned)ceil_div(n_tiles, 4)), dim3(256), 0, stream, (const unsigned long long*)pairs,
                       offsets, n_tiles, n_isects, flatten_ids, tile_ids);
    if (max_tile > BIN_SORT_WAVE) {
        const char* env = getenv("ADK_BIN_BUCKET_SORT");   // "0": every long list through the merge kernel; "<n>": lists above n keys (A/B, tests)
        const int max_n = env ? min(atoi(env), BIN_SORT_BIG) : BIN_SORT_BIG;
        const int buckets = max_n > BIN_SORT_WAVE;
        if (buckets)
            hipLaunchKernelGGL(bin_tile_sort_bucket_kernel, dim3(n_tiles), dim3(512), 0, stream, (const unsigned long long*)pairs, offsets,
                               n_tiles, n_isects, flatten_ids, tile_ids, max_n);
        hipLaunchKernelGGL(bin_tile_sort_merge_kernel, dim3(n_tiles), dim3(512), 0, stream, (const unsigned long long*)pairs, offsets,
                           n_tiles, n_isects, flatten_ids, tile_ids, buckets);
    }
    ADK_RETURN_LAST_ERROR();
}
// adk_bin_local_sort_t for ANY list length up to 4 194 304 entries per tile: the same kernels for the tiles of up to 8 192 entries, and
// bin_tile_sort_long_kernel for the longer ones.  `pairs` is read AND overwritten (the long tiles ping-pong between it and `scratch`,
// adk_bin_local_pairs_bytes(capacity of pairs) bytes, 8 B aligned); the other tiles' segments are left as they were.
extern "C" int adk_bin_local_sort_long_t(int64_t n_isects, int64_t max_tile, int width, int height, int tile_px_w, int tile_px_h, const int32_t* offsets,
                                         void* pairs, void* scratch, int64_t scratch_bytes, int32_t* flatten_ids, uint32_t* tile_ids, hipStream_t stream)
{
    using namespace adk;
    if (n_isects < 0 || width <= 0 || height <= 0 || !offsets || !tile_shape_ok(tile_px_w, tile_px_h)) return ADK_EINVAL;
    if (n_isects == 0) return 0;
    if (max_tile > BIN_SORT_LONG_MAX || n_isects >= ((int64_t)1 << 31)) return ADK_EUNSUPPORTED;
    if (!pairs || !flatten_ids) return ADK_EINVAL;
    const int rc = adk_bin_local_sort_t(n_isects, max_tile < BIN_SORT_BIG ? max_tile : BIN_SORT_BIG, width, height, tile_px_w, tile_px_h, offsets, pairs,
                                        flatten_ids, tile_ids, stream);
    if (rc != 0 || max_tile <= BIN_SORT_BIG) return rc;
    if (!scratch || ((uintptr_t)scratch & 7) || scratch_bytes < n_isects * 8) return ADK_EWORKSPACE;
    const WideGrid g = wide_grid(width, height, tile_px_w, tile_px_h);
    const int n_tiles = g.wide_w * g.wide_h;
    hipLaunchKernelGGL(bin_tile_sort_long_kernel, dim3(n_tiles), dim3(512), 0, stream, (unsigned long long*)pairs, (unsigned long long*)scratch, offsets,
                       n_tiles, n_isects, flatten_ids, tile_ids);
    ADK_RETURN_LAST_ERROR();
}
extern "C" int64_t adk_bin_local_sort_long_max(void) { return BIN_SORT_LONG_MAX; }

extern "C" int adk_bin_local_sort(int64_t n_isects, int64_t max_tile, int width, int height, const int32_t* offsets, const void* pairs,
                                  int32_t* flatten_ids, uint32_t* tile_ids, hipStream_t stream)
{
    return adk_bin_local_sort_t(n_isects, max_tile, width, height, 16, 16, offsets, pairs, flatten_ids, tile_ids, stream);
}

// Optional (meta parity): upstream's sorted 64-bit keys, rebuilt from the tile-sorted list.
extern "C" int adk_bin_make_isect_ids(int64_t n_isects, const uint32_t* tile_ids, const int32_t* flatten_ids,
                                      const uint32_t* depth_keys, int64_t* isect_ids, hipStream_t stream)
{
    if (n_isects < 0) return ADK_EINVAL;
    if (n_isects == 0) return 0;
    if (!tile_ids || !flatten_ids || !depth_keys || !isect_ids) return ADK_EINVAL;
    hipLaunchKernelGGL(adk::make_isect_ids_kernel, dim3((unsigned)adk::ceil_div(n_isects, 256)), dim3(256), 0, stream, tile_ids, flatten_ids, depth_keys, n_isects, isect_ids);
    ADK_RETURN_LAST_ERROR();
}
