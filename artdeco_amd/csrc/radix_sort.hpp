// Stable LSD radix sort of (u32 key, u32 value) pairs for gfx950 -- shared by the tile binning
// (raster_bin.hip) and the Morton ordering of simple-knn (knn.hip).
//
// One pass = 3 launches: per-block digit histogram (ballot-aggregated LDS adds), per-digit scan
// over blocks, stable scatter.  The scatter ranks keys inside a wavefront with 8 ballots (wave64
// match-any), so a round of 256 keys costs 3 barriers, no LDS atomics, and preserves input order.
#pragma once
#include "adk_common.hpp"
#include <stdlib.h>

namespace adk {

#define RS_BLOCK 256
// Keys per thread: 16 for long lists, 8 below ~2 M keys (measured on MI355X: the 1 M-key depth sort drops from 0.157 to
// 0.130 ms with twice as many, half as long workgroups; the 3.8 M-key tile sort is fastest at 16; 4 loses on both).
static inline int rs_items(int64_t n) { return n < (int64_t)2 * 1024 * 1024 ? 8 : 16; }

// Wave64 match-any on an 8-bit digit: mask of the lanes (among `ok` lanes) holding the same digit.
__device__ __forceinline__ unsigned long long match_digit(uint32_t d, bool ok) {
    unsigned long long m = __ballot(ok);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const unsigned long long bb = __ballot((d >> b) & 1u);
        m &= ((d >> b) & 1u) ? bb : ~bb;
    }
    return m;
}

template <int RS_ITEMS>
static __global__ __launch_bounds__(RS_BLOCK) void radix_hist_kernel(const uint32_t* __restrict__ keys, int64_t n, int shift,
                                                                     uint32_t* __restrict__ block_hist, int nblocks)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned long long lanes_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int64_t base = (int64_t)blockIdx.x * (RS_BLOCK * RS_ITEMS);
#pragma unroll 4
    for (int r = 0; r < RS_ITEMS; ++r) {
        const int64_t i = base + r * RS_BLOCK + threadIdx.x;
        const bool ok = i < n;
        const uint32_t d = ok ? ((keys[i] >> shift) & 255u) : 0u;
        // one LDS add per distinct digit per wave (sort keys are often digit-uniform: depth MSBs, tile MSBs)
        const unsigned long long m = match_digit(d, ok);
        if (ok && (m & lanes_lt) == 0ull) atomicAdd(&h[d], (uint32_t)__popcll(m));
    }
    __syncthreads();
    block_hist[(int64_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// One block per digit: exclusive scan of that digit's per-block counts, in place; digit total out.
static __global__ __launch_bounds__(256) void radix_scan_kernel(uint32_t* __restrict__ block_hist, int nblocks,
                                                         uint32_t* __restrict__ digit_total)
{
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t carry_s;
    uint32_t* row = block_hist + (int64_t)blockIdx.x * nblocks;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nblocks; b0 += 256) {
        const int i = b0 + threadIdx.x;
        const uint32_t v = (i < nblocks) ? row[i] : 0u;
        uint32_t s = v; // inclusive wave scan
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(s, o, 64); if (lane >= o) s += t; }
        if (lane == 63) wsum[wv] = s;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wv; ++w) woff += wsum[w];
        const uint32_t carry = carry_s;
        if (i < nblocks) row[i] = carry + woff + s - v;
        __syncthreads();
        if (threadIdx.x == 255) carry_s = carry + woff + s;
        __syncthreads();
    }
    if (threadIdx.x == 0) digit_total[blockIdx.x] = carry_s;
}

template <int RS_ITEMS>
static __global__ __launch_bounds__(RS_BLOCK) void radix_scatter_kernel(
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out, int64_t n, int shift, const uint32_t* __restrict__ block_hist, int nblocks,
    const uint32_t* __restrict__ digit_total)
{
    __shared__ uint32_t base[256];      // running global offset per digit for this block
    __shared__ uint32_t wave_cnt[4][256];
    __shared__ uint32_t wtot[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

    // exclusive scan of the 256 digit totals (every block redoes it: 256 values, trivial)
    {
        const uint32_t v = digit_total[tid];
        uint32_t s = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(s, o, 64); if (lane >= o) s += t; }
        if (lane == 63) wtot[wv] = s;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wv; ++w) woff += wtot[w];
        base[tid] = woff + s - v + block_hist[(int64_t)tid * nblocks + blockIdx.x];
#pragma unroll
        for (int w = 0; w < 4; ++w) wave_cnt[w][tid] = 0;
    }
    __syncthreads();

    const unsigned long long lanes_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int64_t blk_base = (int64_t)blockIdx.x * (RS_BLOCK * RS_ITEMS);
    for (int r = 0; r < RS_ITEMS; ++r) {
        const int64_t i = blk_base + r * RS_BLOCK + tid;
        if (blk_base + r * RS_BLOCK >= n) break; // uniform
        const bool ok = i < n;
        uint32_t key = 0, val = 0;
        if (ok) { key = keys_in[i]; val = vals_in[i]; }
        const uint32_t d = (key >> shift) & 255u;
        const unsigned long long m = match_digit(d, ok);
        const uint32_t rank = (uint32_t)__popcll(m & lanes_lt);
        if (ok && rank == 0) wave_cnt[wv][d] = (uint32_t)__popcll(m);
        __syncthreads();
        if (ok) {
            uint32_t off = base[d] + rank;
            for (int w = 0; w < wv; ++w) off += wave_cnt[w][d];
            keys_out[off] = key;
            vals_out[off] = val;
        }
        __syncthreads();
        {
            const uint32_t c = wave_cnt[0][tid] + wave_cnt[1][tid] + wave_cnt[2][tid] + wave_cnt[3][tid];
            base[tid] += c;
#pragma unroll
            for (int w = 0; w < 4; ++w) wave_cnt[w][tid] = 0;
        }
        __syncthreads();
    }
}

// Same pass, STAGED: the ranks of a chunk are computed exactly as above, but the (key, value) pairs first go to their
// digit-sorted slot in LDS and are written out afterwards by consecutive threads, so every digit's keys of this chunk
// leave as one contiguous run (64 B on average at 2048 keys / 256 digits, 128 B at 4096) instead of 2 x 4-byte stores
// scattered over the whole output per key.  Output is identical (stable order inside a digit = input order).
template <int RS_ITEMS>
static __global__ __launch_bounds__(RS_BLOCK) void radix_scatter_staged_kernel(
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out, int64_t n, int shift, const uint32_t* __restrict__ block_hist, int nblocks,
    const uint32_t* __restrict__ digit_total)
{
    constexpr int CH = RS_BLOCK * RS_ITEMS;
    __shared__ uint32_t gbase[256];   // global position of this chunk's first key of each digit
    __shared__ uint32_t lstart[256];  // position of each digit's run inside the chunk (exclusive scan of the chunk's counts)
    __shared__ uint32_t run[256];     // running write position per digit inside the chunk
    __shared__ uint32_t wave_cnt[4][256];
    __shared__ uint32_t wtot[4], wtot2[4];
    __shared__ uint32_t skey[CH], sval[CH];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    {
        const uint32_t tot = digit_total[tid];
        const uint32_t excl_blk = block_hist[(int64_t)tid * nblocks + blockIdx.x]; // keys of this digit in earlier chunks
        const uint32_t next = (blockIdx.x + 1 < nblocks) ? block_hist[(int64_t)tid * nblocks + blockIdx.x + 1] : tot;
        const uint32_t cnt = next - excl_blk;                                      // ... and in this chunk
        uint32_t s = tot, c = cnt; // two inclusive wave scans: digit totals (global bases) and chunk counts (local starts)
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(s, o, 64), u = __shfl_up(c, o, 64);
            if (lane >= o) { s += t; c += u; }
        }
        if (lane == 63) { wtot[wv] = s; wtot2[wv] = c; }
        __syncthreads();
        uint32_t woff = 0, woff2 = 0;
        for (int w = 0; w < wv; ++w) { woff += wtot[w]; woff2 += wtot2[w]; }
        gbase[tid] = woff + s - tot + excl_blk;
        lstart[tid] = woff2 + c - cnt;
        run[tid] = woff2 + c - cnt;
#pragma unroll
        for (int w = 0; w < 4; ++w) wave_cnt[w][tid] = 0;
    }
    __syncthreads();
    const unsigned long long lanes_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int64_t blk_base = (int64_t)blockIdx.x * CH;
    for (int r = 0; r < RS_ITEMS; ++r) {
        const int64_t i = blk_base + r * RS_BLOCK + tid;
        if (blk_base + r * RS_BLOCK >= n) break; // uniform
        const bool ok = i < n;
        uint32_t key = 0, val = 0;
        if (ok) { key = keys_in[i]; val = vals_in[i]; }
        const uint32_t d = (key >> shift) & 255u;
        const unsigned long long m = match_digit(d, ok);
        const uint32_t rank = (uint32_t)__popcll(m & lanes_lt);
        if (ok && rank == 0) wave_cnt[wv][d] = (uint32_t)__popcll(m);
        __syncthreads();
        if (ok) {
            uint32_t off = run[d] + rank;
            for (int w = 0; w < wv; ++w) off += wave_cnt[w][d];
            skey[off] = key;
            sval[off] = val;
        }
        __syncthreads();
        {
            run[tid] += wave_cnt[0][tid] + wave_cnt[1][tid] + wave_cnt[2][tid] + wave_cnt[3][tid];
#pragma unroll
            for (int w = 0; w < 4; ++w) wave_cnt[w][tid] = 0;
        }
        __syncthreads();
    }
    const int64_t rem = n - blk_base;
    const int m_chunk = rem < (int64_t)CH ? (int)rem : CH;
    for (int i = tid; i < m_chunk; i += RS_BLOCK) {
        const uint32_t key = skey[i];
        const uint32_t d = (key >> shift) & 255u;
        const uint32_t out = gbase[d] + ((uint32_t)i - lstart[d]);
        keys_out[out] = key;
        vals_out[out] = sval[i];
    }
}

// MEASURED DEAD END (round 2, MI355X): a ONESWEEP form of the pass -- one upfront histogram kernel for the digit totals of all
// passes, then ONE launch per pass in which every workgroup takes a ticket, publishes its 256 digit counts and obtains its
// per-digit prefix by decoupled look-back over its predecessors' status words (agent-scope atomics, one thread per digit) --
// was bit-identical and much slower: 1 M-key depth sort 0.12 -> 1.97 ms with a one-word-at-a-time walk, 0.42 ms with 8 words
// in flight per round; 3.8 M-key tile sort 0.15 -> 0.46 ms.  The lists are small (512-928 co-resident workgroups that all
// start together), so nothing hides the chain of cross-XCD round trips (~1 us each) the look-back needs before the first
// inclusive prefixes exist; three dependent launches (~3-5 us gaps) cost less.  Removed again; see git history.

// Scratch needed by one radix sort over n items: histogram table + digit totals.
static inline int64_t radix_scratch_bytes(int64_t n) {
    const int64_t nb = ceil_div(n > 0 ? n : 1, RS_BLOCK * 8); // the smaller chunk bounds the table size
    return (256 * nb + 256) * (int64_t)sizeof(uint32_t);
}

// LSD sort of (key,val) pairs on bits [bit_lo, bit_hi).  Ping-pongs between (k0,v0) and (k1,v1);
// the first pass reads (k_src,v_src) which is left untouched.  Returns index (0/1) of the buffer
// holding the result.
static int radix_sort_pairs(const uint32_t* k_src, const uint32_t* v_src, uint32_t* k0, uint32_t* v0, uint32_t* k1,
                            uint32_t* v1, int64_t n, int bit_lo, int bit_hi, uint32_t* scratch, hipStream_t stream)
{
    const int items = rs_items(n);
    const int nb = (int)ceil_div(n, RS_BLOCK * items);
    // Staged scatter is the default since round 2: the whole bit-exact GPU suite (binning cases, KNN Morton sort, 1 M / 1080p
    // properties; 228 tests) passed with it on MI355X (gpurun_out/r02_staged_suite.log).  1 M-key depth sort 0.130 -> 0.120 ms,
    // 3.8 M-key tile sort 0.190 -> 0.150 ms.  ADK_RADIX_STAGED=0 selects the direct scatter (kept for A/B measurements).
    static const bool staged = [] { const char* e = getenv("ADK_RADIX_STAGED"); return !(e && e[0] == '0'); }();
    uint32_t* hist = scratch;
    uint32_t* dtot = scratch + (int64_t)256 * nb;
    const uint32_t* ki = k_src;
    const uint32_t* vi = v_src;
    int dst = 0;
    for (int shift = bit_lo; shift < bit_hi; shift += 8) {
        uint32_t* ko = dst ? k1 : k0;
        uint32_t* vo = dst ? v1 : v0;
        if (items == 8) hipLaunchKernelGGL(radix_hist_kernel<8>, dim3(nb), dim3(RS_BLOCK), 0, stream, ki, n, shift, hist, nb);
        else hipLaunchKernelGGL(radix_hist_kernel<16>, dim3(nb), dim3(RS_BLOCK), 0, stream, ki, n, shift, hist, nb);
        hipLaunchKernelGGL(radix_scan_kernel, dim3(256), dim3(256), 0, stream, hist, nb, dtot);
        if (staged) {
            if (items == 8) hipLaunchKernelGGL(radix_scatter_staged_kernel<8>, dim3(nb), dim3(RS_BLOCK), 0, stream, ki, vi, ko, vo, n, shift, hist, nb, dtot);
            else hipLaunchKernelGGL(radix_scatter_staged_kernel<16>, dim3(nb), dim3(RS_BLOCK), 0, stream, ki, vi, ko, vo, n, shift, hist, nb, dtot);
        } else {
            if (items == 8) hipLaunchKernelGGL(radix_scatter_kernel<8>, dim3(nb), dim3(RS_BLOCK), 0, stream, ki, vi, ko, vo, n, shift, hist, nb, dtot);
            else hipLaunchKernelGGL(radix_scatter_kernel<16>, dim3(nb), dim3(RS_BLOCK), 0, stream, ki, vi, ko, vo, n, shift, hist, nb, dtot);
        }
        ki = ko; vi = vo;
        dst ^= 1;
    }
    return dst ^ 1;
}

} // namespace adk
