// SceneModel.update_voxel on the device (densification path) for gfx950.
//
// Replaces the torch chain of Reconstruct/scene/scene_models/h3dgsv3.py:227-316 -- a common voxel grid over old + new points,
// a linear hash, three torch.unique (each a sort + host-visible size), torch_scatter.scatter_max for the majority class of
// every occupied voxel, a searchsorted for the new points, boolean-mask writes and two .item()/.any() host reads -- called once
// per LoD level for every important frame (h3dgsv3.py:884-887).  Same results, bit for bit (tests/golden/voxel_*.npz, produced
// by the reference's own method source):
//   * voxel index floor((p - min) / voxel_size) in fp32 -- rounded as torch's CPU kernel (true division, the goldens) or as its
//     GPU kernel (multiplication by the fp32 reciprocal), caller's choice -- hash = ix * (ny nz) + iy * nz + iz (int64);
//   * majority class per voxel, the SMALLEST class among equally frequent ones (torch.unique visits (voxel, class) pairs in
//     sorted order and scatter_max keeps the first maximum);
//   * old points take their voxel's majority class; a new point takes the majority class of the voxel it falls into, or
//     max_cls + 1 + (rank of its hash among the sorted distinct hashes of new-only voxels).
// Route: stable LSD radix sorts of the point indices by (hash, class) (the shared radix sort, 32-bit words, only as many
// 8-bit passes as the values have bits), run flags + two scans for the voxel / (voxel, class) ranks, one packed 64-bit
// atomicMax per (voxel, class) run for the vote (count << 32 | ~class), a binary search per new point.  The host reads the
// grid extents once (they decide the number of sort passes) and the new-voxel count once (it sizes `global_feat`).
#include "adk_common.hpp"
#include "radix_sort.hpp"

namespace adk {

__device__ __forceinline__ uint32_t vox_f32_key(float f) { const uint32_t b = __float_as_uint(f); return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u); }
__device__ __forceinline__ float vox_f32_unkey(uint32_t k) { return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu)); }

__device__ __forceinline__ const float* vox_point(const float* xyz, int64_t N, const float* new_xyz, int64_t i) {
    return i < N ? xyz + 3 * i : new_xyz + 3 * (i - N);
}

// bounds[0..2] = order-preserving keys of the componentwise minimum (init 0xFFFFFFFF), imax[0..2] = max voxel index (init 0),
// max_cls (init INT64_MIN)
__global__ __launch_bounds__(256) void vox_init_kernel(uint32_t* bounds, unsigned long long* imax, long long* max_cls) {
    if (threadIdx.x < 3) { bounds[threadIdx.x] = 0xFFFFFFFFu; imax[threadIdx.x] = 0ull; }
    if (threadIdx.x == 3) *max_cls = (long long)0x8000000000000000ull;
}

__global__ __launch_bounds__(256) void vox_min_kernel(const float* __restrict__ xyz, int64_t N, const float* __restrict__ new_xyz, int64_t M,
                                                      const int64_t* __restrict__ cls, uint32_t* bounds, long long* max_cls) {
    uint32_t k[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    long long mc = (long long)0x8000000000000000ull;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N + M; i += stride) {
        const float* p = vox_point(xyz, N, new_xyz, i);
#pragma unroll
        for (int c = 0; c < 3; ++c) k[c] = min(k[c], vox_f32_key(p[c]));
        // a negative class id (never produced by ARTDECO) would be truncated by the 32-bit sort keys: report it as an
        // out-of-range maximum, which adk_voxel_assign refuses (ADK_EUNSUPPORTED -> the caller's torch path)
        if (i < N) { const long long c = (long long)cls[i]; mc = max(mc, c < 0 ? 0x7fffffffffffffffLL : c); }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
#pragma unroll
        for (int c = 0; c < 3; ++c) k[c] = min(k[c], (uint32_t)__shfl_xor((int)k[c], o, 64));
        mc = max(mc, (long long)__shfl_xor(mc, o, 64));
    }
    // one set of same-address device atomics per WORKGROUP, not per wave (8192 waves x 4 atomics on 4 addresses took 0.38 ms of
    // the 1.1 ms call, profiles/r03_step_kernel_stats.csv)
    __shared__ uint32_t sk[4][3];
    __shared__ long long smc[4];
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sk[wv][0] = k[0]; sk[wv][1] = k[1]; sk[wv][2] = k[2]; smc[wv] = mc; }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(bounds + threadIdx.x, min(min(sk[0][threadIdx.x], sk[1][threadIdx.x]), min(sk[2][threadIdx.x], sk[3][threadIdx.x])));
    if (threadIdx.x == 3 && N > 0) atomicMax(max_cls, max(max(smc[0], smc[1]), max(smc[2], smc[3])));
}

// `vs` is the divisor when recip == 0 (torch's CPU kernel for tensor / python-float: a true division -- what the goldens of
// tests/golden/voxel_*.npz were produced with), or the fp32 reciprocal 1.0f / voxel_size when recip != 0 (torch's GPU kernel
// for the same expression multiplies by the reciprocal computed in fp32: the two differ in the last bit on ~1e-7 of the
// coordinates, enough to move a point sitting on a voxel face).
__device__ __forceinline__ long long vox_index(float p, float mn, float vs, int recip) {
    return (long long)floorf(recip ? (p - mn) * vs : (p - mn) / vs);
}

__global__ __launch_bounds__(256) void vox_imax_kernel(const float* __restrict__ xyz, int64_t N, const float* __restrict__ new_xyz, int64_t M,
                                                       float vs, int recip, const uint32_t* __restrict__ bounds, unsigned long long* imax) {
    const float mn[3] = {vox_f32_unkey(bounds[0]), vox_f32_unkey(bounds[1]), vox_f32_unkey(bounds[2])};
    unsigned long long m[3] = {0ull, 0ull, 0ull};
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N + M; i += stride) {
        const float* p = vox_point(xyz, N, new_xyz, i);
#pragma unroll
        for (int c = 0; c < 3; ++c) m[c] = max(m[c], (unsigned long long)vox_index(p[c], mn[c], vs, recip));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
#pragma unroll
        for (int c = 0; c < 3; ++c) m[c] = max(m[c], (unsigned long long)__shfl_xor((long long)m[c], o, 64));
    __shared__ unsigned long long sm[4][3];   // one set of atomics per workgroup (see vox_min_kernel)
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sm[wv][0] = m[0]; sm[wv][1] = m[1]; sm[wv][2] = m[2]; }
    __syncthreads();
    if (threadIdx.x < 3) atomicMax(imax + threadIdx.x, max(max(sm[0][threadIdx.x], sm[1][threadIdx.x]), max(sm[2][threadIdx.x], sm[3][threadIdx.x])));
}

__global__ void vox_finish_bounds_kernel(const uint32_t* bounds, const unsigned long long* imax, const long long* max_cls, int64_t N,
                                         float* minc, int64_t* info) {
    if (threadIdx.x < 3) { minc[threadIdx.x] = vox_f32_unkey(bounds[threadIdx.x]); info[threadIdx.x] = (int64_t)imax[threadIdx.x] + 1; }
    if (threadIdx.x == 3) info[3] = N > 0 ? (int64_t)*max_cls : -1;
}

// hash of points [first, first + n) of the (old ++ new) list, + iota
__global__ __launch_bounds__(256) void vox_hash_kernel(const float* __restrict__ pts, int64_t n, float vs, int recip, const float* __restrict__ minc,
                                                       long long sy, long long sz, unsigned long long* __restrict__ hash, uint32_t* __restrict__ iota) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long ix = vox_index(pts[3 * i], minc[0], vs, recip), iy = vox_index(pts[3 * i + 1], minc[1], vs, recip), iz = vox_index(pts[3 * i + 2], minc[2], vs, recip);
    hash[i] = (unsigned long long)(ix * sy + iy * sz + iz);
    iota[i] = (uint32_t)i;
}

// key[i] = word `which` (0: class, 1: hash low, 2: hash high) of element idx[i]
__global__ __launch_bounds__(256) void vox_gather_key_kernel(const uint32_t* __restrict__ idx, int64_t n, int which,
                                                             const unsigned long long* __restrict__ hash, const int64_t* __restrict__ cls,
                                                             uint32_t* __restrict__ key) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = idx[i];
    key[i] = which == 0 ? (uint32_t)cls[p] : (which == 1 ? (uint32_t)hash[p] : (uint32_t)(hash[p] >> 32));
}

__global__ __launch_bounds__(256) void vox_copy_u32_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// ---- inclusive scan of u32 (3 launches; n < 2^31) ----------------------------------------------------------------
#define VSCAN_ITEMS 8
__global__ __launch_bounds__(256) void vscan_block_sums_kernel(const uint32_t* __restrict__ in, int64_t n, uint32_t* __restrict__ sums) {
    __shared__ uint32_t ws[4];
    const int64_t base = (int64_t)blockIdx.x * (256 * VSCAN_ITEMS);
    uint32_t s = 0;
#pragma unroll
    for (int r = 0; r < VSCAN_ITEMS; ++r) { const int64_t i = base + r * 256 + threadIdx.x; s += i < n ? in[i] : 0u; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}
// one block: exclusive scan of the block sums in place; total -> *total
__global__ __launch_bounds__(1024) void vscan_sums_kernel(uint32_t* __restrict__ sums, int nb, uint32_t* __restrict__ total) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nb; b0 += 1024) {
        const int i = b0 + threadIdx.x;
        const uint32_t v = i < nb ? sums[i] : 0u;
        uint32_t s = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(s, o, 64); if (lane >= o) s += t; }
        if (lane == 63) wsum[wv] = s;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wv; ++w) woff += wsum[w];
        const uint32_t carry = carry_s;
        if (i < nb) sums[i] = carry + woff + s - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + s;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry_s;
}
__global__ __launch_bounds__(256) void vscan_apply_kernel(const uint32_t* __restrict__ in, int64_t n, const uint32_t* __restrict__ sums,
                                                          uint32_t* __restrict__ out) {
    __shared__ uint32_t ws[4];
    __shared__ uint32_t carry_s;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * (256 * VSCAN_ITEMS);
    if (threadIdx.x == 0) carry_s = sums[blockIdx.x];
    __syncthreads();
    for (int r = 0; r < VSCAN_ITEMS; ++r) {
        const int64_t i = base + r * 256 + threadIdx.x;
        const uint32_t v = i < n ? in[i] : 0u;
        uint32_t s = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(s, o, 64); if (lane >= o) s += t; }
        if (lane == 63) ws[wv] = s;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wv; ++w) woff += ws[w];
        const uint32_t carry = carry_s;
        if (i < n) out[i] = carry + woff + s;
        __syncthreads();
        if (threadIdx.x == 255) carry_s = carry + woff + s;
        __syncthreads();
    }
}
static void inclusive_scan_u32(const uint32_t* in, int64_t n, uint32_t* out, uint32_t* sums, uint32_t* total, hipStream_t stream) {
    const int nb = (int)ceil_div(n, 256 * VSCAN_ITEMS);
    hipLaunchKernelGGL(vscan_block_sums_kernel, dim3(nb), dim3(256), 0, stream, in, n, sums);
    hipLaunchKernelGGL(vscan_sums_kernel, dim3(1), dim3(1024), 0, stream, sums, nb, total);
    hipLaunchKernelGGL(vscan_apply_kernel, dim3(nb), dim3(256), 0, stream, in, n, sums, out);
}

// ---- old points, sorted by (hash, class): run flags ---------------------------------------------------------------
__global__ __launch_bounds__(256) void vox_flags_kernel(const uint32_t* __restrict__ order, int64_t n, const unsigned long long* __restrict__ hash,
                                                        const int64_t* __restrict__ cls, uint32_t* __restrict__ vflag, uint32_t* __restrict__ pflag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = order[i];
    bool v = true, q = true;
    if (i > 0) {
        const uint32_t pp = order[i - 1];
        v = hash[p] != hash[pp];
        q = v || cls[p] != cls[pp];
    }
    vflag[i] = v ? 1u : 0u;
    pflag[i] = q ? 1u : 0u;
}

// start position of every (voxel, class) run, hash of every voxel; start[n_pairs] = n is written by the last element
__global__ __launch_bounds__(256) void vox_runs_kernel(const uint32_t* __restrict__ order, int64_t n, const unsigned long long* __restrict__ hash,
                                                       const uint32_t* __restrict__ vflag, const uint32_t* __restrict__ pflag,
                                                       const uint32_t* __restrict__ vrank, const uint32_t* __restrict__ prank,
                                                       uint32_t* __restrict__ start, unsigned long long* __restrict__ uniq_hash,
                                                       unsigned long long* __restrict__ mode_key) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (pflag[i]) start[prank[i] - 1] = (uint32_t)i;
    if (vflag[i]) { uniq_hash[vrank[i] - 1] = hash[order[i]]; mode_key[vrank[i] - 1] = 0ull; }
    if (i == n - 1) start[prank[i]] = (uint32_t)n;
}

// one thread per (voxel, class) run: packed vote  count << 32 | ~class  (larger count wins, then the smaller class)
__global__ __launch_bounds__(256) void vox_vote_kernel(const uint32_t* __restrict__ order, const uint32_t* __restrict__ n_pairs,
                                                       const uint32_t* __restrict__ start, const uint32_t* __restrict__ vrank,
                                                       const int64_t* __restrict__ cls, unsigned long long* __restrict__ mode_key) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (int64_t)*n_pairs) return;
    const uint32_t s = start[k], e = start[k + 1];
    const uint32_t c = (uint32_t)cls[order[s]];
    atomicMax(mode_key + (vrank[s] - 1), ((unsigned long long)(e - s) << 32) | (unsigned long long)(0xFFFFFFFFu - c));
}

__global__ __launch_bounds__(256) void vox_label_old_kernel(const uint32_t* __restrict__ order, int64_t n, const uint32_t* __restrict__ vrank,
                                                            const unsigned long long* __restrict__ mode_key, int64_t* __restrict__ updated) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    updated[order[i]] = (int64_t)(0xFFFFFFFFu - (uint32_t)mode_key[vrank[i] - 1]);
}

// ---- new points, sorted by hash -----------------------------------------------------------------------------------
// hit[i] = position of the point's voxel among the old voxels, or 0xFFFFFFFF; flag[i] = first point of a NEW voxel
__global__ __launch_bounds__(256) void vox_match_new_kernel(const uint32_t* __restrict__ order, int64_t m, const unsigned long long* __restrict__ hash,
                                                            const unsigned long long* __restrict__ uniq_hash, const uint32_t* __restrict__ n_uniq,
                                                            uint32_t* __restrict__ hit, uint32_t* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const unsigned long long h = hash[order[i]];
    uint32_t found = 0xFFFFFFFFu;
    const uint32_t U = n_uniq ? *n_uniq : 0u;
    uint32_t lo = 0, hi = U; // first position with uniq_hash >= h (torch.searchsorted, left)
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (uniq_hash[mid] < h) lo = mid + 1; else hi = mid; }
    if (lo < U && uniq_hash[lo] == h) found = lo;
    hit[i] = found;
    flag[i] = (found == 0xFFFFFFFFu && (i == 0 || hash[order[i - 1]] != h)) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void vox_label_new_kernel(const uint32_t* __restrict__ order, int64_t m, const uint32_t* __restrict__ hit,
                                                            const uint32_t* __restrict__ rank, const unsigned long long* __restrict__ mode_key,
                                                            int64_t first_new_label, int64_t* __restrict__ updated) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const uint32_t h = hit[i];
    updated[order[i]] = h != 0xFFFFFFFFu ? (int64_t)(0xFFFFFFFFu - (uint32_t)mode_key[h]) : first_new_label + (int64_t)rank[i] - 1;
}

__global__ void vox_store_count_kernel(const uint32_t* total, int64_t* count) { *count = total ? (int64_t)*total : 0; }

static inline int bits_of(unsigned long long v) { int b = 0; while (v) { ++b; v >>= 1; } return b; }

struct VoxWs {
    unsigned long long *hash, *uniq_hash, *mode_key;
    uint32_t *idx, *key, *k0, *v0, *k1, *v1, *vflag, *pflag, *vrank, *prank, *start, *sums, *totals, *scratch;
};
static inline int64_t a256(int64_t x) { return (x + 255) & ~(int64_t)255; }
static int64_t vox_ws_bytes(int64_t n) {
    const int64_t m = n > 0 ? n : 1;
    return 3 * a256(m * 8) + 11 * a256((m + 1) * 4) + a256((ceil_div(m, 256 * VSCAN_ITEMS) + 1) * 4) + 256 + a256(radix_scratch_bytes(m)) + 256;
}
static VoxWs vox_carve(void* ws, int64_t n) {
    const int64_t m = n > 0 ? n : 1;
    char* p = (char*)ws;
    VoxWs w;
    w.hash = (unsigned long long*)p; p += a256(m * 8);
    w.uniq_hash = (unsigned long long*)p; p += a256(m * 8);
    w.mode_key = (unsigned long long*)p; p += a256(m * 8);
    uint32_t** u[] = {&w.idx, &w.key, &w.k0, &w.v0, &w.k1, &w.v1, &w.vflag, &w.pflag, &w.vrank, &w.prank, &w.start};
    for (auto q : u) { *q = (uint32_t*)p; p += a256((m + 1) * 4); }
    w.sums = (uint32_t*)p; p += a256((ceil_div(m, 256 * VSCAN_ITEMS) + 1) * 4);
    w.totals = (uint32_t*)p; p += 256;
    w.scratch = (uint32_t*)p;
    return w;
}

// stable sort of w.idx (point indices) by one 32-bit word of its points, `bits` significant bits
static void vox_sort_by_word(VoxWs& w, int64_t n, int which, int bits, const int64_t* cls, hipStream_t stream) {
    if (bits <= 0) return; // every key equal: the order does not change
    const unsigned g = (unsigned)ceil_div(n, (int64_t)256);
    hipLaunchKernelGGL(vox_gather_key_kernel, dim3(g), dim3(256), 0, stream, w.idx, n, which, w.hash, cls, w.key);
    const int hi = ((bits + 7) / 8) * 8;
    const int res = radix_sort_pairs(w.key, w.idx, w.k0, w.v0, w.k1, w.v1, n, 0, hi, w.scratch, stream);
    hipLaunchKernelGGL(vox_copy_u32_kernel, dim3(g), dim3(256), 0, stream, res ? w.v1 : w.v0, w.idx, n);
}

} // namespace adk

// Stage 1: grid origin (componentwise minimum of old ++ new points), grid extents and the largest class id.
// minc [3] float (device), info [4] int64 (device): extents nx, ny, nz (= largest voxel index + 1) and max(cls_id) (-1 when N = 0).
// The caller reads `info` (one small device-to-host copy) and hands it to adk_voxel_assign.  workspace: 64 bytes.
extern "C" int adk_voxel_bounds(const float* xyz, int64_t N, const float* new_xyz, int64_t M, const int64_t* cls_id, float voxel_size,
                                int use_reciprocal, float* minc, int64_t* info, void* workspace, int64_t workspace_bytes,
                                hipStream_t stream)
{
    const float vs_arg = use_reciprocal ? 1.0f / voxel_size : voxel_size;
    if (N < 0 || M < 0 || N + M == 0 || !(voxel_size > 0.f) || !minc || !info || !workspace || workspace_bytes < 64) return ADK_EINVAL;
    if ((N > 0 && (!xyz || !cls_id)) || (M > 0 && !new_xyz)) return ADK_EINVAL;
    uint32_t* bounds = (uint32_t*)workspace;
    unsigned long long* imax = (unsigned long long*)((char*)workspace + 16);
    long long* max_cls = (long long*)((char*)workspace + 48);
    int g = adk::stream_grid(N + M, 256);
    if (g > 512) g = 512;   // 2 workgroups per CU read 12 MB in a few microseconds; the tail of the kernel is its same-address atomics
    hipLaunchKernelGGL(adk::vox_init_kernel, dim3(1), dim3(64), 0, stream, bounds, imax, max_cls);
    hipLaunchKernelGGL(adk::vox_min_kernel, dim3(g), dim3(256), 0, stream, xyz, N, new_xyz, M, cls_id, bounds, max_cls);
    hipLaunchKernelGGL(adk::vox_imax_kernel, dim3(g), dim3(256), 0, stream, xyz, N, new_xyz, M, vs_arg, use_reciprocal, bounds, imax);
    hipLaunchKernelGGL(adk::vox_finish_bounds_kernel, dim3(1), dim3(64), 0, stream, bounds, imax, max_cls, N, minc, info);
    ADK_RETURN_LAST_ERROR();
}

extern "C" int64_t adk_voxel_workspace_bytes(int64_t N, int64_t M)
{
    if (N < 0 || M < 0) return ADK_EINVAL;
    return adk::vox_ws_bytes(N) + adk::vox_ws_bytes(M) + 512;
}

// Stage 2.  nx, ny, nz, max_cls: the HOST copies of `info`.  updated_orig [N] and updated_new [M] (int64, device) are fully
// written; *new_voxel_count (int64, device) = number of voxels that only new points fall into.  With N = 0 (cold start,
// h3dgsv3.py:244-255) updated_new holds the rank of each point's voxel among the sorted distinct voxels and the count is their number.
extern "C" int adk_voxel_assign(const float* xyz, int64_t N, const float* new_xyz, int64_t M, const int64_t* cls_id, float voxel_size,
                                int use_reciprocal, const float* minc, int64_t nx, int64_t ny, int64_t nz, int64_t max_cls, int64_t* updated_orig,
                                int64_t* updated_new, int64_t* new_voxel_count, void* workspace, int64_t workspace_bytes,
                                hipStream_t stream)
{
    using namespace adk;
    if (N < 0 || M < 0 || N + M == 0 || !(voxel_size > 0.f) || !minc || !new_voxel_count || !workspace) return ADK_EINVAL;
    if ((N > 0 && (!xyz || !cls_id || !updated_orig)) || (M > 0 && (!new_xyz || !updated_new))) return ADK_EINVAL;
    if (nx <= 0 || ny <= 0 || nz <= 0 || N >= ((int64_t)1 << 31) || M >= ((int64_t)1 << 31)) return ADK_EINVAL;
    if (workspace_bytes < adk_voxel_workspace_bytes(N, M) || ((uintptr_t)workspace & 255)) return ADK_EWORKSPACE;
    // the hash must fit 63 bits and class ids 31 bits (they are voxel counters: h3dgsv3.py:312)
    const long double cells = (long double)nx * (long double)ny * (long double)nz;
    if (cells >= 9.0e18L || max_cls >= ((int64_t)1 << 31) || (N > 0 && max_cls < 0)) return ADK_EUNSUPPORTED;
    const unsigned long long max_hash = (unsigned long long)nx * (unsigned long long)ny * (unsigned long long)nz - 1ull;
    const int hbits = bits_of(max_hash), cbits = N > 0 ? bits_of((unsigned long long)max_cls) : 0;
    const long long sy = (long long)(ny * nz), sz = (long long)nz;
    const float vs_arg = use_reciprocal ? 1.0f / voxel_size : voxel_size;
    // layout: [old points' tables (N)] [tail: 256 B] [new points' tables (M)] -- the old points' part does not depend on M, so that
    // adk_voxel_assign_new can come back to it with another batch of new points
    VoxWs wo = vox_carve(workspace, N);
    uint32_t* tail = (uint32_t*)((char*)workspace + vox_ws_bytes(N)); // [0] voxels, [1] pairs, [2] new voxels
    VoxWs wn = vox_carve((char*)workspace + vox_ws_bytes(N) + 256, M);

    if (N > 0) {
        const unsigned g = (unsigned)ceil_div(N, (int64_t)256);
        hipLaunchKernelGGL(vox_hash_kernel, dim3(g), dim3(256), 0, stream, xyz, N, vs_arg, use_reciprocal, minc, sy, sz, wo.hash, wo.idx);
        vox_sort_by_word(wo, N, 0, cbits, cls_id, stream);                 // least significant key first: class ...
        vox_sort_by_word(wo, N, 1, hbits < 32 ? hbits : 32, cls_id, stream); // ... then the hash, low word, high word
        vox_sort_by_word(wo, N, 2, hbits - 32, cls_id, stream);
        hipLaunchKernelGGL(vox_flags_kernel, dim3(g), dim3(256), 0, stream, wo.idx, N, wo.hash, cls_id, wo.vflag, wo.pflag);
        inclusive_scan_u32(wo.vflag, N, wo.vrank, wo.sums, tail + 0, stream);
        inclusive_scan_u32(wo.pflag, N, wo.prank, wo.sums, tail + 1, stream);
        hipLaunchKernelGGL(vox_runs_kernel, dim3(g), dim3(256), 0, stream, wo.idx, N, wo.hash, wo.vflag, wo.pflag, wo.vrank, wo.prank, wo.start,
                           wo.uniq_hash, wo.mode_key);
        hipLaunchKernelGGL(vox_vote_kernel, dim3(g), dim3(256), 0, stream, wo.idx, tail + 1, wo.start, wo.vrank, cls_id, wo.mode_key);
        hipLaunchKernelGGL(vox_label_old_kernel, dim3(g), dim3(256), 0, stream, wo.idx, N, wo.vrank, wo.mode_key, updated_orig);
    }
    if (M > 0) {
        const unsigned g = (unsigned)ceil_div(M, (int64_t)256);
        hipLaunchKernelGGL(vox_hash_kernel, dim3(g), dim3(256), 0, stream, new_xyz, M, vs_arg, use_reciprocal, minc, sy, sz, wn.hash, wn.idx);
        vox_sort_by_word(wn, M, 1, hbits < 32 ? hbits : 32, nullptr, stream);
        vox_sort_by_word(wn, M, 2, hbits - 32, nullptr, stream);
        hipLaunchKernelGGL(vox_match_new_kernel, dim3(g), dim3(256), 0, stream, wn.idx, M, wn.hash, wo.uniq_hash, N > 0 ? tail + 0 : nullptr,
                           wn.vflag, wn.pflag);
        inclusive_scan_u32(wn.pflag, M, wn.prank, wn.sums, tail + 2, stream);
        hipLaunchKernelGGL(vox_label_new_kernel, dim3(g), dim3(256), 0, stream, wn.idx, M, wn.vflag, wn.prank, wo.mode_key,
                           N > 0 ? max_cls + 1 : (int64_t)0, updated_new);
        hipLaunchKernelGGL(vox_store_count_kernel, dim3(1), dim3(1), 0, stream, tail + 2, new_voxel_count);
    } else {
        hipLaunchKernelGGL(vox_store_count_kernel, dim3(1), dim3(1), 0, stream, (const uint32_t*)nullptr, new_voxel_count);
    }
    ADK_RETURN_LAST_ERROR();
}

// Stage 2 again, for ANOTHER batch of new points against the SAME old points: the voxel table (distinct voxel hashes + each voxel's
// majority class) that a previous adk_voxel_assign left in `workspace` is searched, nothing of the N old points is touched.
// Valid only while everything that table was built from is unchanged -- xyz, cls_id, voxel_size, use_reciprocal, minc, nx / ny / nz,
// max_cls (the caller compares the bounds of the new call with those of the first) -- and the workspace has not been written since.
// SceneModel.add_new_gaussians calls update_voxel once per LoD level with the same map and, unless a level changed a label, the same
// class ids (h3dgsv3.py:884-887): the 7 radix passes over the map's points are then needed once per frame, not four times.
extern "C" int adk_voxel_assign_new(int64_t N, const float* new_xyz, int64_t M, float voxel_size, int use_reciprocal, const float* minc,
                                    int64_t nx, int64_t ny, int64_t nz, int64_t max_cls, int64_t* updated_new, int64_t* new_voxel_count,
                                    void* workspace, int64_t workspace_bytes, hipStream_t stream)
{
    using namespace adk;
    if (N <= 0 || M < 0 || !(voxel_size > 0.f) || !minc || !new_voxel_count || !workspace) return ADK_EINVAL;
    if (M > 0 && (!new_xyz || !updated_new)) return ADK_EINVAL;
    if (nx <= 0 || ny <= 0 || nz <= 0 || N >= ((int64_t)1 << 31) || M >= ((int64_t)1 << 31)) return ADK_EINVAL;
    if (workspace_bytes < adk_voxel_workspace_bytes(N, M) || ((uintptr_t)workspace & 255)) return ADK_EWORKSPACE;
    const long double cells = (long double)nx * (long double)ny * (long double)nz;
    if (cells >= 9.0e18L || max_cls >= ((int64_t)1 << 31) || max_cls < 0) return ADK_EUNSUPPORTED;
    const unsigned long long max_hash = (unsigned long long)nx * (unsigned long long)ny * (unsigned long long)nz - 1ull;
    const int hbits = bits_of(max_hash);
    const long long sy = (long long)(ny * nz), sz = (long long)nz;
    const float vs_arg = use_reciprocal ? 1.0f / voxel_size : voxel_size;
    VoxWs wo = vox_carve(workspace, N);
    uint32_t* tail = (uint32_t*)((char*)workspace + vox_ws_bytes(N));
    VoxWs wn = vox_carve((char*)workspace + vox_ws_bytes(N) + 256, M);
    if (M > 0) {
        const unsigned g = (unsigned)ceil_div(M, (int64_t)256);
        hipLaunchKernelGGL(vox_hash_kernel, dim3(g), dim3(256), 0, stream, new_xyz, M, vs_arg, use_reciprocal, minc, sy, sz, wn.hash, wn.idx);
        vox_sort_by_word(wn, M, 1, hbits < 32 ? hbits : 32, nullptr, stream);
        vox_sort_by_word(wn, M, 2, hbits - 32, nullptr, stream);
        hipLaunchKernelGGL(vox_match_new_kernel, dim3(g), dim3(256), 0, stream, wn.idx, M, wn.hash, wo.uniq_hash, tail + 0, wn.vflag, wn.pflag);
        inclusive_scan_u32(wn.pflag, M, wn.prank, wn.sums, tail + 2, stream);
        hipLaunchKernelGGL(vox_label_new_kernel, dim3(g), dim3(256), 0, stream, wn.idx, M, wn.vflag, wn.prank, wo.mode_key, max_cls + 1, updated_new);
        hipLaunchKernelGGL(vox_store_count_kernel, dim3(1), dim3(1), 0, stream, tail + 2, new_voxel_count);
    } else {
        hipLaunchKernelGGL(vox_store_count_kernel, dim3(1), dim3(1), 0, stream, (const uint32_t*)nullptr, new_voxel_count);
    }
    ADK_RETURN_LAST_ERROR();
}
