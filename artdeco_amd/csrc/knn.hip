// Exact K-nearest-neighbour search (squared distances) for gfx950: simple-knn's three entry points.
//
// Replaces SimpleKNN::knn / knn_index2 / knn_indexQ --
// Reconstruct/submodules/simple-knn/simple_knn.cu:188-227, :468-522, :592-651 (bound as distCUDA2,
// distIndex2, distIndexQ in spatial.cu:14-58, ext.cpp:15-19).  Results are the exact K nearest
// neighbours (self excluded by index, distance d = dx*dx + dy*dy + dz*dz as at :400-401), which is
// all the reference's Morton/box machinery computes; the machinery itself is redesigned:
//
//   reference : Morton sort, boxes of 128 points, then EVERY query walks ALL boxes (O(P^2/128) box
//               tests), candidates fetched through an index indirection, K-best kept in global
//               memory, two blocking device->host copies for the bounding box.
//   here      : bounding box reduced on the device (no host round trip); Morton sort with the
//               shared wave64 radix sort; points GATHERED into Morton order as float4 (xyz + id) so
//               every candidate read is a coalesced/broadcast 16 B load; a two-level AABB hierarchy
//               (boxes of 64 points = one wavefront, super-boxes of 64 boxes) pruned against the
//               current K-th distance -- O(P/4096 + ~10*64 + ~15*64) tests per query instead of
//               O(P/128); K-best in registers.  One wavefront = the 64 queries of one box, so its
//               lanes prune alike and read the same candidates.
//
// Compiled with -ffp-contract=off: distances are bit-identical to the numpy oracle.
#include "adk_common.hpp"
#include "radix_sort.hpp"
#include <float.h>

namespace adk {

#define KNN_BOX 64
#define KNN_SUPER 64 // boxes per super-box

struct Aabb { float lo[3], hi[3]; };

__device__ __forceinline__ float atomicMinF(float* a, float v) {
    // valid for any sign: positive floats order like ints, negative ones like reversed uints
    return (v >= 0.f) ? __int_as_float(atomicMin((int*)a, __float_as_int(v)))
                      : __uint_as_float(atomicMax((unsigned*)a, __float_as_uint(v)));
}
__device__ __forceinline__ float atomicMaxF(float* a, float v) {
    return (v >= 0.f) ? __int_as_float(atomicMax((int*)a, __float_as_int(v)))
                      : __uint_as_float(atomicMin((unsigned*)a, __float_as_uint(v)));
}

// bbox[0..2] = min, bbox[3..5] = max over the selected points.  Like the reference's
// DeviceReduce with init {0,0,0} (simple_knn.cu:194-203) the origin is part of the box; it only
// affects the Morton grid, never the result.
__global__ __launch_bounds__(256) void knn_bbox_init_kernel(float* bbox) {
    if (threadIdx.x < 6) bbox[threadIdx.x] = 0.f;
}
__global__ __launch_bounds__(256) void knn_bbox_kernel(const float* __restrict__ pts, const int32_t* __restrict__ sel,
                                                       int M, float* __restrict__ bbox)
{
    float lo[3] = {0.f, 0.f, 0.f}, hi[3] = {0.f, 0.f, 0.f};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) {
        const int64_t g = sel ? sel[i] : i;
#pragma unroll
        for (int a = 0; a < 3; ++a) { const float v = pts[3 * g + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64)); }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { atomicMinF(bbox + a, lo[a]); atomicMaxF(bbox + 3 + a, hi[a]); }
    }
}

__device__ __forceinline__ uint32_t prep_morton(uint32_t x) {
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}
__device__ __forceinline__ uint32_t morton_of(float x, float y, float z, const float* bbox) {
    // simple_knn.cu:56-63 (10 bits per axis); a degenerate axis maps to 0 instead of NaN
    float e[3] = {bbox[3] - bbox[0], bbox[4] - bbox[1], bbox[5] - bbox[2]};
    const float p[3] = {x, y, z};
    uint32_t c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float t = e[a] > 0.f ? (p[a] - bbox[a]) / e[a] : 0.f;
        t = fminf(fmaxf(t, 0.f), 1.f) * 1023.f;
        c[a] = prep_morton((uint32_t)t);
    }
    return c[0] | (c[1] << 1) | (c[2] << 2);
}

__global__ __launch_bounds__(256) void knn_morton_kernel(const float* __restrict__ pts, const int32_t* __restrict__ sel,
                                                         int M, const float* __restrict__ bbox,
                                                         uint32_t* __restrict__ codes, uint32_t* __restrict__ ids)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const int64_t g = sel ? sel[i] : i;
    codes[i] = morton_of(pts[3 * g], pts[3 * g + 1], pts[3 * g + 2], bbox);
    ids[i] = (uint32_t)g;
}

// Gather into Morton order (float4: xyz + original index bits) and build the per-box AABBs
// (one wavefront = one box of 64).
__global__ __launch_bounds__(256) void knn_gather_kernel(const float* __restrict__ pts, const uint32_t* __restrict__ sorted_ids,
                                                         int M, float4* __restrict__ spts, Aabb* __restrict__ boxes)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (i < M) {
        const uint32_t g = sorted_ids[i];
        const float x = pts[3 * (int64_t)g], y = pts[3 * (int64_t)g + 1], z = pts[3 * (int64_t)g + 2];
        spts[i] = make_float4(x, y, z, __uint_as_float(g));
        lo[0] = hi[0] = x; lo[1] = hi[1] = y; lo[2] = hi[2] = z;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64)); }
    }
    const int box = i >> 6;
    if ((threadIdx.x & 63) == 0 && box * KNN_BOX < M) {
        Aabb b;
#pragma unroll
        for (int a = 0; a < 3; ++a) { b.lo[a] = lo[a]; b.hi[a] = hi[a]; }
        boxes[box] = b;
    }
}

__global__ __launch_bounds__(64) void knn_superbox_kernel(const Aabb* __restrict__ boxes, int n_boxes, Aabb* __restrict__ supers)
{
    const int b = blockIdx.x * KNN_SUPER + threadIdx.x;
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (b < n_boxes) {
        const Aabb x = boxes[b];
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = x.lo[a]; hi[a] = x.hi[a]; }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64)); }
    }
    if (threadIdx.x == 0) {
        Aabb s;
#pragma unroll
        for (int a = 0; a < 3; ++a) { s.lo[a] = lo[a]; s.hi[a] = hi[a]; }
        supers[blockIdx.x] = s;
    }
}

// simple_knn.cu:122-132
__device__ __forceinline__ float dist_box_point(const Aabb& b, float x, float y, float z) {
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (x < b.lo[0] || x > b.hi[0]) dx = fminf(fabsf(x - b.lo[0]), fabsf(x - b.hi[0]));
    if (y < b.lo[1] || y > b.hi[1]) dy = fminf(fabsf(y - b.lo[1]), fabsf(y - b.hi[1]));
    if (z < b.lo[2] || z > b.hi[2]) dz = fminf(fabsf(z - b.lo[2]), fabsf(z - b.hi[2]));
    return dx * dx + dy * dy + dz * dz;
}

template <int K>
struct KBest {
    float d[K];
    int id[K];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int j = 0; j < K; ++j) { d[j] = FLT_MAX; id[j] = -1; }
    }
    __device__ __forceinline__ float reject() const { return d[K - 1]; }
    // sorted insertion; a candidate equal to the current K-th is rejected (`dist >= reject`, :402)
    __device__ __forceinline__ void offer(float dist, int idx) {
        if (!(dist < d[K - 1])) return;
#pragma unroll
        for (int j = K - 1; j >= 0; --j) {
            if (j > 0 && dist < d[j - 1]) { d[j] = d[j - 1]; id[j] = id[j - 1]; }
            else { d[j] = dist; id[j] = idx; break; }
        }
    }
};

template <int K>
__device__ __forceinline__ int scan_box(const float4* __restrict__ spts, int M, int box, float x, float y, float z,
                                        uint32_t self_id, KBest<K>& kb)
{
    const int i0 = box * KNN_BOX;
    const int i1 = min(M, i0 + KNN_BOX);
    for (int i = i0; i < i1; ++i) {
        const float4 c = spts[i];
        const uint32_t cid = __float_as_uint(c.w);
        if (cid == self_id) continue;
        const float dx = c.x - x, dy = c.y - y, dz = c.z - z;
        kb.offer(dx * dx + dy * dy + dz * dz, (int)cid);
    }
    return i1 - i0; // points visited (the measurement variant counts them)
}

// MODE 0: queries are the structure's own points, query q = Morton position q (home box known).
// MODE 1: queries are arbitrary points (q_sel indices into pts); home box by binary search of the code.
// OUT 0: write K (dist, index) pairs at out row; OUT 1: write mean of the K(=3) distances.
// COUNT (measurement only, adk_knn_index2_stats): stats[0] += boxes scanned, [1] += points visited, [2] += super-box tests, [3] += box tests --
// "runtime is search-bound, so also report points-visited/query".  The product entry points instantiate COUNT = false.
template <int K, int MODE, int OUT, bool COUNT = false>
__global__ __launch_bounds__(256) void knn_query_kernel(
    const float* __restrict__ pts, const int32_t* __restrict__ q_sel, int Q, const float4* __restrict__ spts, int M,
    const uint32_t* __restrict__ sorted_codes, const Aabb* __restrict__ boxes, int n_boxes,
    const Aabb* __restrict__ supers, int n_supers, const float* __restrict__ bbox, float* __restrict__ out_d,
    int32_t* __restrict__ out_i, unsigned long long* __restrict__ stats = nullptr)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    unsigned n_scanned = 0, n_points = 0, n_super_tests = 0, n_box_tests = 0;
    float x, y, z;
    uint32_t self_id;
    int home, out_row;
    if (MODE == 0) {
        const float4 p = spts[q];
        x = p.x; y = p.y; z = p.z; self_id = __float_as_uint(p.w);
        home = q >> 6;
        out_row = (int)self_id;
    } else {
        self_id = (uint32_t)q_sel[q];
        x = pts[3 * (int64_t)self_id]; y = pts[3 * (int64_t)self_id + 1]; z = pts[3 * (int64_t)self_id + 2];
        const uint32_t code = morton_of(x, y, z, bbox);
        int lo = 0, hi = M; // lower_bound
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (sorted_codes[mid] < code) lo = mid + 1; else hi = mid; }
        home = min(lo, M - 1) >> 6;
        out_row = q;
    }
    KBest<K> kb;
    kb.init();
    { const int v = scan_box<K>(spts, M, home, x, y, z, self_id, kb); if (COUNT) { n_scanned += 1; n_points += v; } }
    const int home_super = home / KNN_SUPER;
    // own super-box first (tight reject early), then the rest
    for (int pass = 0; pass < 2; ++pass) {
        for (int s = 0; s < n_supers; ++s) {
            if ((pass == 0) != (s == home_super)) continue;
            if (COUNT) n_super_tests += 1;
            if (!(dist_box_point(supers[s], x, y, z) < kb.reject())) continue;
            const int b0 = s * KNN_SUPER, b1 = min(n_boxes, b0 + KNN_SUPER);
            for (int b = b0; b < b1; ++b) {
                if (b == home) continue;
                if (COUNT) n_box_tests += 1;
                if (!(dist_box_point(boxes[b], x, y, z) < kb.reject())) continue;
                const int v = scan_box<K>(spts, M, b, x, y, z, self_id, kb);
                if (COUNT) { n_scanned += 1; n_points += v; }
            }
        }
    }
    if (OUT == 0) {
#pragma unroll
        for (int j = 0; j < K; ++j) { out_d[(int64_t)out_row * K + j] = kb.d[j]; out_i[(int64_t)out_row * K + j] = kb.id[j]; }
    } else {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < K; ++j) s += kb.d[j];
        out_d[out_row] = s / 3.0f; // boxMeanDist, simple_knn.cu:185
    }
    if (COUNT && stats) {
        atomicAdd(stats + 0, (unsigned long long)n_scanned); atomicAdd(stats + 1, (unsigned long long)n_points);
        atomicAdd(stats + 2, (unsigned long long)n_super_tests); atomicAdd(stats + 3, (unsigned long long)n_box_tests);
    }
}

__global__ __launch_bounds__(256) void knn_fill_empty_kernel(float* __restrict__ d, int32_t* __restrict__ idx, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { d[i] = FLT_MAX; idx[i] = -1; }
}

static inline int64_t a256(int64_t x) { return (x + 255) & ~(int64_t)255; }

struct KnnWs {
    float* bbox; uint32_t *k0, *v0, *k1, *v1, *scratch; float4* spts; Aabb* boxes; Aabb* supers;
    int n_boxes, n_supers;
};
static int64_t knn_ws_bytes(int64_t M) {
    const int64_t m = M > 0 ? M : 1;
    const int64_t nb = ceil_div(m, KNN_BOX), ns = ceil_div(nb, KNN_SUPER);
    return 256 + 4 * a256(m * 4) + a256(radix_scratch_bytes(m)) + a256(m * 16) + a256(nb * (int64_t)sizeof(Aabb)) + a256(ns * (int64_t)sizeof(Aabb));
}
static KnnWs knn_carve(void* ws, int64_t M) {
    KnnWs w;
    char* p = (char*)ws;
    const int64_t seg = a256(M * 4);
    w.bbox = (float*)p; p += 256;
    w.k0 = (uint32_t*)p; p += seg; w.v0 = (uint32_t*)p; p += seg; w.k1 = (uint32_t*)p; p += seg; w.v1 = (uint32_t*)p; p += seg;
    w.scratch = (uint32_t*)p; p += a256(radix_scratch_bytes(M));
    w.spts = (float4*)p; p += a256(M * 16);
    w.n_boxes = (int)ceil_div(M, KNN_BOX); w.n_supers = (int)ceil_div(w.n_boxes, KNN_SUPER);
    w.boxes = (Aabb*)p; p += a256(w.n_boxes * (int64_t)sizeof(Aabb));
    w.supers = (Aabb*)p;
    return w;
}

// Build the search structure over pts[sel[0..M)] (sel == NULL: all M points).  Returns the buffer index
// (0/1) holding the sorted codes.
static int knn_build(const float* pts, const int32_t* sel, int M, KnnWs& w, const uint32_t** sorted_codes, hipStream_t stream)
{
    hipLaunchKernelGGL(knn_bbox_init_kernel, dim3(1), dim3(256), 0, stream, w.bbox);
    hipLaunchKernelGGL(knn_bbox_kernel, dim3(stream_grid(M, 256)), dim3(256), 0, stream, pts, sel, M, w.bbox);
    const int nb = (int)ceil_div(M, 256);
    // codes/ids land in (k1,v1); the sort ping-pongs (k0,v0) <-> (k1,v1)
    hipLaunchKernelGGL(knn_morton_kernel, dim3(nb), dim3(256), 0, stream, pts, sel, M, w.bbox, w.k1, w.v1);
    const int res = radix_sort_pairs(w.k1, w.v1, w.k0, w.v0, w.k1, w.v1, M, 0, 32, w.scratch, stream);
    const uint32_t* ids = res ? w.v1 : w.v0;
    *sorted_codes = res ? w.k1 : w.k0;
    hipLaunchKernelGGL(knn_gather_kernel, dim3(nb), dim3(256), 0, stream, pts, ids, M, w.spts, w.boxes);
    hipLaunchKernelGGL(knn_superbox_kernel, dim3(w.n_supers), dim3(64), 0, stream, w.boxes, w.n_boxes, w.supers);
    return (int)hipGetLastError();
}

template <int MODE, int OUT>
static int knn_query_dispatch(int K, const float* pts, const int32_t* q_sel, int Q, const KnnWs& w, int M,
                              const uint32_t* codes, float* out_d, int32_t* out_i, hipStream_t stream)
{
    const dim3 grid((unsigned)ceil_div(Q, 256)), block(256);
#define ADK_KNN_CASE(KK)                                                                                              \
    case KK: hipLaunchKernelGGL((knn_query_kernel<KK, MODE, OUT>), grid, block, 0, stream, pts, q_sel, Q, w.spts, M,   \
                                codes, w.boxes, w.n_boxes, w.supers, w.n_supers, w.bbox, out_d, out_i); break;
    switch (K) {
        ADK_KNN_CASE(1) ADK_KNN_CASE(2) ADK_KNN_CASE(3) ADK_KNN_CASE(4) ADK_KNN_CASE(5) ADK_KNN_CASE(6)
        ADK_KNN_CASE(7) ADK_KNN_CASE(8) ADK_KNN_CASE(12) ADK_KNN_CASE(16)
    default: return ADK_EUNSUPPORTED;
    }
#undef ADK_KNN_CASE
    return (int)hipGetLastError();
}

} // namespace adk

extern "C" int64_t adk_knn_workspace_bytes(int64_t n_struct_points)
{
    if (n_struct_points < 0) return ADK_EINVAL;
    return adk::knn_ws_bytes(n_struct_points);
}

// distIndex2: for every point its K nearest other points.  dists/indices [P*K], rows in input order;
// fewer than K neighbours leaves FLT_MAX / -1 in the tail (simple_knn.cu:441-442, spatial.cu:36).
extern "C" int adk_knn_index2(const float* points, int P, int K, float* dists, int32_t* indices, void* workspace,
                              int64_t workspace_bytes, hipStream_t stream)
{
    if (P < 0 || K < 0) return ADK_EINVAL;
    if (P == 0 || K == 0) return 0;
    if (!points || !dists || !indices || !workspace) return ADK_EINVAL;
    if (workspace_bytes < adk::knn_ws_bytes(P) || ((uintptr_t)workspace & 255)) return ADK_EWORKSPACE;
    adk::KnnWs w = adk::knn_carve(workspace, P);
    const uint32_t* codes;
    int rc = adk::knn_build(points, nullptr, P, w, &codes, stream);
    if (rc) return rc;
    return adk::knn_query_dispatch<0, 0>(K, points, nullptr, P, w, P, codes, dists, indices, stream);
}

// MEASUREMENT ONLY (bench_backend.py --knn): adk_knn_index2 for K = 3 with the search's work counted into stats[4] (zeroed by the caller):
// boxes scanned, points visited, super-box tests, box tests, summed over the P queries.  Same results as adk_knn_index2.
extern "C" int adk_knn_index2_stats(const float* points, int P, float* dists, int32_t* indices, void* workspace, int64_t workspace_bytes,
                                    unsigned long long* stats, hipStream_t stream)
{
    if (P <= 0 || !points || !dists || !indices || !workspace || !stats) return ADK_EINVAL;
    if (workspace_bytes < adk::knn_ws_bytes(P) || ((uintptr_t)workspace & 255)) return ADK_EWORKSPACE;
    adk::KnnWs w = adk::knn_carve(workspace, P);
    const uint32_t* codes;
    int rc = adk::knn_build(points, nullptr, P, w, &codes, stream);
    if (rc) return rc;
    hipLaunchKernelGGL((adk::knn_query_kernel<3, 0, 0, true>), dim3((unsigned)adk::ceil_div(P, 256)), dim3(256), 0, stream, points, nullptr, P, w.spts, P,
                       codes, w.boxes, w.n_boxes, w.supers, w.n_supers, w.bbox, dists, indices, stats);
    ADK_RETURN_LAST_ERROR();
}

// distCUDA2: mean of the 3 smallest squared distances per point (simple_knn.cu:150-186).
extern "C" int adk_knn_mean_dist3(const float* points, int P, float* mean_dists, void* workspace,
                                  int64_t workspace_bytes, hipStream_t stream)
{
    if (P < 0) return ADK_EINVAL;
    if (P == 0) return 0;
    if (!points || !mean_dists || !workspace) return ADK_EINVAL;
    if (workspace_bytes < adk::knn_ws_bytes(P) || ((uintptr_t)workspace & 255)) return ADK_EWORKSPACE;
    adk::KnnWs w = adk::knn_carve(workspace, P);
    const uint32_t* codes;
    int rc = adk::knn_build(points, nullptr, P, w, &codes, stream);
    if (rc) return rc;
    return adk::knn_query_dispatch<0, 1>(3, points, nullptr, P, w, P, codes, mean_dists, nullptr, stream);
}

// distIndexQ: queries points[q_idx[0..Q)], candidates restricted to points[n_idx[0..N)], self excluded
// by index; dists/indices [Q*K] in query order (simple_knn.cu:524-576, :592-651).
extern "C" int adk_knn_indexQ(const float* points, int P, const int32_t* q_idx, int Q, const int32_t* n_idx, int N,
                              int K, float* dists, int32_t* indices, void* workspace, int64_t workspace_bytes,
                              hipStream_t stream)
{
    if (P < 0 || Q < 0 || N < 0 || K < 0) return ADK_EINVAL;
    if (Q == 0 || K == 0) return 0;
    if (!points || !q_idx || !dists || !indices) return ADK_EINVAL;
    if (N == 0) { // no candidates: every slot is (FLT_MAX, -1)
        const int64_t n = (int64_t)Q * K;
        hipLaunchKernelGGL(adk::knn_fill_empty_kernel, dim3((unsigned)adk::ceil_div(n, 256)), dim3(256), 0, stream, dists, indices, n);
        ADK_RETURN_LAST_ERROR();
    }
    if (!workspace || !n_idx) return ADK_EINVAL;
    if (workspace_bytes < adk::knn_ws_bytes(N) || ((uintptr_t)workspace & 255)) return ADK_EWORKSPACE;
    adk::KnnWs w = adk::knn_carve(workspace, N);
    const uint32_t* codes;
    int rc = adk::knn_build(points, n_idx, N, w, &codes, stream);
    if (rc) return rc;
    return adk::knn_query_dispatch<1, 0>(K, points, q_idx, Q, w, N, codes, dists, indices, stream);
}
