// Prune-and-append of the whole Gaussian parameter set in one pass, for gfx950.
//
// Replaces the body of SparseGaussianAdam.add_and_prune (Reconstruct/scene/optimizers.py:163-219), which the mapper
// runs on every important frame (h3dgsv3.py:938, :953): for each of ~12 parameters and their two Adam moments (and the
// per-element learning rate of xyz) it evaluates `tensor[valid_mask]` -- a nonzero() with a host sync, an index gather
// -- and a torch.cat with the new rows: ~40 stream drains and ~100 launches moving 3 x 75 floats per Gaussian.
// Here the keep mask is scanned ONCE (per-workgroup counts, one small scan), the host reads the number of kept rows
// once to size the outputs, and ONE launch writes every output tensor: kept rows compacted in order (same order as
// boolean indexing), then the appended rows (copied from the extension tensor, or a fill value: 0 for the moments,
// lr_init for the learning rate).  Rows are moved as 4-byte words so int64 ids and fp32 parameters share the kernel.
#include "adk_common.hpp"

namespace adk {

#define CMP_ROWS 256      // rows per workgroup
#define CMP_MAX_TENSORS 48

__global__ __launch_bounds__(256) void compact_count_kernel(const uint8_t* __restrict__ keep, int64_t N, uint32_t* __restrict__ block_counts)
{
    __shared__ uint32_t wc[4];
    const int64_t r = (int64_t)blockIdx.x * CMP_ROWS + threadIdx.x;
    const bool k = r < N && keep[r] != 0;
    const unsigned long long m = __ballot(k);
    if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = (wc[0] + wc[1]) + (wc[2] + wc[3]);
}

// exclusive scan of block_counts in place (one workgroup), total -> *n_keep
__global__ __launch_bounds__(1024) void compact_scan_kernel(uint32_t* __restrict__ block_counts, int nb, int64_t* __restrict__ n_keep)
{
    __shared__ uint32_t part[1024];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < nb ? block_counts[i] : 0u;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) { // Hillis-Steele inclusive scan
            const uint32_t t = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0u;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nb) block_counts[i] = carry + part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_keep = (int64_t)carry;
}

struct CompactArgs {
    const uint32_t* src[CMP_MAX_TENSORS];  // [N, words]
    const uint32_t* ext[CMP_MAX_TENSORS];  // [E, words] or null => fill
    uint32_t* dst[CMP_MAX_TENSORS];        // [K + E, words]
    uint32_t fill[CMP_MAX_TENSORS];        // bit pattern for appended rows without an extension tensor
    int words[CMP_MAX_TENSORS];            // 4-byte words per row
    int64_t N, E;
    const uint8_t* keep;
    const uint32_t* block_offs;            // exclusive scan of the per-workgroup kept counts
    const int64_t* n_keep;
    int prune_blocks;                      // workgroups [0, prune_blocks) compact, the rest append
};

// grid (x: row blocks of the pruned part, then row blocks of the appended part; y: tensor)
__global__ __launch_bounds__(256) void compact_apply_kernel(CompactArgs a)
{
    __shared__ uint32_t wbase[4];
    __shared__ int64_t spos[CMP_ROWS];
    const int t = blockIdx.y, W = a.words[t];
    uint32_t* __restrict__ dst = a.dst[t];
    if ((int)blockIdx.x < a.prune_blocks) {
        const int64_t r0 = (int64_t)blockIdx.x * CMP_ROWS, r = r0 + threadIdx.x;
        const bool k = r < a.N && a.keep[r] != 0;
        const unsigned long long m = __ballot(k);
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane == 0) wbase[wv] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = 0;
        for (int w = 0; w < wv; ++w) before += wbase[w];
        const uint32_t local = before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        spos[threadIdx.x] = k ? (int64_t)a.block_offs[blockIdx.x] + local : -1; // destination row, order preserved
        __syncthreads();
        const uint32_t* __restrict__ src = a.src[t] + r0 * W;
        const int rows = (int)min((int64_t)CMP_ROWS, a.N - r0), total = rows * W;
        for (int i = threadIdx.x; i < total; i += 256) { // coalesced read of the slab; kept rows land contiguously
            const int rr = i / W, w = i - rr * W;
            const int64_t p = spos[rr];
            if (p >= 0) dst[p * W + w] = src[i];
        }
    } else {
        const int64_t K = *a.n_keep;
        const int64_t j0 = (int64_t)((int)blockIdx.x - a.prune_blocks) * CMP_ROWS;
        const int rows = (int)min((int64_t)CMP_ROWS, a.E - j0), total = rows * W;
        const uint32_t* __restrict__ ext = a.ext[t];
        uint32_t* out = dst + (K + j0) * W;
        for (int i = threadIdx.x; i < total; i += 256) out[i] = ext ? ext[j0 * W + i] : a.fill[t];
    }
}

} // namespace adk

static inline int64_t cmp_blocks(int64_t n) { return (n + CMP_ROWS - 1) / CMP_ROWS; }

// workspace: per-workgroup kept counts / offsets (u32 per 256 rows)
extern "C" int64_t adk_compact_workspace_bytes(int64_t N)
{
    if (N < 0) return ADK_EINVAL;
    return (cmp_blocks(N) + 1) * (int64_t)sizeof(uint32_t) + 256;
}

// Step 1: scan the keep mask (bytes, torch.bool storage).  *n_keep (int64, device) = number of kept rows; the
// workspace then holds what adk_compact_apply needs.  The caller reads n_keep to size the outputs.
extern "C" int adk_compact_plan(int64_t N, const uint8_t* keep, int64_t* n_keep, void* workspace, int64_t workspace_bytes,
                                hipStream_t stream)
{
    if (N < 0 || !n_keep) return ADK_EINVAL;
    if (N >= ((int64_t)1 << 31)) return ADK_EUNSUPPORTED;
    if (N == 0) return adk::clear_bytes(n_keep, sizeof(int64_t), stream);
    if (!keep || !workspace) return ADK_EINVAL;
    if (workspace_bytes < adk_compact_workspace_bytes(N)) return ADK_EWORKSPACE;
    const int nb = (int)cmp_blocks(N);
    uint32_t* counts = (uint32_t*)workspace;
    hipLaunchKernelGGL(adk::compact_count_kernel, dim3(nb), dim3(256), 0, stream, keep, N, counts);
    hipLaunchKernelGGL(adk::compact_scan_kernel, dim3(1), dim3(1024), 0, stream, counts, nb, n_keep);
    ADK_RETURN_LAST_ERROR();
}

// Step 2: for each of n_tensors tensors, dst[t] [K + E, words[t]] = concat(src[t][keep], ext[t] or fill[t]) -- the
// `torch.cat([x[valid_mask], extension])` of optimizers.py:197-219 for every parameter, moment and learning rate in
// one launch.  words[t]: 4-byte words per row (2 per int64 element).  ext[t] == NULL appends E rows of fill_bits[t]
// (0 for the moments, the bits of lr_init for a learning rate).  n_keep: the device value adk_compact_plan wrote.
extern "C" int adk_compact_apply(int n_tensors, const void* const* src, const void* const* ext, void* const* dst,
                                 const uint32_t* fill_bits, const int* words, int64_t N, int64_t E, const uint8_t* keep,
                                 const int64_t* n_keep, const void* workspace, hipStream_t stream)
{
    if (n_tensors < 0 || N < 0 || E < 0) return ADK_EINVAL;
    if (n_tensors == 0 || (N == 0 && E == 0)) return 0;
    if (n_tensors > CMP_MAX_TENSORS) return ADK_EUNSUPPORTED;
    if (!src || !ext || !dst || !fill_bits || !words || !n_keep || (N > 0 && (!keep || !workspace))) return ADK_EINVAL;
    adk::CompactArgs a;
    for (int t = 0; t < n_tensors; ++t) {
        if (!dst[t] || words[t] <= 0 || (N > 0 && !src[t])) return ADK_EINVAL;
        a.src[t] = (const uint32_t*)src[t]; a.ext[t] = (const uint32_t*)ext[t]; a.dst[t] = (uint32_t*)dst[t];
        a.fill[t] = fill_bits[t]; a.words[t] = words[t];
    }
    a.N = N; a.E = E; a.keep = keep; a.block_offs = (const uint32_t*)workspace; a.n_keep = n_keep;
    a.prune_blocks = (int)cmp_blocks(N);
    const int total_blocks = a.prune_blocks + (int)cmp_blocks(E);
    if (total_blocks == 0) return 0;
    hipLaunchKernelGGL(adk::compact_apply_kernel, dim3((unsigned)total_blocks, (unsigned)n_tensors), dim3(256), 0, stream, a);
    ADK_RETURN_LAST_ERROR();
}
