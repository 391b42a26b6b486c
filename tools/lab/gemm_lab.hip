// DEV TOOL (prototype): C[M,N] = A[M,K] W[N,K]^T + bias, fp16 operands, fp32 accumulation, fp16 output -- the nn.Linear of the MASt3R
// blocks at M = 768 tokens, where hipBLASLt's 64x64-tile kernels take 10-20 us whatever the shape (profiles/r02_frontend_kernel_stats.csv).
// v_mfma_f32_32x32x16_f16; both operands are K-contiguous in memory, so an MFMA fragment (row l & 31, 8 halves at k = 8 (l >> 5)) is one
// 16-byte piece of a row: no transposes anywhere.  Workgroup = 4 waves (2 x 2) on a BM x BN tile, BK = 32 per step, LDS tiles stored as
// [k-chunk][row][16 B] planes (plane stride 2048 + 64 B): fragment reads (ds_read_b128) and staging writes are both conflict-free; the
// next step's global loads are in flight in registers while the current one is multiplied, two LDS buffers, one barrier per step.
// SPLITK > 1: blockIdx.z owns a K range and adds its partial tile into a zeroed fp32 workspace; a second kernel adds the bias and converts.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_tn_f16_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ W,
                                                          const _Float16* __restrict__ bias, _Float16* __restrict__ C,
                                                          float* __restrict__ partial, int M, int N, int K, int k_per_split)
{
    constexpr int BK = 32, WM = BM / 2, WN = BN / 2, MT = WM / 32, NT = WN / 32;      // wave tile WM x WN = MT x NT MFMA tiles
    constexpr int PA = BM * 8 + 32, PW = BN * 8 + 32;                                  // plane strides in halves (+64 B skew)
    constexpr int CA = BM * 4 / 256, CW = BN * 4 / 256;                                // 16-byte chunks per thread per step
    __shared__ __attribute__((aligned(16))) _Float16 sA[2][4 * PA];
    __shared__ __attribute__((aligned(16))) _Float16 sW[2][4 * PW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int k_begin = blockIdx.z * k_per_split, k_end = min(K, k_begin + k_per_split);

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f16x8 ra[CA], rw[CW];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int c = 0; c < CA; ++c) {
            const int id = tid + 256 * c, row = id >> 2, p = id & 3;
            const int gr = min(m0 + row, M - 1);
            ra[c] = *reinterpret_cast<const f16x8*>(A + (int64_t)gr * K + k0 + 8 * p);
        }
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            const int id = tid + 256 * c, row = id >> 2, p = id & 3;
            const int gr = min(n0 + row, N - 1);
            rw[c] = *reinterpret_cast<const f16x8*>(W + (int64_t)gr * K + k0 + 8 * p);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int c = 0; c < CA; ++c) {
            const int id = tid + 256 * c, row = id >> 2, p = id & 3;
            *reinterpret_cast<f16x8*>(&sA[buf][p * PA + row * 8]) = ra[c];
        }
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            const int id = tid + 256 * c, row = id >> 2, p = id & 3;
            *reinterpret_cast<f16x8*>(&sW[buf][p * PW + row * 8]) = rw[c];
        }
    };

    const int g = lane >> 5, i32 = lane & 31;
    fetch(k_begin);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        const bool more = k0 + BK < k_end;
        if (more) fetch(k0 + BK);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 af[MT], bf[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const f16x8*>(&sA[buf][(2 * ks + g) * PA + (wm * WM + 32 * i + i32) * 8]);
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const f16x8*>(&sW[buf][(2 * ks + g) * PW + (wn * WN + 32 * j + i32) * 8]);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            stash(buf ^ 1);      // the other buffer: nobody reads it during this step
            __syncthreads();
            buf ^= 1;
        }
    }

    // epilogue.  D layout of 32x32: lane l, register r -> row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = n0 + wn * WN + 32 * j + i32;
            if (col >= N) continue;
            const float b = (partial || !bias) ? 0.f : (float)bias[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * WM + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * g;
                if (row >= M) continue;
                if (partial) atomicAdd(&partial[(int64_t)row * N + col], acc[i][j][r]);
                else C[(int64_t)row * N + col] = (_Float16)(acc[i][j][r] + b);
            }
        }
}

__global__ __launch_bounds__(256) void gemm_finish_kernel(const float* __restrict__ partial, const _Float16* __restrict__ bias,
                                                          _Float16* __restrict__ C, int64_t total, int N)
{
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= total) return;
    const float4 p = *reinterpret_cast<const float4*>(partial + i);
    const int col = (int)(i % N);
    f16x4 o;
    o[0] = (_Float16)(p.x + (bias ? (float)bias[col] : 0.f)); o[1] = (_Float16)(p.y + (bias ? (float)bias[col + 1] : 0.f));
    o[2] = (_Float16)(p.z + (bias ? (float)bias[col + 2] : 0.f)); o[3] = (_Float16)(p.w + (bias ? (float)bias[col + 3] : 0.f));
    *reinterpret_cast<f16x4*>(C + i) = o;
}

__global__ __launch_bounds__(256) void zero_kernel(float4* p, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// variant: 0 = 128x128, 1 = 64x128, 2 = 128x64, 3 = 64x64; splitk >= 1 (workspace [M,N] fp32 needed when > 1).  K % 32 == 0.
extern "C" int gemm_lab(int variant, int splitk, const void* A, const void* W, const void* bias, void* C, float* workspace, int M, int N, int K,
                        hipStream_t st)
{
    if (K % 32 || N % 4) return -1;
    const _Float16 *a = (const _Float16*)A, *w = (const _Float16*)W, *b = (const _Float16*)bias;
    _Float16* c = (_Float16*)C;
    int kps = K;
    float* part = nullptr;
    if (splitk > 1) {
        kps = ((K / 32 + splitk - 1) / splitk) * 32;
        part = workspace;
        const int64_t n4 = (int64_t)M * N / 4;
        hipLaunchKernelGGL(zero_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, (float4*)workspace, n4);
    }
#define LAUNCH(BM, BN) hipLaunchKernelGGL((gemm_tn_f16_kernel<BM, BN>), dim3((N + BN - 1) / BN, (M + BM - 1) / BM, splitk), dim3(256), 0, st, a, w, b, c, part, M, N, K, kps)
    switch (variant) {
    case 0: LAUNCH(128, 128); break;
    case 1: LAUNCH(64, 128); break;
    case 2: LAUNCH(128, 64); break;
    case 3: LAUNCH(64, 64); break;
    default: return -2;
    }
#undef LAUNCH
    if (splitk > 1) {
        const int64_t total = (int64_t)M * N;
        hipLaunchKernelGGL(gemm_finish_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, st, workspace, b, c, total, N);
    }
    return (int)hipGetLastError();
}
