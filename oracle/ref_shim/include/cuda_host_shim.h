// TEST INFRASTRUCTURE (oracle/): a minimal host-side stand-in for the CUDA execution model, just large enough to
// compile the reference's own `__global__` kernels (VSLAM/backend/src/matching_kernels.cu and
// Reconstruct/submodules/simple-knn/simple_knn.cu) with g++ and run them on the CPU, so that the numpy oracles and the
// HIP kernels can be pinned to the reference's source instead of to a restatement of it.  Never part of the product.
//
// Model: a launch runs its blocks one after another; the threads of a block are ucontext fibers scheduled
// round-robin, a fiber yields at __syncthreads(), so barrier-separated phases see each other's __shared__ writes
// exactly as on a GPU (a `__shared__` array is a function-local static: one block is alive at a time).
#pragma once
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <ucontext.h>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static

struct uint3 { unsigned x, y, z; };
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct int4 { int x, y, z, w; };

static uint3 threadIdx, blockIdx;
static dim3 blockDim, gridDim;

using std::abs;
inline float min(float a, float b) { return a < b ? a : b; }   // CUDA's device overloads (fminf semantics on non-NaN input)
inline float max(float a, float b) { return a > b ? a : b; }
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
inline unsigned min(unsigned a, int b) { return min(a, (unsigned)b); }   // CUDA: mixed signedness resolves to unsigned
inline unsigned min(int a, unsigned b) { return min((unsigned)a, b); }
inline unsigned max(unsigned a, int b) { return max(a, (unsigned)b); }
inline unsigned max(int a, unsigned b) { return max((unsigned)a, b); }

namespace shim {

// A fiber is RUNNABLE, parked at the block barrier (__syncthreads) or parked at a warp barrier (__syncwarp: the explicit form of the
// warp-synchronous lock step some reference kernels rely on implicitly -- gn_kernels.cu's warpReduce).  The block barrier opens when
// every live fiber has reached it; a warp barrier opens when every live lane of that warp is parked (at either kind of barrier).
enum FiberState : char { RUNNABLE = 0, WAIT_BLOCK = 1, WAIT_WARP = 2, DONE = 3 };

struct Fibers {
    static constexpr size_t STACK = 256 * 1024;
    std::vector<ucontext_t> ctx;
    std::vector<char> stacks;
    std::vector<char> state;
    ucontext_t main;
    int cur = 0;
    void (*body)(void*) = nullptr;
    void* arg = nullptr;
};
static Fibers g_f;

static void fiber_entry() {
    g_f.body(g_f.arg);
    g_f.state[g_f.cur] = DONE;
    swapcontext(&g_f.ctx[g_f.cur], &g_f.main);
}

inline void syncthreads() { g_f.state[g_f.cur] = WAIT_BLOCK; swapcontext(&g_f.ctx[g_f.cur], &g_f.main); }
inline void syncwarp() { g_f.state[g_f.cur] = WAIT_WARP; swapcontext(&g_f.ctx[g_f.cur], &g_f.main); }

inline dim3 to_dim3(dim3 d) { return d; }
inline dim3 to_dim3(long long n) { return dim3((unsigned)n); }

template <class F> static void launch(dim3 grid, dim3 block, F f) {
    const int nt = (int)(block.x * block.y * block.z);
    gridDim = grid; blockDim = block;
    g_f.ctx.resize(nt); g_f.state.assign(nt, RUNNABLE);
    if (g_f.stacks.size() < (size_t)nt * Fibers::STACK) g_f.stacks.resize((size_t)nt * Fibers::STACK);
    g_f.body = [](void* p) { (*static_cast<F*>(p))(); };
    g_f.arg = &f;
    for (unsigned bz = 0; bz < grid.z; ++bz) for (unsigned by = 0; by < grid.y; ++by) for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = {bx, by, bz};
        for (int t = 0; t < nt; ++t) {
            getcontext(&g_f.ctx[t]);
            g_f.ctx[t].uc_stack.ss_sp = g_f.stacks.data() + (size_t)t * Fibers::STACK;
            g_f.ctx[t].uc_stack.ss_size = Fibers::STACK;
            g_f.ctx[t].uc_link = nullptr;
            makecontext(&g_f.ctx[t], fiber_entry, 0);
            g_f.state[t] = RUNNABLE;
        }
        for (;;) {
            int live = 0;
            for (int t = 0; t < nt; ++t) {     // every runnable thread runs to its next barrier (or to its end), in thread order
                if (g_f.state[t] != RUNNABLE) { live += g_f.state[t] != DONE; continue; }
                g_f.cur = t;
                threadIdx = {(unsigned)t % block.x, ((unsigned)t / block.x) % block.y, (unsigned)t / (block.x * block.y)};
                swapcontext(&g_f.main, &g_f.ctx[t]);
                live += g_f.state[t] != DONE;
            }
            if (live == 0) break;
            bool opened = false;
            for (int w0 = 0; w0 < nt; w0 += 32) {   // warp barriers
                bool all_parked = true, any_warp = false;
                for (int t = w0; t < std::min(nt, w0 + 32); ++t) {
                    all_parked = all_parked && g_f.state[t] != RUNNABLE;
                    any_warp = any_warp || g_f.state[t] == WAIT_WARP;
                }
                if (all_parked && any_warp) {
                    for (int t = w0; t < std::min(nt, w0 + 32); ++t) if (g_f.state[t] == WAIT_WARP) g_f.state[t] = RUNNABLE;
                    opened = true;
                }
            }
            if (opened) continue;
            bool all_block = true;                  // the block barrier: every live thread is there
            for (int t = 0; t < nt; ++t) all_block = all_block && (g_f.state[t] == WAIT_BLOCK || g_f.state[t] == DONE);
            if (!all_block) { std::fprintf(stderr, "cuda_host_shim: deadlock (threads parked at different barriers)\n"); std::abort(); }
            for (int t = 0; t < nt; ++t) if (g_f.state[t] == WAIT_BLOCK) g_f.state[t] = RUNNABLE;
        }
    }
}

} // namespace shim

#define __syncwarp() shim::syncwarp()
#define __syncthreads() shim::syncthreads()

// ---- the slice of the CUDA runtime / thrust / cub / cooperative-groups API the two files use ----
enum cudaMemcpyKind { cudaMemcpyHostToDevice, cudaMemcpyDeviceToHost, cudaMemcpyDeviceToDevice };
template <class T> inline int cudaMalloc(T** p, size_t n) { *p = (T*)std::malloc(n); return 0; }
inline int cudaFree(void* p) { std::free(p); return 0; }
inline int cudaMemcpy(void* d, const void* s, size_t n, cudaMemcpyKind) { std::memcpy(d, s, n); return 0; }

namespace cooperative_groups {
struct grid_group { unsigned long long thread_rank() const { return (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; } };
inline grid_group this_grid() { return {}; }
}

namespace thrust {
template <class T> struct device_ptr { T* p; T* get() const { return p; } };
template <class T> class device_vector {   // not std::vector: vector<bool> has no data()
    T* p_ = nullptr; size_t n_ = 0;
public:
    explicit device_vector(size_t n = 0, T v = T()) { resize(n, v); }
    device_vector(const device_vector&) = delete;
    ~device_vector() { std::free(p_); }
    void resize(size_t n, T v = T()) { p_ = (T*)std::realloc(p_, n ? n * sizeof(T) : 1); for (size_t i = n_; i < n; ++i) p_[i] = v; n_ = n; }
    device_ptr<T> data() { return {p_}; }
    T* begin() { return p_; }
    T* end() { return p_ + n_; }
};
template <class It> inline void sequence(It b, It e) { std::iota(b, e, 0); }
}

namespace cub {
struct DeviceReduce {
    template <class In, class Out, class Op, class T>
    static int Reduce(void* tmp, size_t& bytes, In in, Out out, int n, Op op, T init) {
        if (!tmp) { bytes = 1; return 0; }
        T acc = init;
        for (int i = 0; i < n; ++i) acc = op(acc, in[i]);
        *out = acc;
        return 0;
    }
};
struct DeviceRadixSort {   // ascending, stable -- the contract of cub's LSD radix sort
    template <class K, class V>
    static int SortPairs(void* tmp, size_t& bytes, const K* kin, K* kout, const V* vin, V* vout, int n) {
        if (!tmp) { bytes = 1; return 0; }
        std::vector<int> order(n);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return kin[a] < kin[b]; });
        for (int i = 0; i < n; ++i) { kout[i] = kin[order[i]]; vout[i] = vin[order[i]]; }
        return 0;
    }
};
}
