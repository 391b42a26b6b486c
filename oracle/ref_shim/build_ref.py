#!/usr/bin/env python
"""Recipe for oracle/_ref: the REFERENCE's own kernels, compiled for the host.  TEST INFRASTRUCTURE ONLY.

What it builds (outputs only under oracle/_ref/, which is git-ignored but travels to the GPU box):

  oracle/_ref/libref_knn.so          SimpleKNN::knn / knn_index2 / knn_indexQ of
                                     /root/reference/Reconstruct/submodules/simple-knn/simple_knn.cu  (C ABI: ref_knn*)
  oracle/_ref/ref_matching.so        iter_proj_cuda / refine_matches_cuda of
                                     /root/reference/VSLAM/backend/src/matching_kernels.cu             (pybind11 module)
  oracle/_ref/ref_gn.so              gauss_newton_points_cuda / _rays_cuda / _calib_cuda of
                                     /root/reference/VSLAM/backend/src/gn_kernels.cu                   (pybind11 module)
                                     -- the three accumulate kernels, the pose retraction and the host-side SparseBlock
                                     assembly / solve, with oracle/ref_shim/include/eigen_host_shim.h standing in for Eigen
                                     (dense fp64 LL^T for SimplicialLLT) and `torch::kCUDA` reading `torch::kCPU`.  One more
                                     mechanical edit here: gn_kernels.cu's warpReduce (:36-43) relies on the implicit lock step
                                     of a warp; `__syncwarp();` is inserted after each of its six statements so that the fiber
                                     model executes them step by step for all 32 lanes, as the hardware does.

The reference sources are compiled from where they lie: the only edit is mechanical and done in memory by this script --
`kernel<<<grid, block>>>(args)` (not C++) becomes `shim::launch(grid, block, [&]{ kernel(args); })` -- and the result is
written to oracle/_ref/gen/ (a build intermediate, never committed).  Everything CUDA-specific the files include
(`cuda_runtime.h`, `cub`, `thrust`, `cooperative_groups`, `cuda/std/limits`) resolves to oracle/ref_shim/include/, a
host stand-in that runs blocks serially and the threads of a block as fibers (so `__syncthreads()` works).  torch's real
CPU headers provide `PackedTensorAccessor32`, `AT_DISPATCH_FLOATING_TYPES_AND_HALF` and `c10::Half` (one rounding to
half per operator, which is what `scalar_t = half` arithmetic does in the reference kernel).

Arithmetic: g++ on x86-64 with -ffp-contract=off -- every fp32/fp16/fp64 operation the source writes, individually rounded.
nvcc would additionally contract a*b+c into FMAs; that is the one documented difference from the CUDA binary.

Usage:  python oracle/ref_shim/build_ref.py [--force]     (needs /root/reference; a no-op when it is absent)
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE = os.path.dirname(HERE)
OUT = os.path.join(ORACLE, "_ref")
GEN = os.path.join(OUT, "gen")
REF = os.environ.get("ARTDECO_REFERENCE", "/root/reference")
KNN_DIR = os.path.join(REF, "Reconstruct", "submodules", "simple-knn")
KNN_SRC = os.path.join(KNN_DIR, "simple_knn.cu")
MATCH_SRC = os.path.join(REF, "VSLAM", "backend", "src", "matching_kernels.cu")
GN_SRC = os.path.join(REF, "VSLAM", "backend", "src", "gn_kernels.cu")

_LAUNCH = re.compile(r"<<\s*<")
_CLOSE = re.compile(r">>\s*>")


def _match_paren(s: str, i: int) -> int:
    """index just past the parenthesis group that opens at s[i] == '('."""
    depth = 0
    for j in range(i, len(s)):
        if s[j] == "(":
            depth += 1
        elif s[j] == ")":
            depth -= 1
            if depth == 0:
                return j + 1
    raise ValueError("unbalanced parentheses after a kernel launch")


def _split_top(s: str) -> list[str]:
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([":
            depth += 1
        elif ch in ")]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    out.append(cur)
    return [x.strip() for x in out]


def rewrite_launches(src: str) -> str:
    """kernel<<<g, b>>>(args)  ->  shim::launch(shim::to_dim3(g), shim::to_dim3(b), [&]() { kernel(args); })"""
    out, pos = "", 0
    while True:
        m = _LAUNCH.search(src, pos)
        if not m:
            return out + src[pos:]
        # kernel expression: identifier (+ one template argument list) immediately before <<<
        k_end = m.start()
        k = k_end
        while k > 0 and src[k - 1].isspace():
            k -= 1
        if src[k - 1] == ">":
            depth = 0
            while True:
                k -= 1
                depth += {">": 1, "<": -1}.get(src[k], 0)
                if depth == 0:
                    break
        while k > 0 and (src[k - 1].isalnum() or src[k - 1] in "_:"):
            k -= 1
        kernel = src[k:k_end].strip()
        close = _CLOSE.search(src, m.end())
        cfg = _split_top(src[m.end():close.start()])
        a0 = src.index("(", close.end())
        a1 = _match_paren(src, a0)
        args = src[a0:a1]
        out += src[pos:k] + f"shim::launch(shim::to_dim3({cfg[0]}), shim::to_dim3({cfg[1]}), [&]() {{ {kernel}{args}; }})"
        pos = a1


KNN_MAIN = r'''
#include "cuda_host_shim.h"
#include "simple_knn.gen.cpp"
extern "C" {
void ref_knn(int P, const float* points, float* mean_dists) { SimpleKNN::knn(P, (float3*)points, mean_dists); }
void ref_knn_index2(int K, int P, const float* points, float* dists, int* indices) {
    SimpleKNN::knn_index2(K, P, (float3*)points, dists, indices);
}
void ref_knn_indexQ(int K, int P, const float* points, int Q, const int* q_idx, int N, const int* n_idx, float* dists, int* indices) {
    SimpleKNN::knn_indexQ(K, P, (float3*)points, Q, (int*)q_idx, N, (int*)n_idx, dists, indices);
}
}
'''

MATCH_MAIN = r'''
#include <torch/extension.h>
// torch only defines RestrictPtrTraits under nvcc/hipcc (torch/headeronly/core/TensorAccessor.h:22-27); same definition
namespace torch { template <typename T> struct RestrictPtrTraits { typedef T* __restrict__ PtrType; }; }
// AT_DISPATCH_*(tensor.type(), ...): the overload older torch shipped (deprecated, since removed) for DeprecatedTypeProperties
namespace detail { inline at::ScalarType scalar_type(const at::DeprecatedTypeProperties& t) { return t.scalarType(); } }
#include "matching_kernels.gen.cpp"
// the binding /root/reference/VSLAM/backend/src/gn.cpp:116-122 makes for these two entry points
PYBIND11_MODULE(ref_matching, m) {
    m.def("iter_proj", &iter_proj_cuda);
    m.def("refine_matches", &refine_matches_cuda);
}
'''


GN_MAIN = r'''
#include <torch/extension.h>
// torch::linalg::linalg_norm (gn_kernels.cu:802) of the C++ frontend's torch/linalg.h, which this wheel does not ship: the same ATen op
namespace torch { namespace linalg {
inline at::Tensor linalg_norm(const at::Tensor& x, std::optional<c10::Scalar> ord, at::OptionalIntArrayRef dim, bool keepdim,
                              std::optional<at::ScalarType> dtype) { return at::linalg_norm(x, ord, dim, keepdim, dtype); }
} }
namespace torch { template <typename T> struct RestrictPtrTraits { typedef T* __restrict__ PtrType; }; }
// the reference moves the solver's results "to the device" (gn_kernels.cu:123-149); in this host build the device is the CPU
#define kCUDA kCPU
#include "gn_kernels.gen.cpp"
// the entry points /root/reference/VSLAM/backend/src/gn.cpp:3-82 forwards to (its CHECK_INPUT guards demand CUDA tensors)
PYBIND11_MODULE(ref_gn, m) {
    m.def("gauss_newton_points", &gauss_newton_points_cuda);
    m.def("gauss_newton_rays", &gauss_newton_rays_cuda);
    m.def("gauss_newton_calib", &gauss_newton_calib_cuda);
}
'''

_WARP_STEP = re.compile(r"^(\s*sdata\[tid\] \+= sdata\[tid \+ +\d+\];)\s*$", re.M)


def explicit_warp_steps(src: str) -> str:
    """`__syncwarp();` after every statement of warpReduce (gn_kernels.cu:36-43): the warp-synchronous lock step made explicit."""
    out, n = _WARP_STEP.subn(r"\1 __syncwarp();", src)
    if n != 6:
        raise RuntimeError(f"gn_kernels.cu: expected the six steps of warpReduce, found {n}")
    return out


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle/_ref build failed:\n" + " ".join(cmd) + "\n" + r.stdout[-4000:] + r.stderr[-8000:])


def targets():
    return {"knn": os.path.join(OUT, "libref_knn.so"), "matching": os.path.join(OUT, "ref_matching.so"), "gn": os.path.join(OUT, "ref_gn.so")}


def available() -> bool:
    return all(os.path.exists(p) for p in targets().values())


def build(force: bool = False) -> bool:
    """Returns True when oracle/_ref is usable afterwards (built now or earlier)."""
    if not (os.path.exists(KNN_SRC) and os.path.exists(MATCH_SRC) and os.path.exists(GN_SRC)):
        return available()   # GPU box: /root/reference is absent, the prebuilt files travelled with the snapshot
    t = targets()
    newest_in = max(os.path.getmtime(p) for p in (KNN_SRC, MATCH_SRC, GN_SRC, __file__, os.path.join(HERE, "include", "cuda_host_shim.h"),
                                                  os.path.join(HERE, "include", "eigen_host_shim.h")))
    if not force and available() and min(os.path.getmtime(p) for p in t.values()) >= newest_in:
        return True
    os.makedirs(GEN, exist_ok=True)
    inc = os.path.join(HERE, "include")
    common = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-w", "-I" + inc, "-I" + GEN]
    with open(os.path.join(GEN, "simple_knn.gen.cpp"), "w") as f:
        f.write(rewrite_launches(open(KNN_SRC).read()))
    with open(os.path.join(GEN, "ref_knn_main.cpp"), "w") as f:
        f.write(KNN_MAIN)
    _run(common + ["-I" + KNN_DIR, os.path.join(GEN, "ref_knn_main.cpp"), "-o", t["knn"]])

    import torch
    from torch.utils import cpp_extension as ce
    with open(os.path.join(GEN, "matching_kernels.gen.cpp"), "w") as f:
        f.write(rewrite_launches(open(MATCH_SRC).read()))
    with open(os.path.join(GEN, "ref_matching_main.cpp"), "w") as f:
        f.write(MATCH_MAIN)
    tlib = ce.library_paths()[0]
    cmd = common + ["-DTORCH_EXTENSION_NAME=ref_matching", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
    cmd += ["-isystem" + p for p in ce.include_paths()] + ["-isystem" + sysconfig.get_paths()["include"]]
    cmd += [os.path.join(GEN, "ref_matching_main.cpp"), "-o", t["matching"], "-L" + tlib, "-Wl,-rpath," + tlib,
            "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python"]
    _run(cmd)

    with open(os.path.join(GEN, "gn_kernels.gen.cpp"), "w") as f:
        f.write(rewrite_launches(explicit_warp_steps(open(GN_SRC).read())))
    with open(os.path.join(GEN, "ref_gn_main.cpp"), "w") as f:
        f.write(GN_MAIN)
    cmd = common + ["-DTORCH_EXTENSION_NAME=ref_gn", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
    cmd += ["-isystem" + p for p in ce.include_paths()] + ["-isystem" + sysconfig.get_paths()["include"]]
    cmd += [os.path.join(GEN, "ref_gn_main.cpp"), "-o", t["gn"], "-L" + tlib, "-Wl,-rpath," + tlib,
            "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python"]
    _run(cmd)
    return True


if __name__ == "__main__":
    ok = build(force="--force" in sys.argv)
    print("oracle/_ref:", "ready" if ok else "not built (/root/reference absent and no prebuilt files)")
    for p in targets().values():
        print(" ", p, "ok" if os.path.exists(p) else "MISSING")
