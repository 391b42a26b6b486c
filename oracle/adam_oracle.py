"""CPU oracle for adamUpdate / adamUpdateBasic -- TEST INFRASTRUCTURE, never on the product path.

The native source (on-the-fly-nvs fork of diff_gaussian_rasterization) is NOT in
/root/reference (README.md:79-86; Reconstruct/requirements.txt:6 points at an
absent directory), so the algorithm is restated from the call-site contract
Reconstruct/scene/optimizers.py:41-57 (dense) and :106-161 (row-gated), using the
published Taming-3DGS sparse-Adam kernel the fork derives from:

    m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ; p += -lr*m / (sqrt(v)+eps)

no bias correction; rows with visible==False keep param, m and v untouched.
Every operation is a single IEEE fp32 op in this order (no FMA), which is what
artdeco_amd/csrc/adam.hip is compiled to (-ffp-contract=off) => bit-exact check.

Parity: UNPINNED by the reference (no test or golden vector exists for this op).
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def adam_update_oracle(param, grad, exp_avg, exp_avg_sq, visible, lr, b1, b2, eps, N, M):
    """Returns new (param, exp_avg, exp_avg_sq) as fp32 arrays of the input shape."""
    shape = param.shape
    p = np.ascontiguousarray(param, dtype=f32).reshape(N, M).copy()
    g = np.ascontiguousarray(grad, dtype=f32).reshape(N, M)
    m = np.ascontiguousarray(exp_avg, dtype=f32).reshape(N, M).copy()
    v = np.ascontiguousarray(exp_avg_sq, dtype=f32).reshape(N, M).copy()
    vis = np.asarray(visible).astype(bool).reshape(N)
    lr = np.asarray(lr, dtype=f32)
    if lr.size == 1:
        lr_e = np.full((N, M), lr.reshape(-1)[0], dtype=f32)
    elif lr.size == N * M:
        lr_e = lr.reshape(N, M)
    elif lr.size == N:
        lr_e = np.repeat(lr.reshape(N, 1), M, axis=1)
    else:
        raise ValueError("lr must have 1, N or N*M elements")
    b1, b2, eps = f32(b1), f32(b2), f32(eps)
    omb1, omb2 = f32(1.0) - b1, f32(1.0) - b2
    m_new = b1 * m + omb1 * g
    v_new = b2 * v + (omb2 * g) * g
    step = ((-lr_e) * m_new) / (np.sqrt(v_new) + eps)
    p_new = p + step
    p[vis], m[vis], v[vis] = p_new[vis], m_new[vis], v_new[vis]
    return p.reshape(shape), m.reshape(shape), v.reshape(shape)


def adam_update_basic_oracle(param, grad, exp_avg, exp_avg_sq, lr, b1, b2, eps):
    n = int(np.asarray(param).size)
    return adam_update_oracle(param, grad, exp_avg, exp_avg_sq, np.ones(n, bool), f32(lr), b1, b2, eps, n, 1)
