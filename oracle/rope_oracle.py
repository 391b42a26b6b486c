"""CPU oracle for curope.rope_2d -- TEST INFRASTRUCTURE only.

Restates the reference's own CPU branch, rope_2d_cpu (croco/models/curope/curope.cpp:11-47):
for each half x in {y-part, x-part}, pair d < D/4: angle = fwd * pos / base^(d/(D/4));
(u, v) = (tok[d + x*D/2], tok[d + D/4 + x*D/2]) -> (u cos - v sin, v cos + u sin).

Parity: PINNED -- tests/golden/rope_*.npz are produced by the reference's pure-torch RoPE2D module
(croco/models/pos_embed.py:112-158), see tests/golden/make_golden_rope.py.
"""
from __future__ import annotations

import numpy as np


def rope_2d_oracle(tokens: np.ndarray, positions: np.ndarray, base: float, fwd: float) -> np.ndarray:
    """tokens [B,N,H,D] float32, positions [B,N,2] int64 -> rotated copy (float32 math)."""
    tok = np.array(tokens, dtype=np.float32, copy=True)
    B, N, H, D4 = tok.shape
    Q = D4 // 4
    d = np.arange(Q, dtype=np.float32)
    denom = np.power(np.float32(base), d / np.float32(Q)).astype(np.float32)
    for x in range(2):
        p = positions[:, :, x].astype(np.float32)[:, :, None]           # [B,N,1]
        ang = (np.float32(fwd) * p / denom[None, None, :]).astype(np.float32)  # [B,N,Q]
        c, s = np.cos(ang)[:, :, None, :], np.sin(ang)[:, :, None, :]
        u = tok[..., x * 2 * Q: x * 2 * Q + Q].copy()
        v = tok[..., x * 2 * Q + Q: x * 2 * Q + 2 * Q].copy()
        tok[..., x * 2 * Q: x * 2 * Q + Q] = u * c - v * s
        tok[..., x * 2 * Q + Q: x * 2 * Q + 2 * Q] = v * c + u * s
    return tok
