"""ORACLE (test infrastructure only -- never imported by artdeco_amd/): the image-space chain of SceneModel.add_new_gaussians.

CPU restatement (numpy, fp32) of what the reference computes per LoD level with generic torch operators
(Reconstruct/scene/scene_models/h3dgsv3.py:775-891, Reconstruct/utils.py:93-108, 121-131, 188-216):

  avg_pool2d(image, 2)                                               h3dgsv3.py:776
  F.interpolate(.., (h, w), "bilinear", align_corners=True)          :781, :789
  get_lapla_norm: 3x3 Laplacian summed over channels, |.|, border rows / columns zeroed, 7x7 disc mean, clamp(0, 1)
                                                                     utils.py:93-108, disc kernel h3dgsv3.py:209-220
  sample(): grid_sample(bilinear, align_corners=True, zero padding) at u * 2 / (w - 1) - 1          utils.py:203-216
  torch.quantile(depth, 0.02) with linear interpolation              h3dgsv3.py:815
  depth2points, RGB2SH, inverse_sigmoid, the scale / opacity / d_max initialisation                  :847-891

Pinned: tests/golden/densify_*.npz are produced by running the reference's OWN add_new_gaussians on CPU
(tests/golden/make_golden_densify.py); tests/test_densify.py checks this restatement against them (maps <= 1e-5, identical
sample masks under the same uniform draw, emitted attributes <= 1e-5 relative) before the HIP kernels are compared with either.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
C0 = F32(0.28209479177387814)


def avg_pool2(img: np.ndarray) -> np.ndarray:
    """[C,H,W] -> [C,H//2,W//2] (F.avg_pool2d(img, 2): odd trailing row / column dropped)."""
    C, H, W = img.shape
    h, w = H // 2, W // 2
    v = img[:, :2 * h, :2 * w].astype(F32)
    return ((v[:, 0::2, 0::2] + v[:, 0::2, 1::2]) + (v[:, 1::2, 0::2] + v[:, 1::2, 1::2])) * F32(0.25)


def _axis_taps(n_in: int, n_out: int):
    """(i0, i1, w0, w1) of torch's bilinear upsample with align_corners=True: scale = (in-1)/(out-1) in fp32."""
    scale = F32(0.0) if n_out <= 1 else F32(n_in - 1) / F32(n_out - 1)
    src = (scale * np.arange(n_out, dtype=F32)).astype(F32)
    i0 = np.minimum(src.astype(np.int64), n_in - 1)
    i1 = np.minimum(i0 + 1, n_in - 1)
    w1 = (src - i0.astype(F32)).astype(F32)
    w0 = (F32(1.0) - w1).astype(F32)
    return i0, i1, w0, w1


def resize_bilinear(img: np.ndarray, h: int, w: int) -> np.ndarray:
    """F.interpolate(img[None], (h, w), mode="bilinear", align_corners=True)[0] for [C,H,W] float32."""
    C, H, W = img.shape
    y0, y1, wy0, wy1 = _axis_taps(H, h)
    x0, x1, wx0, wx1 = _axis_taps(W, w)
    img = img.astype(F32)
    top = img[:, y0][:, :, x0] * wx0 + img[:, y0][:, :, x1] * wx1
    bot = img[:, y1][:, :, x0] * wx0 + img[:, y1][:, :, x1] * wx1
    return (top * wy0[None, :, None] + bot * wy1[None, :, None]).astype(F32)


def disc_kernel(radius: int = 3) -> np.ndarray:
    """h3dgsv3.py:209-220: ones inside sqrt(x^2 + y^2) <= radius + 0.5, normalised to sum 1."""
    y, x = np.meshgrid(np.arange(-radius, radius + 1), np.arange(-radius, radius + 1), indexing="ij")
    k = (np.sqrt((x ** 2 + y ** 2).astype(F32)) <= radius + 0.5).astype(F32)
    return (k / k.sum(dtype=F32)).astype(F32)


def lapla_norm(img: np.ndarray, disc: np.ndarray) -> np.ndarray:
    """utils.py:93-108 on a [C,h,w] image -> [h,w] in [0,1]."""
    C, h, w = img.shape
    p = np.zeros((C, h + 2, w + 2), F32)
    p[:, 1:-1, 1:-1] = img
    lap = (p[:, :-2, 1:-1] + p[:, 2:, 1:-1] + p[:, 1:-1, :-2] + p[:, 1:-1, 2:] - F32(4.0) * p[:, 1:-1, 1:-1]).sum(axis=0, dtype=F32)
    n = np.abs(lap).astype(F32)
    n[:, 0] = 0; n[:, -1] = 0; n[0, :] = 0; n[-1, :] = 0
    r = disc.shape[0] // 2
    q = np.zeros((h + 2 * r, w + 2 * r), F32)
    q[r:r + h, r:r + w] = n
    out = np.zeros((h, w), F32)
    for dy in range(2 * r + 1):
        for dx in range(2 * r + 1):
            if disc[dy, dx] != 0:
                out += disc[dy, dx] * q[dy:dy + h, dx:dx + w]
    return np.clip(out, 0, 1).astype(F32)


def grid_sample_at(m: np.ndarray, u: np.ndarray, v: np.ndarray, width: int, height: int) -> np.ndarray:
    """utils.sample(m[None, None], uv, width, height)[0, 0, 0] for a [Hs,Ws] map: normalise (u, v) by (width, height) - 1,
    un-normalise by the map's own size (align_corners=True), bilinear with zero padding."""
    Hs, Ws = m.shape
    gx = (u.astype(F32) * F32(2.0 / (width - 1)) - F32(1.0)).astype(F32)
    gy = (v.astype(F32) * F32(2.0 / (height - 1)) - F32(1.0)).astype(F32)
    ix = ((gx + F32(1.0)) * F32(0.5) * F32(Ws - 1)).astype(F32)
    iy = ((gy + F32(1.0)) * F32(0.5) * F32(Hs - 1)).astype(F32)
    x0 = np.floor(ix); y0 = np.floor(iy)
    wx1 = (ix - x0).astype(F32); wy1 = (iy - y0).astype(F32)
    wx0 = (F32(1.0) - wx1).astype(F32); wy0 = (F32(1.0) - wy1).astype(F32)
    x0 = x0.astype(np.int64); y0 = y0.astype(np.int64)

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < Ws) & (yy >= 0) & (yy < Hs)
        return np.where(ok, m[np.clip(yy, 0, Hs - 1), np.clip(xx, 0, Ws - 1)], F32(0.0)).astype(F32)

    return (tap(y0, x0) * (wx0 * wy0) + tap(y0, x0 + 1) * (wx1 * wy0) + tap(y0 + 1, x0) * (wx0 * wy1)
            + tap(y0 + 1, x0 + 1) * (wx1 * wy1)).astype(F32)


def quantile_linear(x: np.ndarray, q: float) -> np.float32:
    """torch.quantile(x, q) (interpolation="linear"): rank = q (n - 1) in fp32, lerp between the two order statistics."""
    s = np.sort(x.reshape(-1).astype(F32))
    rank = F32(q) * F32(s.size - 1)
    lo = int(np.floor(rank))
    hi = min(lo + 1, s.size - 1)
    wgt = F32(rank - F32(lo))
    a, b = s[lo], s[hi]
    return F32(a + wgt * (b - a)) if wgt < 0.5 else F32(b - (b - a) * (F32(1.0) - wgt))


def densify_level(img_lod, init_proba, penalty, rand, depth_map, conf_map, lod, *, W, H, f, R, t, approx_centre, qmin, gs_add_ratio=1.0):
    """One LoD level after the two probability maps exist (h3dgsv3.py:798-891).  img_lod [3,h,w]; init_proba / penalty [h,w]
    ALREADY multiplied by init_proba_scaler; rand [h,w] the uniform draw; depth_map / conf_map [Hs,Ws]; R [3,3], t [3] the
    keyframe's world-to-camera pose; qmin = min(1e-2, quantile).  Returns the final sample mask and the per-point tensors."""
    h, w = init_proba.shape
    mask = rand < ((init_proba - penalty) * F32(gs_add_ratio)).astype(F32)
    vv, uu = np.nonzero(mask)
    u, v = uu.astype(F32), vv.astype(F32)
    depth = grid_sample_at(depth_map, u, v, W // lod, H // lod)
    conf = grid_sample_at(conf_map, u, v, W // lod, H // lod)
    valid = (conf >= 0) & (depth > F32(qmin))
    final = np.zeros_like(mask)
    final[vv[valid], uu[valid]] = True
    u, v, depth, conf = u[valid], v[valid], depth[valid], conf[valid]
    fl, cx, cy = F32(f) / F32(lod), F32((W - 1) / 2) / F32(lod), F32((H - 1) / 2) / F32(lod)
    cam = np.stack([(u - cx) / fl, (v - cy) / fl, np.ones_like(u)], -1).astype(F32) * depth[:, None]
    pts = ((cam - t.astype(F32)[None]) @ R.astype(F32)).astype(F32)
    f_dc = ((img_lod[:, final].T - F32(0.5)) / C0).astype(F32)[:, None, :]
    p = init_proba[final]
    s = np.clip(F32(1.0) / np.sqrt(p), 1, F32(W / 10)).astype(F32) * F32(1.0 / f)
    s = s * np.linalg.norm(pts - approx_centre.astype(F32)[None], axis=-1).astype(F32)
    scales = np.log(F32(lod) * np.clip(s, 1e-6, 1e6)).astype(F32)[:, None].repeat(3, 1)
    o = (F32(0.2) * conf).astype(F32)
    opac = np.log(o / (F32(1.0) - o)).astype(F32)[:, None]
    return {"mask": final, "xyz": pts, "f_dc": f_dc, "scaling": scales, "opacity": opac, "d_max": (depth * F32(lod))[:, None].astype(F32),
            "depth": depth, "conf": conf}
