"""CPU oracle for `gsplat.rendering.rasterization` -- TEST INFRASTRUCTURE, never on the product path.

gsplat is a third-party pip dependency of ARTDECO (README.md:82-83 `pip install gsplat`,
unpinned; >= 1.5 is implied by `meta['radii'][0].max(dim=1)` at
Reconstruct/scene/scene_models/h3dgsv3.py:689) and its source is NOT under /root/reference.
This file restates the published algorithm of gsplat >= 1.5 (fully_fused_projection,
spherical_harmonics, isect_tiles / isect_offset_encode, rasterize_to_pixels -- SURVEY.md App. A)
for exactly the configuration ARTDECO calls it with (h3dgsv3.py:664-680): packed=False,
rasterize_mode="classic", render_mode="RGB+D" (also "RGB"/"D"), sh_degree<=3, pinhole, tile 16.

Arithmetic policy (what makes integer outputs bit-exact between this oracle and the HIP path):
every fp32 quantity that feeds an integer decision (depth sort key, radii, tile range) is computed
as a chain of single IEEE-754 fp32 operations in the order written here, no FMA contraction
(artdeco_amd/csrc/raster_project.hip is compiled with -ffp-contract=off and mirrors the order);
the one transcendental on that path, log(opacity*255), is evaluated in fp64 and rounded to fp32.

Parity: UNPINNED by the reference (it has no rasterizer test, golden vector or fixture).
Gradients are obtained by autograd (fp64) over this restatement, i.e. independently of the
hand-derived backward in the HIP kernels.
"""
from __future__ import annotations

import math

import numpy as np
import torch

TILE = 16
ALPHA_THRESHOLD = 1.0 / 255.0
MAX_ALPHA = 0.999
TRANSMITTANCE_EPS = 1e-4

SH_C0 = 0.2820947917738781
SH_C1 = 0.48860251190292


def _sqrt(t):
    """Correctly rounded sqrt.  torch's CPU sqrt goes through MKL VML (<= 1 ulp, NOT correctly
    rounded, and CPU-model dependent), which breaks bit-exactness of the fp32 integer path; numpy's
    sqrt is the IEEE instruction.  The fp64 / autograd path keeps torch.sqrt."""
    if t.dtype == torch.float32 and not t.requires_grad:
        return torch.from_numpy(np.sqrt(t.detach().numpy()))
    return torch.sqrt(t)


# ----------------------------------------------------------------------------- projection
def _quat_to_rotmat(q):
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    inv_norm = 1.0 / _sqrt(((x * x + y * y) + z * z) + w * w)
    x, y, z, w = x * inv_norm, y * inv_norm, z * inv_norm, w * inv_norm
    x2, y2, z2 = x * x, y * y, z * z
    xy, xz, yz = x * y, x * z, y * z
    wx, wy, wz = w * x, w * y, w * z
    # row-major R[i][j]
    return [[1.0 - 2.0 * (y2 + z2), 2.0 * (xy - wz), 2.0 * (xz + wy)],
            [2.0 * (xy + wz), 1.0 - 2.0 * (x2 + z2), 2.0 * (yz - wx)],
            [2.0 * (xz - wy), 2.0 * (yz + wx), 1.0 - 2.0 * (x2 + y2)]]


def _dot3(a0, b0, a1, b1, a2, b2):
    return (a0 * b0 + a1 * b1) + a2 * b2


def project(means, quats, scales, opacities, viewmat, K, width, height, eps2d=0.3,
            near_plane=0.01, far_plane=1e10, radius_clip=0.0, force_valid=None):
    """fully_fused_projection (non-packed, one camera).  All inputs one dtype (fp32 for the
    bit-exact integer outputs, fp64 + requires_grad for the gradient reference).

    Returns dict: radii int32 [N,2], means2d [N,2], depths [N], conics [N,3], valid bool [N].
    Culled Gaussians have radii 0 and zeros elsewhere.  force_valid (bool [N]): the cull decisions of ANOTHER pass (the fp32 integer
    pass) decide which rows are zeroed in the outputs -- a high-precision pass must not drop a Gaussian that sits on the fp32 lists.
    """
    dt = means.dtype
    R = [[viewmat[i, j] for j in range(3)] for i in range(3)]
    t = [viewmat[i, 3] for i in range(3)]
    x, y, z = means[:, 0], means[:, 1], means[:, 2]
    mc = [(_dot3(R[i][0], x, R[i][1], y, R[i][2], z)) + t[i] for i in range(3)]
    valid = (mc[2] >= near_plane) & (mc[2] <= far_plane)

    Rq = _quat_to_rotmat(quats)
    s = [scales[:, 0], scales[:, 1], scales[:, 2]]
    M = [[Rq[i][j] * s[j] for j in range(3)] for i in range(3)]
    cov = [[None] * 3 for _ in range(3)]
    for i in range(3):
        for j in range(i, 3):
            cov[i][j] = _dot3(M[i][0], M[j][0], M[i][1], M[j][1], M[i][2], M[j][2])
            cov[j][i] = cov[i][j]
    # covar_c = R * cov * R^T (upper triangle only; symmetric by construction)
    A = [[_dot3(R[i][0], cov[0][j], R[i][1], cov[1][j], R[i][2], cov[2][j]) for j in range(3)] for i in range(3)]
    C = [[None] * 3 for _ in range(3)]
    for i in range(3):
        for j in range(i, 3):
            C[i][j] = _dot3(A[i][0], R[j][0], A[i][1], R[j][1], A[i][2], R[j][2])
            C[j][i] = C[i][j]

    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    tan_fovx = 0.5 * width / fx
    tan_fovy = 0.5 * height / fy
    lim_x_pos = (width - cx) / fx + 0.3 * tan_fovx
    lim_x_neg = cx / fx + 0.3 * tan_fovx
    lim_y_pos = (height - cy) / fy + 0.3 * tan_fovy
    lim_y_neg = cy / fy + 0.3 * tan_fovy
    # guard the division for culled points so no NaN/Inf leaks into autograd
    zs = torch.where(valid, mc[2], torch.ones_like(mc[2]))
    rz = 1.0 / zs
    rz2 = rz * rz
    tx = zs * torch.minimum(lim_x_pos, torch.maximum(-lim_x_neg, mc[0] * rz))
    ty = zs * torch.minimum(lim_y_pos, torch.maximum(-lim_y_neg, mc[1] * rz))
    j00 = fx * rz
    j02 = -fx * tx * rz2
    j11 = fy * rz
    j12 = -fy * ty * rz2
    v00 = j00 * C[0][0] + j02 * C[0][2]
    v01 = j00 * C[0][1] + j02 * C[1][2]
    v02 = j00 * C[0][2] + j02 * C[2][2]
    v11 = j11 * C[1][1] + j12 * C[1][2]
    v12 = j11 * C[1][2] + j12 * C[2][2]
    c00 = v00 * j00 + v02 * j02
    c01 = v01 * j11 + v02 * j12
    c11 = v11 * j11 + v12 * j12
    m2x = (fx * mc[0]) * rz + cx
    m2y = (fy * mc[1]) * rz + cy

    c00 = c00 + eps2d
    c11 = c11 + eps2d
    det = c00 * c11 - c01 * c01
    valid = valid & (det > 0)
    dets = torch.where(valid, det, torch.ones_like(det))
    inv_det = 1.0 / dets
    conic_a = c11 * inv_det
    conic_b = -c01 * inv_det
    conic_c = c00 * inv_det

    # opacity-aware extent (gsplat >= 1.5; arXiv 2402.00525 B.2), non-differentiable from here on
    with torch.no_grad():
        thr = torch.tensor(ALPHA_THRESHOLD, dtype=dt)
        op = opacities.detach()
        valid = valid & ~(op < thr)
        ratio = torch.where(valid, op / thr, torch.ones_like(op))
        lg = torch.log(ratio.double()).to(dt)
        extend = torch.minimum(torch.tensor(3.33, dtype=dt), _sqrt(2.0 * lg))
        d00, d11, dd = c00.detach(), c11.detach(), dets.detach()
        b = 0.5 * (d00 + d11)
        tmp = _sqrt(torch.clamp_min(b * b - dd, 0.01))
        v1 = b + tmp
        r1 = extend * _sqrt(v1)
        rad_x = torch.ceil(torch.minimum(extend * _sqrt(torch.clamp_min(d00, 0)), r1))
        rad_y = torch.ceil(torch.minimum(extend * _sqrt(torch.clamp_min(d11, 0)), r1))
        valid = valid & ~((rad_x <= radius_clip) & (rad_y <= radius_clip))
        mx, my = m2x.detach(), m2y.detach()
        outside = (mx + rad_x <= 0) | (mx - rad_x >= width) | (my + rad_y <= 0) | (my - rad_y >= height)
        valid = valid & ~outside
        radii = torch.stack([rad_x, rad_y], -1)
        radii = torch.where(valid[:, None], radii, torch.zeros_like(radii)).to(torch.int32)

    if force_valid is not None:
        valid = force_valid
    zero = torch.zeros_like(m2x)
    means2d = torch.stack([torch.where(valid, m2x, zero), torch.where(valid, m2y, zero)], -1)
    depths = torch.where(valid, mc[2], zero)
    conics = torch.stack([torch.where(valid, conic_a, zero), torch.where(valid, conic_b, zero),
                          torch.where(valid, conic_c, zero)], -1)
    return {"radii": radii, "means2d": means2d, "depths": depths, "conics": conics, "valid": valid}


def camera_position(viewmat):
    """inverse(viewmat)[:3, 3] for an affine [R t; 0 1]: -R^-1 t via the adjugate of R."""
    R = viewmat[:3, :3]
    t = viewmat[:3, 3]
    c00 = R[1, 1] * R[2, 2] - R[1, 2] * R[2, 1]
    c01 = R[0, 2] * R[2, 1] - R[0, 1] * R[2, 2]
    c02 = R[0, 1] * R[1, 2] - R[0, 2] * R[1, 1]
    c10 = R[1, 2] * R[2, 0] - R[1, 0] * R[2, 2]
    c11 = R[0, 0] * R[2, 2] - R[0, 2] * R[2, 0]
    c12 = R[0, 2] * R[1, 0] - R[0, 0] * R[1, 2]
    c20 = R[1, 0] * R[2, 1] - R[1, 1] * R[2, 0]
    c21 = R[0, 1] * R[2, 0] - R[0, 0] * R[2, 1]
    c22 = R[0, 0] * R[1, 1] - R[0, 1] * R[1, 0]
    det = (R[0, 0] * c00 + R[0, 1] * c10) + R[0, 2] * c20
    inv_det = 1.0 / det
    Ri = torch.stack([torch.stack([c00, c01, c02]), torch.stack([c10, c11, c12]), torch.stack([c20, c21, c22])]) * inv_det
    return -(Ri @ t)


# ----------------------------------------------------------------------------- spherical harmonics
def sh_to_rgb(degree, dirs, coeffs):
    """spherical_harmonics() + the `clamp_min(colors + 0.5, 0)` of rasterization(): [N,3].
    dirs [N,3] (not normalised), coeffs [N,K,3] with K >= (degree+1)^2."""
    inorm = 1.0 / _sqrt((dirs[:, 0] * dirs[:, 0] + dirs[:, 1] * dirs[:, 1]) + dirs[:, 2] * dirs[:, 2])
    x, y, z = (dirs[:, 0] * inorm)[:, None], (dirs[:, 1] * inorm)[:, None], (dirs[:, 2] * inorm)[:, None]
    c = coeffs
    res = SH_C0 * c[:, 0]
    if degree >= 1:
        res = res + SH_C1 * (-y * c[:, 1] + z * c[:, 2] - x * c[:, 3])
    if degree >= 2:
        z2 = z * z
        fTmp0B = -1.092548430592079 * z
        fC1 = x * x - y * y
        fS1 = 2.0 * x * y
        pSH6 = 0.9461746957575601 * z2 - 0.3153915652525201
        pSH7 = fTmp0B * x
        pSH5 = fTmp0B * y
        pSH8 = 0.5462742152960395 * fC1
        pSH4 = 0.5462742152960395 * fS1
        res = res + pSH4 * c[:, 4] + pSH5 * c[:, 5] + pSH6 * c[:, 6] + pSH7 * c[:, 7] + pSH8 * c[:, 8]
    if degree >= 3:
        fTmp0C = -2.285228997322329 * z2 + 0.4570457994644658
        fTmp1B = 1.445305721320277 * z
        fC2 = x * fC1 - y * fS1
        fS2 = x * fS1 + y * fC1
        pSH12 = z * (1.865881662950577 * z2 - 1.119528997770346)
        pSH13 = fTmp0C * x
        pSH11 = fTmp0C * y
        pSH14 = fTmp1B * fC1
        pSH10 = fTmp1B * fS1
        pSH15 = -0.5900435899266435 * fC2
        pSH9 = -0.5900435899266435 * fS2
        res = (res + pSH9 * c[:, 9] + pSH10 * c[:, 10] + pSH11 * c[:, 11] + pSH12 * c[:, 12]
               + pSH13 * c[:, 13] + pSH14 * c[:, 14] + pSH15 * c[:, 15])
    return torch.clamp_min(res + 0.5, 0.0)


# ----------------------------------------------------------------------------- tile binning (integers)
def isect_tiles(means2d, radii, depths, width, height, tile=TILE):
    """isect_tiles + stable sort + isect_offset_encode for one camera, in numpy integers.

    Returns dict: tiles_per_gauss int32 [N], isect_ids int64 [I] (sorted), flatten_ids int32 [I]
    (sorted), offsets int32 [tile_h, tile_w], tile_w, tile_h, unsorted copies (isect_ids_unsorted ...)."""
    m = means2d.detach().to(torch.float32).numpy()
    r = radii.numpy().astype(np.float32)
    d = depths.detach().to(torch.float32).numpy()
    tile_w = (width + tile - 1) // tile
    tile_h = (height + tile - 1) // tile
    n_tiles = tile_w * tile_h
    tile_n_bits = int(math.floor(math.log2(n_tiles))) + 1 if n_tiles > 0 else 1
    ts = np.float32(tile)
    trx, try_ = r[:, 0] / ts, r[:, 1] / ts
    tx, ty = m[:, 0] / ts, m[:, 1] / ts
    vis = (radii[:, 0] > 0).numpy() & (radii[:, 1] > 0).numpy()

    def lo(v, n):  # min(max(0, (uint32)floor(v)), n)
        return np.clip(np.floor(v), 0, n).astype(np.int64)

    def hi(v, n):
        return np.clip(np.ceil(v), 0, n).astype(np.int64)

    x0, x1 = lo(tx - trx, tile_w), hi(tx + trx, tile_w)
    y0, y1 = lo(ty - try_, tile_h), hi(ty + try_, tile_h)
    cnt = np.where(vis, (y1 - y0) * (x1 - x0), 0).astype(np.int64)
    cum = np.cumsum(cnt)
    I = int(cum[-1]) if len(cum) else 0
    ids = np.zeros(I, dtype=np.int64)
    flat = np.zeros(I, dtype=np.int32)
    depth_bits = d.view(np.int32).astype(np.int64) & 0xFFFFFFFF
    start = cum - cnt
    g_idx = np.nonzero(cnt)[0]
    # vectorised emit: for each gaussian, its covered tiles in row-major (y outer, x inner) order
    if I > 0:
        rep = np.repeat(g_idx, cnt[g_idx])
        local = np.arange(I, dtype=np.int64) - np.repeat(start[g_idx], cnt[g_idx])
        wx = (x1 - x0)[rep]
        ty_i = y0[rep] + local // wx
        tx_i = x0[rep] + local % wx
        tile_id = ty_i * tile_w + tx_i
        ids = (tile_id << 32) | depth_bits[rep]
        flat = rep.astype(np.int32)
    order = np.argsort(ids, kind="stable")
    ids_s, flat_s = ids[order], flat[order]
    tile_of = (ids_s >> 32).astype(np.int64)
    offsets = np.searchsorted(tile_of, np.arange(n_tiles, dtype=np.int64), side="left").astype(np.int32)
    return {"tiles_per_gauss": cnt.astype(np.int32), "isect_ids": ids_s, "flatten_ids": flat_s,
            "offsets": offsets.reshape(tile_h, tile_w), "tile_w": tile_w, "tile_h": tile_h,
            "tile_n_bits": tile_n_bits, "isect_ids_unsorted": ids, "flatten_ids_unsorted": flat, "n_isects": I}


# ----------------------------------------------------------------------------- compositing
KNIFE_EPS = 5e-4


class _one_thread:
    """The per-tile tensors ([n,256]) are far too small for OpenMP: with 8+ threads torch spends its time in fork/join
    (measured here: 0.8 s single-threaded vs 5-30 s with 4-8 threads for one 256x192 frame)."""

    def __enter__(self):
        self.n = torch.get_num_threads()
        torch.set_num_threads(1)

    def __exit__(self, *a):
        torch.set_num_threads(self.n)


def _tile_decisions(m2, cn, op, px, py):
    """(keep, live, ov, T_incl, T_excl, sigma) of one tile in the dtype of the inputs, no autograd: which (splat, pixel) pairs pass
    `sigma >= 0 and alpha >= 1/255`, and which of those are blended before the pixel terminates."""
    dt = m2.dtype
    thr_alpha = torch.tensor(ALPHA_THRESHOLD, dtype=torch.float32).to(dt)
    P = px.shape[0]
    dx = m2[:, 0][:, None] - px[None, :]
    dy = m2[:, 1][:, None] - py[None, :]
    sigma = 0.5 * (cn[:, 0:1] * dx * dx + cn[:, 2:3] * dy * dy) + cn[:, 1:2] * dx * dy
    ov = op[:, None] * torch.exp(-sigma)
    alpha = torch.clamp_max(ov, MAX_ALPHA)
    keep = ~((sigma < 0) | (alpha < thr_alpha))
    one_m = 1.0 - torch.where(keep, alpha, torch.zeros_like(alpha))
    T_incl = torch.cumprod(one_m, dim=0)
    T_excl = torch.cat([torch.ones(1, P, dtype=dt), T_incl[:-1]], 0)
    live = keep & (T_incl > TRANSMITTANCE_EPS)
    return keep, live, ov, T_incl, T_excl, sigma


def knife_pairs(dec, keep, eps, eps_T=None):
    """bool [n, P]: the (splat, pixel) pairs of `_tile_decisions` output `dec` that sit on one of the rasteriser's DISCONTINUITIES while the
    pixel is still live -- alpha within a relative `eps` of 1/255 (skip) or of 0.999 (clamp), sigma at 0, or the transmittance after a kept
    splat within a relative `eps_T` (default eps) of 1e-4 (terminate)."""
    ov, Ti, Te, sg = dec[2], dec[3], dec[4], dec[5]
    eps_T = eps if eps_T is None else eps_T
    reached = Te > TRANSMITTANCE_EPS * (1.0 - eps_T)
    near = ((ov * 255.0 - 1.0).abs() <= eps) | ((ov - MAX_ALPHA).abs() <= eps * MAX_ALPHA) \
        | (sg.abs() <= 1e-6) | (keep & ((Ti - TRANSMITTANCE_EPS).abs() <= eps_T * TRANSMITTANCE_EPS))
    return near & reached


def composite_tile(m2, cn, col, op, px, py, first_index=0, want_extras=False, knife_eps=None, decide=None, knife_eps_T=None):
    """ONE tile of rasterize_to_pixels_fwd (App. A item 4), vectorised over [n splats in list order, P pixels], autograd-capable.

    m2 [n,2], cn [n,3], col [n,CDIM], op [n] of the tile's list (depth order); px, py [P] pixel centres; first_index = position of the
    list's first entry in the sorted intersection list.  Returns (colour [P,CDIM], T_final [P], last index int32 [P], extras or None);
    extras = (knife bool [P], main list position int64 [P] (-1: none), main_w [P], second_w [P], touch bool [n,P], skip_knife bool [P]) --
    `touch` marks the (splat, pixel) pairs that either contribute (alpha T > 0) or sit on a SKIP / clamp decision within knife_eps while the
    pixel is still live: the pairs whose gradients move if that pixel's decisions fall differently; `skip_knife` = `knife` without the
    termination edge (a splat entering or leaving at T = 1e-4 moves nothing a gradient comparison sees).
    decide = (m2, cn, op) in float32: the per-pixel DECISIONS (skip below 1/255, terminate at T <= 1e-4) are taken by a single-precision
    evaluation of these -- the values an fp32 rasteriser decides on -- and the differentiable arithmetic of the call's own dtype runs ON
    them (the knife band of `extras` is then measured on the fp32 values too: where ANOTHER fp32 evaluation may still decide otherwise)."""
    dt = m2.dtype
    eps = KNIFE_EPS if knife_eps is None else knife_eps
    thr_alpha = torch.tensor(ALPHA_THRESHOLD, dtype=torch.float32).to(dt)
    n, P = m2.shape[0], px.shape[0]
    dx = m2[:, 0][:, None] - px[None, :]  # [n, P]
    dy = m2[:, 1][:, None] - py[None, :]
    sigma = 0.5 * (cn[:, 0:1] * dx * dx + cn[:, 2:3] * dy * dy) + cn[:, 1:2] * dx * dy
    alpha = torch.clamp_max(op[:, None] * torch.exp(-sigma), MAX_ALPHA)
    dec = None
    if decide is not None:
        with torch.no_grad():
            dec = _tile_decisions(decide[0], decide[1], decide[2], px.to(decide[0].dtype), py.to(decide[0].dtype))
        keep = dec[0]
    else:
        keep = ~((sigma < 0) | (alpha < thr_alpha))
    a = torch.where(keep, alpha, torch.zeros_like(alpha))
    one_m = 1.0 - a
    T_incl = torch.cumprod(one_m, dim=0)  # T after splat k
    T_excl = torch.cat([torch.ones(1, P, dtype=dt), T_incl[:-1]], 0)
    # a kept splat whose next_T <= 1e-4 terminates the pixel BEFORE being added; since T is
    # monotone, every later kept splat also fails the test, so the mask is simply:
    live = dec[1] if dec is not None else keep & (T_incl.detach() > TRANSMITTANCE_EPS)
    w = torch.where(live, a * T_excl, torch.zeros_like(a))  # alpha * T
    colour = (w[:, :, None] * col[:, None, :]).sum(0)  # [P, cdim]
    T_final = torch.prod(torch.where(live, one_m, torch.ones_like(one_m)), dim=0)
    idx = torch.arange(first_index, first_index + n, dtype=torch.int32)[:, None].expand(-1, P)
    lastk = torch.where(live, idx, torch.zeros_like(idx)).max(0).values
    extras = None
    if want_extras:
        with torch.no_grad():
            if dec is not None:
                ov, Ti, Te, sg = dec[2], dec[3], dec[4], dec[5]
            else:
                ov, Ti, Te, sg = (op[:, None] * torch.exp(-sigma)).detach(), T_incl.detach(), T_excl.detach(), sigma.detach()
            on_edge = knife_pairs((None, None, ov, Ti, Te, sg), keep, eps, knife_eps_T)
            on_skip_edge = knife_pairs((None, None, ov, Ti, Te, sg), keep, eps, 0.0)     # without the termination edge
            knife = on_edge.any(0)
            wd = w.detach()
            k = min(2, n)
            top = torch.topk(wd, k, dim=0)
            w1 = top.values[0]
            w2 = top.values[1] if k > 1 else torch.zeros_like(w1)
            first = (wd == w1[None, :]).to(torch.int8).argmax(0)  # first list position holding the maximum
            main = torch.where(w1 > 0, first, torch.full_like(first, -1))
            extras = (knife, main, w1, w2, (wd > 0) | on_skip_edge, on_skip_edge.any(0))
    return colour, T_final, lastk, extras


def rasterize_to_pixels(means2d, conics, colors, opacities, width, height, isects, backgrounds=None, tile_window=None,
                        extras=None):
    """Per-tile vectorised, autograd-capable restatement of rasterize_to_pixels_fwd (App. A item 4).

    colors [N,CDIM].  Returns render_colors [H,W,CDIM], render_alphas [H,W,1], last_ids int32 [H,W]
    (index into the sorted intersection list of the last splat that contributed; 0 if none).

    tile_window = (tx0, ty0, tx1, ty1): only tiles with tx0 <= tx < tx1, ty0 <= ty < ty1 are composited (the rest of the
    image stays empty) -- keeps the fp64 autograd reference tractable at the BASELINE sizes.
    extras: a dict that receives
      "knife"   bool [H,W]: the pixel has a reached splat sitting on one of the rasteriser's DISCONTINUITIES within a
                relative KNIFE_EPS -- alpha vs 1/255 (skip), alpha vs 0.999 (clamp: gradient switches off), sigma vs 0,
                or T(1-alpha) vs 1e-4 (terminate).  fp32 and fp64 evaluations may legitimately decide such a pixel
                differently, so tolerance tests exclude exactly these pixels instead of allowing a blanket outlier fraction.
      "main_ids" int32 [H,W]: Gaussian with the largest alpha*T (first in list order on exact ties), -1 if none;
      "main_w", "second_w" [H,W]: the two largest alpha*T values of the pixel.
    """
    dt = means2d.dtype
    cdim = colors.shape[1]
    tile_w, tile_h = isects["tile_w"], isects["tile_h"]
    offsets = isects["offsets"].reshape(-1)
    flat = torch.from_numpy(isects["flatten_ids"].astype(np.int64))
    I = isects["n_isects"]
    Hp, Wp = tile_h * TILE, tile_w * TILE
    out_c = torch.zeros(Hp, Wp, cdim, dtype=dt)
    out_T = torch.ones(Hp, Wp, dtype=dt)
    last = torch.zeros(Hp, Wp, dtype=torch.int32)
    ys, xs = torch.meshgrid(torch.arange(TILE), torch.arange(TILE), indexing="ij")
    if extras is not None:
        ex_knife = torch.zeros(Hp, Wp, dtype=torch.bool)
        ex_main = torch.full((Hp, Wp), -1, dtype=torch.int32)
        ex_w1 = torch.zeros(Hp, Wp, dtype=dt)
        ex_w2 = torch.zeros(Hp, Wp, dtype=dt)
    threads = _one_thread()
    threads.__enter__()
    for tid in range(tile_w * tile_h):
        s = int(offsets[tid])
        e = int(offsets[tid + 1]) if tid + 1 < tile_w * tile_h else I
        if e <= s:
            continue
        ty, tx = divmod(tid, tile_w)
        if tile_window is not None and not (tile_window[0] <= tx < tile_window[2] and tile_window[1] <= ty < tile_window[3]):
            continue
        g = flat[s:e]
        px = (tx * TILE + xs).reshape(-1).to(dt) + 0.5  # [256]
        py = (ty * TILE + ys).reshape(-1).to(dt) + 0.5
        col, T_final, lastk, ex = composite_tile(means2d[g], conics[g], colors[g], opacities[g], px, py, first_index=s,
                                                 want_extras=extras is not None)
        sl = (slice(ty * TILE, (ty + 1) * TILE), slice(tx * TILE, (tx + 1) * TILE))
        out_c[sl] = col.reshape(TILE, TILE, cdim)
        out_T[sl] = T_final.reshape(TILE, TILE)
        last[sl] = lastk.reshape(TILE, TILE)
        if extras is not None:
            knife, main, w1, w2 = ex[:4]
            ex_knife[sl] = knife.reshape(TILE, TILE)
            mid = torch.where(main >= 0, g[main.clamp_min(0)].to(torch.int32), torch.full_like(main, -1, dtype=torch.int32))
            ex_main[sl] = mid.reshape(TILE, TILE)
            ex_w1[sl] = w1.reshape(TILE, TILE)
            ex_w2[sl] = w2.reshape(TILE, TILE)
    threads.__exit__()
    out_c, out_T, last = out_c[:height, :width], out_T[:height, :width], last[:height, :width]
    if extras is not None:
        extras.update(knife=ex_knife[:height, :width], main_ids=ex_main[:height, :width], main_w=ex_w1[:height, :width],
                      second_w=ex_w2[:height, :width])
    if backgrounds is not None:
        out_c = out_c + out_T[..., None] * backgrounds.to(dt)[None, None, :]
    return out_c, (1.0 - out_T)[..., None], last


def rasterize_to_pixels_loop(means2d, conics, colors, opacities, width, height, isects, backgrounds=None):
    """Literal per-pixel sequential loop (pure Python; tiny inputs only) used to validate the
    vectorised version above against the kernel's control flow: skip / terminate-before-add / last id."""
    m2, cn, col, op = (t.detach().double().numpy() for t in (means2d, conics, colors, opacities))
    cdim = col.shape[1]
    tile_w = isects["tile_w"]
    offsets = isects["offsets"].reshape(-1)
    flat, I = isects["flatten_ids"], isects["n_isects"]
    out = np.zeros((height, width, cdim))
    alpha_out = np.zeros((height, width, 1))
    last = np.zeros((height, width), dtype=np.int32)
    for i in range(height):
        for j in range(width):
            tid = (i // TILE) * tile_w + (j // TILE)
            s = int(offsets[tid])
            e = int(offsets[tid + 1]) if tid + 1 < len(offsets) else I
            px, py = j + 0.5, i + 0.5
            T, cur = 1.0, 0
            acc = np.zeros(cdim)
            for k in range(s, e):
                g = flat[k]
                dx, dy = m2[g, 0] - px, m2[g, 1] - py
                sigma = 0.5 * (cn[g, 0] * dx * dx + cn[g, 2] * dy * dy) + cn[g, 1] * dx * dy
                alpha = min(MAX_ALPHA, op[g] * math.exp(-sigma))
                if sigma < 0 or alpha < np.float32(ALPHA_THRESHOLD):
                    continue
                next_T = T * (1 - alpha)
                if next_T <= TRANSMITTANCE_EPS:
                    break
                acc += col[g] * alpha * T
                cur = k
                T = next_T
            out[i, j] = acc + (T * backgrounds.numpy() if backgrounds is not None else 0)
            alpha_out[i, j, 0] = 1 - T
            last[i, j] = cur
    return out, alpha_out, last


# ----------------------------------------------------------------------------- full pipeline
def rasterization(means, quats, scales, opacities, colors, viewmat, K, width, height,
                  sh_degree=3, eps2d=0.3, render_mode="RGB+D", near_plane=0.01, far_plane=1e10,
                  radius_clip=0.0, backgrounds=None, grad_dtype=None, tile_window=None, extras=None):
    """One-camera restatement of gsplat.rendering.rasterization as ARTDECO calls it.

    Integer decisions always come from an fp32 pass.  With grad_dtype=torch.float64 the
    differentiable quantities are recomputed in fp64 (inputs may require grad) on top of the
    fp32 binning, so that autograd yields a high-precision gradient reference for the SAME
    tile lists the kernel uses.

    Returns (render [H,W,C], alphas [H,W,1], meta dict).
    """
    f32 = torch.float32
    det = lambda t: t.detach().to(f32)
    p32 = project(det(means), det(quats), det(scales), det(opacities), det(viewmat), det(K), width, height,
                  eps2d, near_plane, far_plane, radius_clip)
    isects = isect_tiles(p32["means2d"], p32["radii"], p32["depths"], width, height)
    dt = grad_dtype or f32
    if grad_dtype is None:
        p = p32
        mm, vm, op, cc = det(means), det(viewmat), det(opacities), det(colors)
    else:
        cast = lambda t: t.to(dt)
        mm, vm, op, cc = cast(means), cast(viewmat), cast(opacities), cast(colors)
        p = project(mm, cast(quats), cast(scales), op, vm, cast(K), width, height, eps2d, near_plane, far_plane, radius_clip)
    vis = p32["valid"]
    if sh_degree is not None:
        dirs = mm - camera_position(vm)[None, :]
        # masked-out Gaussians never reach the rasteriser; give them a safe direction
        dirs = torch.where(vis[:, None], dirs, torch.ones_like(dirs))
        rgb = sh_to_rgb(sh_degree, dirs, cc)
    else:
        rgb = cc
    if render_mode == "RGB+D":
        feat = torch.cat([rgb, p["depths"][:, None]], -1)
    elif render_mode == "D":
        feat = p["depths"][:, None]
    elif render_mode == "RGB":
        feat = rgb
    else:
        raise ValueError(render_mode)
    render, alphas, last = rasterize_to_pixels(p["means2d"], p["conics"], feat, op, width, height, isects, backgrounds,
                                               tile_window=tile_window, extras=extras)
    meta = {"radii": p32["radii"], "means2d": p["means2d"], "depths": p["depths"], "conics": p["conics"],
            "colors": feat, "isects": isects, "last_ids": last, "p32": p32}
    return render, alphas, meta


def rasterization_window(sc, tile_window, eps2d=0.01, sh_degree=3, dtype=torch.float64):
    """The gradient reference at the BASELINE sizes (1 M / 1080p, 4 M / 2592x1944): fp32 projection and binning of
    the WHOLE scene (vectorised, seconds), then high-precision autograd restricted to the tiles of `tile_window`
    and to the Gaussians on those tiles' lists (every other Gaussian has an exactly zero gradient for a loss supported
    on the window).  sc: dict from synthetic_scene() (fp32; "viewmat" may be replaced).

    Returns dict: ids (int64 [M], Gaussians in the window, ascending), leaves (means/quats/scales/opacities/colors of those
    + the full viewmat, `dtype`, requires_grad), render [H,W,4], alphas [H,W,1], extras (see rasterize_to_pixels), p32,
    isects (of the whole scene, fp32 integer pass).
    """
    W, H = sc["width"], sc["height"]
    p32 = project(sc["means"], sc["quats"], sc["scales"], sc["opacities"], sc["viewmat"], sc["K"], W, H, eps2d)
    isects = isect_tiles(p32["means2d"], p32["radii"], p32["depths"], W, H)
    tile_w, tile_h, I = isects["tile_w"], isects["tile_h"], isects["n_isects"]
    off = isects["offsets"].reshape(-1)
    tx0, ty0, tx1, ty1 = tile_window
    spans = []
    for ty in range(ty0, ty1):
        for tx in range(tx0, tx1):
            tid = ty * tile_w + tx
            spans.append((int(off[tid]), int(off[tid + 1]) if tid + 1 < tile_w * tile_h else I))
    ids = np.unique(np.concatenate([isects["flatten_ids"][s:e] for s, e in spans] or [np.zeros(0, np.int32)])).astype(np.int64)
    sub = dict(isects)
    sub["flatten_ids"] = np.searchsorted(ids, isects["flatten_ids"]).astype(np.int32)  # only window entries are ever read
    tid_ = torch.from_numpy(ids)
    leaves = {k: sc[k][tid_].to(dtype).clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")}
    leaves["viewmat"] = sc["viewmat"].to(dtype).clone().requires_grad_(True)
    p = project(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["viewmat"], sc["K"].to(dtype),
                W, H, eps2d)
    vis = p32["valid"][tid_]
    dirs = leaves["means"] - camera_position(leaves["viewmat"])[None, :]
    dirs = torch.where(vis[:, None], dirs, torch.ones_like(dirs))
    rgb = sh_to_rgb(sh_degree, dirs, leaves["colors"])
    feat = torch.cat([rgb, p["depths"][:, None]], -1)
    extras = {}
    render, alphas, last = rasterize_to_pixels(p["means2d"], p["conics"], feat, leaves["opacities"], W, H, sub,
                                               tile_window=tile_window, extras=extras)
    return {"ids": tid_, "leaves": leaves, "render": render, "alphas": alphas, "extras": extras, "p32": p32, "isects": isects,
            "last_ids": last}


# ----------------------------------------------------------------------------- synthetic scenes
def synthetic_scene(N, width, height, seed=0, sh_k=16, z_range=(2.0, 6.0), sigma_px=2.0, dtype=torch.float32):
    """Seeded frustum-uniform Gaussian cloud + identity camera exactly as SURVEY.md 8(d) specifies."""
    g = torch.Generator().manual_seed(seed)
    fx = fy = 0.8 * width
    cx, cy = width / 2.0, height / 2.0
    z = torch.rand(N, generator=g) * (z_range[1] - z_range[0]) + z_range[0]
    u = torch.rand(N, generator=g) * 2.1 - 1.05
    v = torch.rand(N, generator=g) * 2.1 - 1.05
    means = torch.stack([u * z * width / (2 * fx), v * z * height / (2 * fy), z], -1)
    s0 = sigma_px * 4.0 / fx
    scales = torch.exp(math.log(s0) + 0.3 * torch.randn(N, 3, generator=g))
    quats = torch.randn(N, 4, generator=g)
    quats = quats / quats.norm(dim=1, keepdim=True)
    opacities = torch.sigmoid(torch.randn(N, generator=g))
    sh = 0.3 * torch.randn(N, sh_k, 3, generator=g)
    viewmat = torch.eye(4)
    K = torch.tensor([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    out = dict(means=means, quats=quats, scales=scales, opacities=opacities, colors=sh, viewmat=viewmat, K=K,
               width=width, height=height)
    return {k: (v.to(dtype) if torch.is_tensor(v) else v) for k, v in out.items()}
