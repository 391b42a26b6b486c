"""Host side of the MI355X Gaussian rasteriser: one fused autograd node over the C-ABI stages.

Mirrors what gsplat.rendering.rasterization does for ARTDECO's call
(Reconstruct/scene/scene_models/h3dgsv3.py:664-680) but as ONE torch.autograd.Function per
camera instead of upstream's chain (projection -> SH -> isect -> rasterize): the intermediate
per-Gaussian tensors live in one packed 48 B record that both the tile kernels and the
backward consume, and the only host wait is for n_isects (it sizes the intersection list; upstream
drains the stream for it) -- here it is counted right after the projection and fetched through
pinned memory while the depth sort runs, so the stream stays busy.
"""
from __future__ import annotations

import contextlib
import os
import sys
import threading
from dataclasses import dataclass

import torch

from . import _lib

_COLOR_SH, _COLOR_RGB, _COLOR_DEPTH = 0, 1, 2


class StageTimer:
    """HIP-event timing of the native stages, recorded on the stream the kernels are launched on
    (torch's current stream).  bench.py installs one with set_stage_timer() for the timed region;
    when none is installed the stages run without any event overhead."""

    def __init__(self, only=None):
        self.events: dict[str, list] = {}
        self.only = None if only is None else frozenset(only)
        self._native: dict[str, list] = {}   # stage -> [sum, min, count] folded from the C side's pairs (adk_mapper_step)

    def _absorb(self, drained: dict) -> None:
        for k, (s, m, c) in drained.items():
            if self.only is not None and k not in self.only:
                continue
            a = self._native.setdefault(k, [0.0, m, 0])
            a[0] += s; a[1] = min(a[1], m); a[2] += c

    @contextlib.contextmanager
    def stage(self, name: str):
        if self.only is not None and name not in self.only:
            yield
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        try:
            yield
        finally:
            e1.record()
            self.events.setdefault(name, []).append((e0, e1))

    def summary_ms(self) -> dict[str, dict]:
        torch.cuda.synchronize()
        acc: dict[str, list] = {}   # stage -> [sum, min, count]
        for k, evs in self.events.items():
            ts = [a.elapsed_time(b) for a, b in evs]
            acc[k] = [sum(ts), min(ts), len(ts)]
        # stages that ran inside adk_mapper_step recorded their event pairs on the C side (artdeco_amd/native_step.py): the pairs still
        # pending there belong to the installed timer; a timer that has been removed absorbed its pairs when it was
        from . import native_step
        if _TIMER is self:
            self._absorb(native_step.drain_timings())
        for k, (s, m, c) in self._native.items():
            a = acc.setdefault(k, [0.0, m, 0])
            a[0] += s; a[1] = min(a[1], m); a[2] += c
        return {k: {"mean_ms": a[0] / a[2], "min_ms": a[1], "count": a[2]} for k, a in acc.items()}


_TIMER: StageTimer | None = None
_NULL = contextlib.nullcontext()


def set_stage_timer(t: StageTimer | None) -> None:
    global _TIMER
    if _TIMER is not None:
        # the pairs pending on the C side were recorded under the OUTGOING timer: it absorbs them (a later summary_ms() still reports them),
        # nothing stays behind for the next timer.  Without an outgoing timer nothing can be pending (pairs are only recorded under one),
        # and the call must stay free: no import, no header parse, no hipEventSynchronize.
        from . import native_step
        _TIMER._absorb(native_step.drain_timings())
    _TIMER = t


def _stage(name: str):
    return _TIMER.stage(name) if _TIMER is not None else _NULL


# Sizes of the most recent forward on this process (N Gaussians in, I tile intersections) --
# bench.py reads them to turn kernel times into algorithmic bytes/s.
LAST_STATS: dict[str, int] = {}


@dataclass(frozen=True)
class RasterConfig:
    width: int
    height: int
    sh_degree: int            # used when color_mode == SH
    sh_K: int
    color_mode: int
    eps2d: float
    near_plane: float
    far_plane: float
    radius_clip: float
    want_isect_ids: bool = False
    inv_depth: bool = False      # depth channel holds 1/z (GaussianRasterizer adapter)
    want_main_ids: bool = False  # per-pixel id of the dominant Gaussian (GaussianRasterizer adapter)
    tile_px: tuple = (16, 16)    # INTERNAL tile shape of the lists / tile kernels: (16, 16) = gsplat's, (32, 16) = wide (same pixels, fewer pairs)


class _Workspace:
    """Grow-only scratch per (device, stream); handed to the library as pointer + size."""

    def __init__(self):
        self._buf: dict[tuple, torch.Tensor] = {}
        self._lock = threading.Lock()

    def get(self, device: torch.device, nbytes: int) -> torch.Tensor:
        key = (device.index, _lib.raw_stream(device))
        with self._lock:
            buf = self._buf.get(key)
            if buf is None or buf.numel() < nbytes:
                # 288 GB of HBM: over-allocate by 1.5x so a slowly growing map never reallocates per step
                buf = torch.empty(int(nbytes * 1.5) + 4096, dtype=torch.uint8, device=device)
                self._buf[key] = buf
            return buf


_WS = _Workspace()
_TLS = threading.local()


@contextlib.contextmanager
def color_adam(state: dict | None):
    """While active, RasterizeGaussians.backward applies the sparse-Adam step of the split SH colours
    (f_dc / f_rest) inside the projection backward (adk_project_bwd_adam) instead of returning their gradients.
    state: {"f_dc", "f_rest": the parameter tensors; "m_dc", "v_dc", "m_rest", "v_rest": their moments;
    "lr_dc", "lr_rest": 0-dim device tensors; "betas": (b1, b2); "eps": float}.  Only takes effect when the tensors
    rendered are exactly those parameters; otherwise the gradients are returned as usual.  Process-wide, not
    thread-local: autograd runs backward nodes on its own device thread."""
    global _COLOR_ADAM
    prev = _COLOR_ADAM
    _COLOR_ADAM = state
    try:
        yield
    finally:
        _COLOR_ADAM = prev


_COLOR_ADAM: dict | None = None


def _color_adam_for(colors: torch.Tensor, rest: torch.Tensor):
    st = _COLOR_ADAM
    if st is None:
        return None
    ok = (colors.data_ptr() == st["f_dc"].data_ptr() and rest.data_ptr() == st["f_rest"].data_ptr()
          and all(st[k].is_contiguous() and st[k].dtype == torch.float32 and st[k].device == colors.device
                  for k in ("m_dc", "v_dc", "m_rest", "v_rest"))
          and st["m_dc"].numel() == colors.numel() and st["v_dc"].numel() == colors.numel()
          and st["m_rest"].numel() == rest.numel() and st["v_rest"].numel() == rest.numel()
          and st["lr_dc"].numel() == 1 and st["lr_rest"].numel() == 1 and st["lr_dc"].is_cuda and st["lr_rest"].is_cuda)
    return st if ok else None


def _small_inverse_check():
    """A singular 4x4 met by artdeco_amd.small_inverse's wrapped torch inverse surfaces at this (already existing) host wait."""
    si = sys.modules.get("artdeco_amd.small_inverse")
    if si is not None:
        si.check()


def _count_slot(device: torch.device):
    """Per-thread, per-device pinned int64 + event for the asynchronous read of n_isects."""
    slots = getattr(_TLS, "slots", None)
    if slots is None:
        slots = _TLS.slots = {}
    slot = slots.get(device.index)
    if slot is None:
        slot = slots[device.index] = (torch.empty(2, dtype=torch.int64).pin_memory(), torch.cuda.Event())
    return slot


_CAPACITY_HINT: dict[tuple, int] = {}
_GRAIN = 1 << 20


def _empty_rounded(n: int, **kw) -> torch.Tensor:
    """torch.empty(n) backed by an allocation rounded up to a multiple of 2^20 elements.  The intersection count changes a
    little every step; exact-size requests of ~15 MB land in different size classes of torch's caching allocator and every new
    class is a hipMalloc (milliseconds, and a stall of the whole stream): measured as 2.08 ms/step over 30 steps but 2.7 over 100.
    With rounded capacities the same few blocks are reused for the life of the scene."""
    cap = max(((int(n) + _GRAIN - 1) // _GRAIN) * _GRAIN, _GRAIN)
    return torch.empty(cap, **kw)[:n]


def _isect_capacity_guess(dev: torch.device, N: int, W: int, H: int, tpw: int = 16) -> int:
    """Capacity (entries) for the scatter that is launched before n_isects is known: 1.25 x the previous call at the same
    resolution, or 0 on the first call (the scatter then runs after the wait)."""
    prev = _CAPACITY_HINT.get((dev.index, W, H, tpw))
    return 0 if prev is None else ((int(prev * 1.25) + _GRAIN - 1) // _GRAIN) * _GRAIN


def _bin_global(lib, cfg, N, W, H, tile_w, tile_h, rec, depth_keys, gauss_ids, tiles_per_gauss, dev, stream, i32):
    """The global route: radix sort of the N depth keys, emit in depth order, stable radix sort by tile (raster_bin.hip).  Used when
    the tile histogram does not fit LDS (> 32768 tiles), when a tile holds more than 4 194 304 entries (lists above 8 192 take the long-list
    sort of the tile-local route since round 5; ADK_BIN_LONG=0 sends them here as before), or with ADK_BIN_LOCAL=0."""
    # The list size only depends on the projection: count it now, start its copy to pinned host memory,
    # and read it AFTER the depth sort has been enqueued -- the host waits for the count (an event), not
    # for the sort, so the stream never drains (upstream syncs on n_isects after isect_tiles).
    n_isects_dev = torch.empty(2, dtype=torch.int64, device=dev)
    rc = lib.adk_bin_count_isects(N, tiles_per_gauss.data_ptr(), n_isects_dev.data_ptr(), stream)
    _lib.check(rc, "adk_bin_count_isects")
    host_count, count_ready = _count_slot(dev)
    host_count[:1].copy_(n_isects_dev[:1], non_blocking=True)
    count_ready.record()

    sorted_ids = torch.empty(N, **i32)
    block_offs = torch.empty((N + 255) // 256 + 1, **i32)
    ws_bytes = lib.adk_bin_depth_workspace_bytes(N)
    ws = _WS.get(dev, ws_bytes)
    with _stage("bin_depth_order"):
        rc = lib.adk_bin_depth_order(N, depth_keys.data_ptr(), gauss_ids.data_ptr(), tiles_per_gauss.data_ptr(),
                                     sorted_ids.data_ptr(), block_offs.data_ptr(), n_isects_dev[1:].data_ptr(),
                                     ws.data_ptr(), ws.numel(), stream)
    _lib.check(rc, "adk_bin_depth_order")
    count_ready.synchronize()  # the one host wait of the pipeline (sizes the list)
    _small_inverse_check()
    n_isects = int(host_count[0])
    LAST_STATS.update(N=N, I=n_isects, width=W, height=H, tile_px=(16, 16))   # plain numbers only: no tensor is kept alive from here

    flatten_ids = _empty_rounded(n_isects, **i32)
    tile_ids = _empty_rounded(n_isects, **i32)
    offsets = torch.empty(tile_h, tile_w, **i32)
    ws_bytes = lib.adk_bin_tiles_workspace_bytes(n_isects)
    ws = _WS.get(dev, ws_bytes)
    with _stage("bin_tiles"):
        rc = lib.adk_bin_tiles(N, n_isects, sorted_ids.data_ptr(), block_offs.data_ptr(),
                               tiles_per_gauss.data_ptr(), rec.data_ptr(), W, H, flatten_ids.data_ptr(),
                               tile_ids.data_ptr(), offsets.data_ptr(), ws.data_ptr(), ws.numel(), stream)
    _lib.check(rc, "adk_bin_tiles")
    return flatten_ids, tile_ids, offsets, n_isects


def _bin_lists(lib, cfg, tile_px, want_tile_ids, N, W, H, rec, depth_keys, gauss_ids, tiles_per_gauss, dev, stream, i32, stats=True):
    """(flatten_ids, tile_ids | None, offsets, n_isects, tile_px_w, tile_px_h): every (Gaussian, tile) pair in (tile, depth, id) order for
    internal tiles of `tile_px`; the shape actually used comes back (the global route only produces gsplat's 16x16 lists)."""
    tpw, tph = tile_px
    use_local = bool(lib.adk_bin_local_supported_t(W, H, tpw, tph)) and os.environ.get("ADK_BIN_LOCAL", "1") != "0"
    if not use_local:
        tpw, tph = 16, 16                                   # the global route produces gsplat's 16x16 lists only
    tile_w, tile_h = (W + tpw - 1) // tpw, (H + tph - 1) // tph
    flatten_ids = tile_ids = offsets = None
    if use_local:
        # TILE-LOCAL route: counting sort by tile + one in-LDS sort per tile (raster_bin.hip).  The host needs n_isects
        # (it sizes flatten_ids) and the fullest tile (it picks the sort kernels); both arrive with one pinned copy, and
        # the scatter -- which only needs a CAPACITY -- is launched before the host waits, so the stream stays busy.
        offsets = torch.empty(tile_h, tile_w, **i32)
        stats_dev = torch.empty(2, dtype=torch.int64, device=dev)
        table = torch.empty(int(lib.adk_bin_local_workspace_bytes_t(W, H, tpw, tph)) + 256, dtype=torch.uint8, device=dev)
        tbase = (table.data_ptr() + 255) & ~255
        tbytes = table.numel() - (tbase - table.data_ptr())
        with _stage("bin_count"):
            rc = lib.adk_bin_local_count_t(N, tiles_per_gauss.data_ptr(), rec.data_ptr(), W, H, tpw, tph, offsets.data_ptr(),
                                           stats_dev.data_ptr(), tbase, tbytes, stream)
        _lib.check(rc, "adk_bin_local_count")
        host_count, count_ready = _count_slot(dev)
        host_count.copy_(stats_dev, non_blocking=True)
        count_ready.record()
        guess = _isect_capacity_guess(dev, N, W, H, tpw)
        pairs = torch.empty(guess, dtype=torch.int64, device=dev)  # guess is a multiple of 2^20: a stable size class
        if guess > 0:
            with _stage("bin_scatter"):
                rc = lib.adk_bin_local_scatter_t(N, guess, depth_keys.data_ptr(), tiles_per_gauss.data_ptr(), rec.data_ptr(),
                                                 W, H, tpw, tph, offsets.data_ptr(), tbase, tbytes, pairs.data_ptr(), stream)
            _lib.check(rc, "adk_bin_local_scatter")
        count_ready.synchronize()  # the one host wait of the pipeline
        _small_inverse_check()
        n_isects, max_tile = int(host_count[0]), int(host_count[1])
        _CAPACITY_HINT[(dev.index, W, H, tpw)] = n_isects   # keyed on the image only: one entry per resolution however the map grows
        if max_tile > int(lib.adk_bin_local_sort_long_max()) or (max_tile > 8192 and os.environ.get("ADK_BIN_LONG", "1") == "0"):
            use_local = False      # beyond the long-list sort's limit (4 M entries on one tile): global route below (16x16 lists)
            tpw, tph = 16, 16
        else:
            if n_isects > pairs.numel():   # the estimate was too small (first call / the map grew by > 25 %): scatter again
                pairs = _empty_rounded(n_isects, dtype=torch.int64, device=dev)
                rc = lib.adk_bin_local_scatter_t(N, n_isects, depth_keys.data_ptr(), tiles_per_gauss.data_ptr(), rec.data_ptr(),
                                                 W, H, tpw, tph, offsets.data_ptr(), tbase, tbytes, pairs.data_ptr(), stream)
                _lib.check(rc, "adk_bin_local_scatter")
            if stats:
                LAST_STATS.update(N=N, I=n_isects, width=W, height=H, tile_px=(tpw, tph))
            flatten_ids = _empty_rounded(n_isects, **i32)
            tile_ids = _empty_rounded(n_isects, **i32) if want_tile_ids else None
            # tile lists above 8 192 entries are sorted by recursive partition between `pairs` and a second buffer (round 5: no list-length cliff)
            scratch = _empty_rounded(n_isects, dtype=torch.int64, device=dev) if max_tile > 8192 else None
            with _stage("bin_sort"):
                rc = lib.adk_bin_local_sort_long_t(n_isects, max_tile, W, H, tpw, tph, offsets.data_ptr(), pairs.data_ptr(), _lib.ptr(scratch),
                                                   0 if scratch is None else scratch.numel() * 8, flatten_ids.data_ptr(), _lib.ptr(tile_ids), stream)
            _lib.check(rc, "adk_bin_local_sort_long")
    if not use_local:
        flatten_ids, tile_ids, offsets, n_isects = _bin_global(lib, cfg, N, W, H, (W + 15) // 16, (H + 15) // 16, rec, depth_keys, gauss_ids,
                                                               tiles_per_gauss, dev, stream, i32)
    return flatten_ids, tile_ids, offsets, n_isects, tpw, tph


def gsplat_tile_lists(rec, depth_keys, tiles_per_gauss, width, height):
    """(flatten_ids, isect_offsets [tile_h, tile_w]) for gsplat's own 16x16 tiles from a finished projection: what upstream's meta
    carries under those names.  The drop-in computes them on first access when the render itself used wider internal tiles."""
    lib = _lib.load()
    dev, N = rec.device, rec.shape[0]
    with _lib.on_device(dev):
        stream = _lib.raw_stream(dev)
        i32 = dict(dtype=torch.int32, device=dev)
        gauss_ids = torch.arange(N, **i32)
        flat, _t, offs, _n, _w, _h = _bin_lists(lib, None, (16, 16), False, N, width, height, rec, depth_keys, gauss_ids, tiles_per_gauss, dev,
                                                stream, i32, stats=False)
    return flat, offs


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError(f"rasterization: {name} must be float32, got {t.dtype}")
    return t.contiguous()


class RasterizeGaussians(torch.autograd.Function):
    """(means, quats, scales, opacities, colors, viewmat[4,4], K[3,3], backgrounds|None) ->
    render_colors [H,W,4], render_alphas [H,W,1], + non-differentiable aux tensors."""

    @staticmethod
    def forward(ctx, means, quats, scales, opacities, colors, sh_rest, viewmat, K, backgrounds, cfg: RasterConfig):
        lib = _lib.load()
        _lib.require_cuda(means, quats, scales, opacities, viewmat, K)
        dev = means.device
        N = means.shape[0]
        W, H = cfg.width, cfg.height
        means, quats, scales = _f32c(means, "means"), _f32c(quats, "quats"), _f32c(scales, "scales")
        opacities = _f32c(opacities, "opacities")
        colors_c = _f32c(colors, "colors") if colors is not None else None
        rest_c = _f32c(sh_rest, "sh_rest") if sh_rest is not None else None
        viewmat, K = _f32c(viewmat, "viewmats"), _f32c(K, "Ks")
        bg = _f32c(backgrounds, "backgrounds") if backgrounds is not None else None
        with _lib.on_device(dev):
            stream = _lib.raw_stream(dev)
            i32 = dict(dtype=torch.int32, device=dev)
            rec = torch.empty(N, 12, dtype=torch.float32, device=dev)
            radii = torch.empty(N, 2, **i32)
            depth_keys = torch.empty(N, **i32)
            gauss_ids = torch.empty(N, **i32)
            tiles_per_gauss = torch.empty(N, **i32)
            with _stage("project_fwd"):
              rc = lib.adk_project_fwd(N, means.data_ptr(), quats.data_ptr(), scales.data_ptr(), opacities.data_ptr(),
                                     _lib.ptr(colors_c), _lib.ptr(rest_c), cfg.sh_K, cfg.sh_degree, cfg.color_mode, viewmat.data_ptr(),
                                     K.data_ptr(), W, H, cfg.eps2d, cfg.near_plane, cfg.far_plane, cfg.radius_clip,
                                     int(cfg.inv_depth), rec.data_ptr(), radii.data_ptr(), depth_keys.data_ptr(), gauss_ids.data_ptr(),
                                     tiles_per_gauss.data_ptr(), stream)
            _lib.check(rc, "adk_project_fwd")

            # the rasteriser's outputs do not depend on the list size: allocated here, BEFORE the host wait below, so that what
            # stands between the wait and the forward rasteriser's launch is the sort's launch and nothing else
            render_colors = torch.empty(H, W, 4, dtype=torch.float32, device=dev)
            render_alphas = torch.empty(H, W, 1, dtype=torch.float32, device=dev)
            final_T = torch.empty(H, W, dtype=torch.float32, device=dev)   # exact T_final for the backward
            last_ids = torch.empty(H, W, **i32)
            main_ids = torch.empty(H, W, **i32) if cfg.want_main_ids else None
            flatten_ids, tile_ids, offsets, n_isects, tpw, tph = _bin_lists(lib, cfg, cfg.tile_px, cfg.want_isect_ids, N, W, H, rec, depth_keys,
                                                                            gauss_ids, tiles_per_gauss, dev, stream, i32)

            with _stage("raster_fwd"):
              rc = lib.adk_raster_fwd_t(W, H, tpw, tph, rec.data_ptr(), flatten_ids.data_ptr(), offsets.data_ptr(), n_isects,
                                      _lib.ptr(bg), render_colors.data_ptr(), render_alphas.data_ptr(), final_T.data_ptr(),
                                      last_ids.data_ptr(), _lib.ptr(main_ids), stream)
            _lib.check(rc, "adk_raster_fwd")

            # what the caller sees as gsplat's lists: the internal ones when they ARE 16x16, else (only on request) a second,
            # 16x16 binning of the same projection; the tile kernels and the backward keep using the internal lists
            isect_ids = torch.empty(0, dtype=torch.int64, device=dev)
            pub_flat, pub_offsets = flatten_ids, offsets
            if cfg.want_isect_ids:
                n_pub = n_isects
                if (tpw, tph) != (16, 16):
                    pub_flat, tile_ids, pub_offsets, n_pub, _w, _h = _bin_lists(lib, cfg, (16, 16), True, N, W, H, rec, depth_keys, gauss_ids,
                                                                               tiles_per_gauss, dev, stream, i32, stats=False)
                isect_ids = _empty_rounded(n_pub, dtype=torch.int64, device=dev)
                rc = lib.adk_bin_make_isect_ids(n_pub, tile_ids.data_ptr(), pub_flat.data_ptr(),
                                                depth_keys.data_ptr(), isect_ids.data_ptr(), stream)
                _lib.check(rc, "adk_bin_make_isect_ids")

        ctx.set_materialize_grads(False)  # no zero tensors for the eight non-differentiable by-products
        ctx.cfg = cfg
        ctx.n_isects = n_isects
        ctx.tile_px = (tpw, tph)
        ctx.has_bg = bg is not None
        ctx.has_rest = rest_c is not None
        ctx.save_for_backward(means, quats, scales, colors_c if colors_c is not None else means.new_empty(0),
                              rest_c if rest_c is not None else means.new_empty(0), viewmat, K, bg if bg is not None else means.new_empty(0), rec, radii, flatten_ids,
                              offsets, final_T, last_ids)
        aux = (radii, rec, tiles_per_gauss, pub_flat, pub_offsets, isect_ids, last_ids,
               main_ids if main_ids is not None else torch.empty(0, **i32), final_T, depth_keys)
        ctx.mark_non_differentiable(*aux)
        return (render_colors, render_alphas) + aux

    @staticmethod
    def backward(ctx, v_colors, v_alphas, *unused):
        lib = _lib.load()
        cfg: RasterConfig = ctx.cfg
        (means, quats, scales, colors, rest, viewmat, K, bg, rec, radii, flatten_ids, offsets, final_T,
         last_ids) = ctx.saved_tensors
        dev = means.device
        N = means.shape[0]
        W, H = cfg.width, cfg.height
        needs = ctx.needs_input_grad
        with _lib.on_device(dev):
            stream = _lib.raw_stream(dev)
            v_colors = (v_colors if v_colors is not None else torch.zeros(H, W, 4, device=dev)).contiguous()
            v_alphas = (v_alphas if v_alphas is not None else torch.zeros(H, W, 1, device=dev)).contiguous()
            v_rec = torch.zeros(N, 12, dtype=torch.float32, device=dev)
            with _stage("raster_bwd"):
              rc = lib.adk_raster_bwd_t(W, H, ctx.tile_px[0], ctx.tile_px[1], rec.data_ptr(), flatten_ids.data_ptr(), offsets.data_ptr(), ctx.n_isects,
                                    bg.data_ptr() if ctx.has_bg else None, final_T.data_ptr(),
                                    last_ids.data_ptr(), v_colors.data_ptr(), v_alphas.data_ptr(), v_rec.data_ptr(),
                                    stream)
            _lib.check(rc, "adk_raster_bwd")

            v_means = torch.empty_like(means) if needs[0] else None
            v_quats = torch.empty_like(quats) if needs[1] else None
            v_scales = torch.empty_like(scales) if needs[2] else None
            v_opac = torch.empty(N, dtype=torch.float32, device=dev) if needs[3] else None
            has_colors = cfg.color_mode != _COLOR_DEPTH
            want_col_grads = has_colors and (needs[4] or (ctx.has_rest and needs[5]))
            fuse_cols = want_col_grads and ctx.has_rest and cfg.color_mode == _COLOR_SH and _color_adam_for(colors, rest) is not None
            v_cols = torch.empty_like(colors) if (want_col_grads and not fuse_cols) else None
            v_rest = torch.empty_like(rest) if (want_col_grads and ctx.has_rest and not fuse_cols) else None
            v_viewmat = cam_grad = None
            if needs[6]:
                v_viewmat = torch.empty(4, 4, dtype=torch.float32, device=dev)
                cam_grad = _cam_grad_scratch(dev, stream)
            opt = _color_adam_for(colors, rest) if (want_col_grads and ctx.has_rest and cfg.color_mode == _COLOR_SH) else None
            if opt is not None:
                with _stage("project_bwd"):
                    rc = lib.adk_project_bwd_adam(N, means.data_ptr(), quats.data_ptr(), scales.data_ptr(), colors.data_ptr(),
                                                  rest.data_ptr(), cfg.sh_K, cfg.sh_degree, viewmat.data_ptr(), K.data_ptr(), W, H,
                                                  cfg.eps2d, cfg.near_plane, cfg.far_plane, int(cfg.inv_depth), radii.data_ptr(),
                                                  v_rec.data_ptr(), _lib.ptr(v_means), _lib.ptr(v_quats), _lib.ptr(v_scales),
                                                  _lib.ptr(v_opac), _lib.ptr(cam_grad), _lib.ptr(v_viewmat), opt["m_dc"].data_ptr(),
                                                  opt["v_dc"].data_ptr(), opt["m_rest"].data_ptr(), opt["v_rest"].data_ptr(),
                                                  opt["lr_dc"].data_ptr(), opt["lr_rest"].data_ptr(), float(opt["betas"][0]),
                                                  float(opt["betas"][1]), float(opt["eps"]), stream)
                _cam_grad_reset(cam_grad, rc)
                _lib.check(rc, "adk_project_bwd_adam")
                return v_means, v_quats, v_scales, v_opac, None, None, v_viewmat, None, None, None
            with _stage("project_bwd"):
              rc = lib.adk_project_bwd(N, means.data_ptr(), quats.data_ptr(), scales.data_ptr(),
                                     colors.data_ptr() if has_colors else None, rest.data_ptr() if ctx.has_rest else None, cfg.sh_K, cfg.sh_degree,
                                     cfg.color_mode, viewmat.data_ptr(), K.data_ptr(), W, H, cfg.eps2d,
                                     cfg.near_plane, cfg.far_plane, int(cfg.inv_depth), radii.data_ptr(), v_rec.data_ptr(),
                                     _lib.ptr(v_means), _lib.ptr(v_quats), _lib.ptr(v_scales), _lib.ptr(v_opac),
                                     _lib.ptr(v_cols), _lib.ptr(v_rest), _lib.ptr(cam_grad), _lib.ptr(v_viewmat), stream)
            _cam_grad_reset(cam_grad, rc)
            _lib.check(rc, "adk_project_bwd")
        return v_means, v_quats, v_scales, v_opac, v_cols, v_rest, v_viewmat, None, None, None


_CAM_GRAD: dict = {}


def _cam_grad_scratch(dev, stream) -> torch.Tensor:
    """The 16-double accumulator of the camera gradient: adk_project_bwd leaves it zeroed again, so one per (device,
    stream, THREAD) is cleared ONCE instead of a fill kernel per step.  Per thread: the accumulate and finalize kernels are two
    launches of one C call, and two threads issuing backward on one stream (the viewer thread renders under the same null
    stream, h3dgsv3.py:624) could otherwise interleave them -- accumulate(A), accumulate(B), finalize(A) -- on a shared buffer.
    A failed launch leaves the buffer dirty: _cam_grad_reset() clears it on any non-zero return code."""
    key = (dev.index, int(stream or 0), threading.get_ident())
    buf = _CAM_GRAD.get(key)
    if buf is None:
        buf = _CAM_GRAD[key] = torch.zeros(16, dtype=torch.float64, device=dev)   # fp64 accumulator since ABI v19
    return buf


def _cam_grad_reset(buf, rc: int) -> None:
    if rc != 0 and buf is not None:
        buf.zero_()


def default_tile_px() -> tuple:
    """Internal tile shape: ADK_TILE_SHAPE = "16x16" (default: gsplat's own tiles) or "32x16" (one wave per 32x16 tile: 26 % fewer
    (splat, tile) pairs to list, stage, reduce and flush, per-pixel results unchanged -- built and measured in round 3: the
    backward's VALU instructions fall by 8 %, but 4 080 waves of 138 VGPRs leave half as many waves resident and the kernels
    run 30 % SLOWER at 1 M / 1080p; profiles/r03_wide_tiles.txt).  `flatten_ids` / `isect_offsets` / `isect_ids` as upstream
    defines them (16x16) are produced separately when the internal shape differs and a caller asks for them."""
    v = os.environ.get("ADK_TILE_SHAPE", "16x16").lower()
    return (32, 16) if v == "32x16" else (16, 16)


def camera_config(colors, width, height, *, sh_degree, eps2d=0.3, near_plane=0.01, far_plane=1e10, radius_clip=0.0,
                  depth_only=False, want_isect_ids=False, inv_depth=False, want_main_ids=False, sh_rest=None, tile_px=None):
    """(RasterConfig, colours, sh_rest) of one render_camera call: the argument checks and the colour mode."""
    if depth_only:
        mode, K_sh, deg, cols = _COLOR_DEPTH, 0, 0, None
        sh_rest = None
    elif sh_degree is not None and sh_rest is not None:
        if colors.dim() != 3 or colors.shape[1:] != (1, 3) or sh_rest.dim() != 3 or sh_rest.shape[-1] != 3:
            raise ValueError("split SH colours must be f_dc [N,1,3] + f_rest [N,K-1,3]")
        K_sh, deg, mode, cols = 1 + sh_rest.shape[1], int(sh_degree), _COLOR_SH, colors
        if not 0 <= deg <= 3 or (deg + 1) ** 2 > K_sh:
            raise ValueError(f"sh_degree {deg} not supported by {K_sh} coefficients")
    elif sh_degree is not None:
        if colors.dim() != 3 or colors.shape[-1] != 3:
            raise ValueError(f"SH colours must be [N,K,3], got {tuple(colors.shape)}")
        K_sh, deg, mode, cols = colors.shape[1], int(sh_degree), _COLOR_SH, colors
        if not 0 <= deg <= 3:
            raise NotImplementedError("sh_degree must be in 0..3")
        if (deg + 1) ** 2 > K_sh:
            raise ValueError(f"sh_degree {deg} needs {(deg + 1) ** 2} coefficients, colors has {K_sh}")
    else:
        if colors.dim() != 2 or colors.shape[-1] != 3:
            raise NotImplementedError(f"post-activation colours must be [N,3], got {tuple(colors.shape)}")
        mode, K_sh, deg, cols = _COLOR_RGB, 0, 0, colors
    cfg = RasterConfig(int(width), int(height), deg, K_sh, mode, float(eps2d), float(near_plane),
                       float(far_plane), float(radius_clip), bool(want_isect_ids), bool(inv_depth), bool(want_main_ids),
                       tuple(tile_px) if tile_px is not None else default_tile_px())
    return cfg, cols, sh_rest


def render_camera(means, quats, scales, opacities, colors, viewmat, K, width, height, *, sh_degree,
                  eps2d=0.3, near_plane=0.01, far_plane=1e10, radius_clip=0.0, backgrounds=None,
                  depth_only=False, want_isect_ids=False, inv_depth=False, want_main_ids=False, sh_rest=None, tile_px=None):
    """One camera.  colors: SH coefficients [N,K,3] when sh_degree is not None, else RGB [N,3]
    (ignored when depth_only).  sh_rest: optional [N,K-1,3] -- then `colors` is band 0 only
    ([N,1,3], ARTDECO's f_dc) and no concatenation is materialised.  backgrounds: [4] or None."""
    cfg, cols, sh_rest = camera_config(colors, width, height, sh_degree=sh_degree, eps2d=eps2d, near_plane=near_plane,
                                       far_plane=far_plane, radius_clip=radius_clip, depth_only=depth_only,
                                       want_isect_ids=want_isect_ids, inv_depth=inv_depth, want_main_ids=want_main_ids,
                                       sh_rest=sh_rest, tile_px=tile_px)
    return RasterizeGaussians.apply(means, quats, scales, opacities, cols, sh_rest, viewmat, K, backgrounds, cfg)


class HandCtx:
    """What a torch.autograd.Function's forward / backward use of their `ctx`, for driving the two static methods by hand
    (artdeco_amd.fused runs the mapper's fixed chain LoD -> rasteriser -> loss that way: the autograd engine -- graph
    construction, the hop to its device thread and back, gradient accumulation -- was a third of the host time of a step)."""

    def __init__(self, needs_input_grad=()):
        self.needs_input_grad = tuple(needs_input_grad)
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def mark_non_differentiable(self, *tensors):
        pass

    def set_materialize_grads(self, value):
        pass
