"""Frontend Sim(3) tracker on the device: host-side mirror of `VSLAM/CameraTracker.py`.

`CameraTracker` keeps the reference's constructor, attributes and `track(frame) -> (lost, is_keyframe,
is_keyframe_map)` contract (CameraTracker.py:19-155), so `VSLAM/Frontend.py:34,80` can use it unchanged; everything
between the MASt3R match and the keyframe decision is ONE C-ABI call (`adk_track_frame`, artdeco_amd/csrc/tracker.hip)
and ONE 32-float host read (more only when six iterations did not converge), where the reference issues ~100 torch launches and three blocking reads per
Gauss-Newton iteration.  Poses may be pypose `Sim3` LieTensors (the reference's type) or plain [1,8] tensors
(t, q xyzw, s).  There is no CPU fallback: CPU tensors raise.
"""
from __future__ import annotations

import dataclasses

import numpy as np
import torch

from . import _lib


@dataclasses.dataclass
class TrackOutcome:
    """Host view of `adk_track_frame`'s 32-float result (layout: include/artdeco_hip.h)."""
    T_WCf: torch.Tensor      # [8] device
    T_CkCf: torch.Tensor     # [8] device
    lost: bool
    failed: bool
    iterations: int
    n_opt: int
    n_kf: int
    n_unique: int
    dist_quantile: float
    cost: float
    fx: float = 0.0          # the focal lengths the iterations ended with (= K's unless optimize_focal)
    fy: float = 0.0


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError(f"tracker: {name} must be float32")
    return t.contiguous()


def raw_pose(T) -> torch.Tensor:
    """pypose Sim3 / duck-typed stand-in / plain tensor -> flat [8] float32 tensor."""
    t = T.tensor() if hasattr(T, "tensor") else T  # LieTensor.tensor() strips the Lie type; plain tensors have no such method
    return t.reshape(-1)[:8].to(torch.float32)


def wrap_pose(like, raw: torch.Tensor):
    """Give `raw` [8] the type of `like` (pypose Sim3 when `like` is a LieTensor)."""
    if isinstance(like, torch.Tensor) and not hasattr(like, "ltype"):
        return raw.reshape(like.shape).to(like.dtype)
    if hasattr(like, "ltype"):
        import pypose as pp
        return pp.Sim3(raw.reshape(1, 8))
    return type(like)(raw.reshape(1, 8))


class TrackJob:
    """One tracking problem on the device.  Construction enqueues the preparation and the first CHUNK of Gauss-Newton
    iterations (no host synchronisation); `outcome()` reads the 32-float result and, only if that chunk did not converge,
    enqueues further chunks (resume) up to cfg["max_iters"].  Inputs, workspace and result stay referenced here.

    Xf_canon / Cf / Qf: the frame's canonical pointmap [n,3], SUMMED confidence [n] (average = Cf / Nf) and descriptor
    confidence [n]; Xk_canon / Ck / Nk / Qk the keyframe's (Qk = Qkf of the match); idx_f2k [n] int64, valid_match [n]
    bool in keyframe pixel order; T_WCf / T_WCk [8]; cfg = config["tracking"] (config/base.yaml:19-34).
    chunk: iterations per call (None = all of cfg["max_iters"] at once, i.e. never more than one host read)."""

    def __init__(self, height, width, K, Xf_canon, Cf, Nf, Qf, Xk_canon, Ck, Nk, Qk, idx_f2k, valid_match, T_WCf, T_WCk, cfg,
                 covariance_filter=True, thres_keyframe=0.8, debug=False, chunk=6, optimize_focal=False):
        n = int(height) * int(width)
        _lib.require_cuda(K, Xf_canon, Cf, Qf, Xk_canon, Ck, Qk, idx_f2k, valid_match, T_WCf, T_WCk)
        dev = Xf_canon.device
        K = _f32c(K.to(dev), "K")
        Xf_canon, Xk_canon = _f32c(Xf_canon, "Xf_canon"), _f32c(Xk_canon, "Xk_canon")
        Cf, Ck, Qf, Qk = (_f32c(t.reshape(-1), s) for t, s in ((Cf, "Cf"), (Ck, "Ck"), (Qf, "Qf"), (Qk, "Qk")))
        if Xf_canon.shape != (n, 3) or Xk_canon.shape != (n, 3) or any(t.numel() != n for t in (Cf, Ck, Qf, Qk)):
            raise ValueError("tracker: pointmaps [H*W,3] and confidences [H*W] expected")
        if idx_f2k.dtype != torch.int64 or idx_f2k.numel() != n:
            raise TypeError("tracker: idx_f2k must be int64 [H*W]")
        if valid_match.dtype != torch.bool or valid_match.numel() != n:
            raise TypeError("tracker: valid_match must be bool [H*W]")
        idx_f2k, valid_match = idx_f2k.reshape(-1).contiguous(), valid_match.reshape(-1).contiguous()
        T_WCf, T_WCk = _f32c(T_WCf.reshape(-1), "T_WCf"), _f32c(T_WCk.reshape(-1), "T_WCk")
        if K.numel() != 9 or T_WCf.numel() != 8 or T_WCk.numel() != 8:
            raise ValueError("tracker: K [3,3] and poses [8] expected")
        self.height, self.width, self.n, self.dev = int(height), int(width), n, dev
        self.cfg, self.cov, self.thres = cfg, bool(covariance_filter), float(thres_keyframe)
        self.focal = bool(optimize_focal)
        self.max_iters = int(cfg["max_iters"])
        self.chunk = self.max_iters if chunk is None else max(1, int(chunk))
        self.inv_Nf, self.inv_Nk = 1.0 / float(Nf), 1.0 / float(Nk)
        self._t = (K, Xf_canon, Cf, Qf, Xk_canon, Ck, Qk, idx_f2k, valid_match, T_WCf, T_WCk)  # keep alive across resumes
        self.lib = _lib.load()
        with torch.cuda.device(dev):
            self.result = torch.empty(32, dtype=torch.float32, device=dev)
            self.dbg = {}
            if debug:
                self.dbg = dict(Xc=torch.empty(n, 3, dtype=torch.float32, device=dev), var=torch.empty(n, 3, dtype=torch.float32, device=dev),
                                valid_opt=torch.empty(n, dtype=torch.uint8, device=dev), acc0=torch.zeros(45, dtype=torch.float32, device=dev))
            self.ws = torch.empty(int(self.lib.adk_track_workspace_bytes(self.height, self.width)), dtype=torch.uint8, device=dev)
        self.enqueued = 0
        self.host_reads = 0
        self._enqueue(resume=False)

    def _enqueue(self, resume: bool) -> None:
        K, Xf_canon, Cf, Qf, Xk_canon, Ck, Qk, idx_f2k, valid_match, T_WCf, T_WCk = self._t
        cfg, dbg = self.cfg, ({} if resume else self.dbg)
        num = min(self.chunk, self.max_iters - self.enqueued)
        with torch.cuda.device(self.dev):
            rc = self.lib.adk_track_frame(
                self.height, self.width, K.data_ptr(), Xf_canon.data_ptr(), Cf.data_ptr(), self.inv_Nf, Qf.data_ptr(), Xk_canon.data_ptr(),
                Ck.data_ptr(), self.inv_Nk, Qk.data_ptr(), idx_f2k.data_ptr(), valid_match.data_ptr(), T_WCf.data_ptr(), T_WCk.data_ptr(),
                float(cfg["sigma_pixel"]), float(cfg["sigma_depth"]), float(cfg["huber"]), float(cfg["C_conf"]), float(cfg["Q_conf"]),
                float(cfg["min_match_frac"]), int(cfg["pixel_border"]), float(cfg["depth_eps"]), float(cfg["rel_error"]),
                float(cfg["delta_norm"]), int(num), int(self.cov), int(self.focal), self.thres, int(resume), self.result.data_ptr(),
                _lib.ptr(dbg.get("Xc")),
                _lib.ptr(dbg.get("var")), _lib.ptr(dbg.get("valid_opt")), _lib.ptr(dbg.get("acc0")), self.ws.data_ptr(), self.ws.numel(),
                _lib.stream_of(Xf_canon))
        _lib.check(rc, "adk_track_frame")
        self.enqueued += num

    def outcome(self) -> "TrackOutcome":
        """The host synchronisation(s) of a tracked frame: one 128-byte copy per enqueued chunk (normally one)."""
        while True:
            h = self.result.cpu().numpy()
            self.host_reads += 1
            if h[24] != 0 or self.enqueued >= self.max_iters:
                break
            self._enqueue(resume=True)
        r = self.result
        return TrackOutcome(T_WCf=r[0:8], T_CkCf=r[8:16], lost=bool(h[16] != 0), failed=bool(h[17] != 0), iterations=int(h[18]),
                            n_opt=int(h[19]), n_kf=int(h[20]), n_unique=int(h[21]), dist_quantile=float(h[22]), cost=float(h[23]),
                            fx=float(h[26]), fy=float(h[27]))


def track_frame(height, width, K, Xf_canon, Cf, Nf, Qf, Xk_canon, Ck, Nk, Qk, idx_f2k, valid_match, T_WCf, T_WCk, cfg,
                covariance_filter=True, thres_keyframe=0.8, debug=False, chunk=None, optimize_focal=False):
    """Run one tracking problem to completion; returns (result [32] device tensor, debug dict).  With chunk=None all
    cfg["max_iters"] iterations are enqueued at once and nothing synchronises; with a chunk size the host reads the
    result between chunks (see TrackJob)."""
    job = TrackJob(height, width, K, Xf_canon, Cf, Nf, Qf, Xk_canon, Ck, Nk, Qk, idx_f2k, valid_match, T_WCf, T_WCk, cfg,
                   covariance_filter, thres_keyframe, debug, chunk, optimize_focal)
    if chunk is not None:
        job.outcome()
    return job.result, job.dbg


def fuse_pointmap(result, Xkf, Ckf, X_canon, C):
    """In place: X_canon, C <- ImageFrame.update_pointmap(T_CkCf.Act(Xkf), Ckf) (CameraTracker.py:136-141); a no-op on
    the device when `result` says lost / failed."""
    _lib.require_cuda(result, Xkf, Ckf, X_canon, C)
    n = X_canon.shape[0]
    Xkf, Ckf = _f32c(Xkf, "Xkf"), _f32c(Ckf.reshape(-1), "Ckf")
    if not (X_canon.is_contiguous() and C.is_contiguous()) or X_canon.dtype != torch.float32 or C.dtype != torch.float32:
        raise TypeError("tracker: X_canon / C must be contiguous float32 (updated in place)")
    if Xkf.shape != (n, 3) or Ckf.numel() != n or C.numel() != n:
        raise ValueError("tracker: Xkf [n,3], Ckf [n], C [n] expected")
    with torch.cuda.device(X_canon.device):
        rc = _lib.load().adk_track_fuse_pointmap(n, result.data_ptr(), Xkf.data_ptr(), Ckf.data_ptr(), X_canon.data_ptr(), C.data_ptr(),
                                                 _lib.stream_of(X_canon))
    _lib.check(rc, "adk_track_fuse_pointmap")


def read_outcome(result: torch.Tensor) -> TrackOutcome:
    """Host view of a FINISHED result (one 128-byte copy)."""
    h = result.cpu().numpy()
    return TrackOutcome(T_WCf=result[0:8], T_CkCf=result[8:16], lost=bool(h[16] != 0), failed=bool(h[17] != 0), iterations=int(h[18]),
                        n_opt=int(h[19]), n_kf=int(h[20]), n_unique=int(h[21]), dist_quantile=float(h[22]), cost=float(h[23]),
                        fx=float(h[26]), fy=float(h[27]))


def keyframe_decisions(o: TrackOutcome, n: int, match_frac_thresh: float, min_displacement: float, last_dist: float):
    """check_keyframe (CameraTracker.py:159-167) and check_keyframe_map (:170-186) from the device counts, with the
    reference's mixed float32-tensor / python-float comparisons.  Returns (is_keyframe, is_keyframe_map, new last_dist)."""
    match_frac_k = np.float32(o.n_kf) / np.float32(n)          # tensor / int -> float32
    unique_frac_f = o.n_unique / n                               # python float
    if np.float32(unique_frac_f) < match_frac_k:                 # python min(tensor, float)
        add_new_kf = unique_frac_f < match_frac_thresh
    else:
        add_new_kf = bool(match_frac_k < np.float32(match_frac_thresh))
    if add_new_kf:
        return True, True, 0
    is_map = (o.dist_quantile - last_dist) > min_displacement
    return False, bool(is_map), (o.dist_quantile if is_map else last_dist)


class CameraTracker:
    """Drop-in for `VSLAM.CameraTracker.CameraTracker` (same constructor and `track` contract).  `match_fn` /
    `inference_mono_fn` default to the reference's `VSLAM.utils_mast3r.mast3r_match_asymmetric` /
    `mast3r_inference_mono` (resolved lazily, i.e. inside ARTDECO's tree); tests inject their own."""

    def __init__(self, args, config, min_displacement, thres_keyframe, model, frames, H_slam, W_slam, K_slam, device,
                 match_fn=None, inference_mono_fn=None):
        self.config = config
        self.cfg = config["tracking"]
        self.model = model
        self.keyframes = frames
        self.device = device
        self.H_slam = H_slam
        self.W_slam = W_slam
        self.K_slam = K_slam
        self.min_displacement = min_displacement
        self.thres_keyframe = thres_keyframe
        self.optimize_focal = args.optimize_focal
        self.covariance_filter = args.covariance_filter
        self.point_fusion_frontend = args.point_fusion_frontend
        self._match_fn = match_fn
        self._mono_fn = inference_mono_fn
        self.last_embedding = None
        self.last_dist = 0
        self.last_outcome = None
        self.iters_per_call = 6  # Gauss-Newton iterations enqueued per host read (typical convergence: 3-5)
        self.reset_idx_f2k()

    def _match(self):
        if self._match_fn is None:
            from VSLAM.utils_mast3r import mast3r_match_asymmetric
            self._match_fn = mast3r_match_asymmetric
        return self._match_fn

    def _mono(self):
        if self._mono_fn is None:
            from VSLAM.utils_mast3r import mast3r_inference_mono
            self._mono_fn = mast3r_inference_mono
        return self._mono_fn

    def track_init(self, frame):
        X_init, C_init, feat, pos = self._mono()(self.model, frame)
        frame.update_pointmap(X_init, C_init)
        self.last_embedding = [feat, pos]
        return False, True, True

    def reset_idx_f2k(self):
        self.idx_f2k = None

    def track(self, frame):
        if frame.frame_id == 0:
            return self.track_init(frame)
        keyframe = self.keyframes.last_keyframe().to(self.device)
        idx_f2k, valid_match_k, Xff, Cff, Qff, Xkf, Ckf, Qkf, featf, posf = self._match()(
            self.config, self.model, frame, keyframe, idx_i2j_init=self.idx_f2k, embeddings_j=self.last_embedding)
        self.idx_f2k = idx_f2k.clone()
        frame.update_pointmap(Xff, Cff)
        n = self.H_slam * self.W_slam
        job = TrackJob(self.H_slam, self.W_slam, self.K_slam, frame.X_canon, frame.C, frame.N, Qff, keyframe.X_canon, keyframe.C,
                       keyframe.N, Qkf, idx_f2k[0], valid_match_k[0], raw_pose(frame.T_WC).to(self.device),
                       raw_pose(keyframe.T_WC).to(self.device), self.cfg, self.covariance_filter, self.thres_keyframe,
                       chunk=self.iters_per_call, optimize_focal=self.optimize_focal)
        o = self.last_outcome = job.outcome()
        if self.optimize_focal and not o.lost:
            # the reference updates self.K_slam in place after every iteration (CameraTracker.py:376-377), also when a later
            # Cholesky fails; the device keeps the running focal in its state and reports it in result[26..27]
            with torch.no_grad():
                self.K_slam[0, 0] = job.result[26].to(self.K_slam.device)
                self.K_slam[1, 1] = job.result[27].to(self.K_slam.device)
        X_new = C_new = None
        if self.point_fusion_frontend and not (o.lost or o.failed):
            X_new, C_new = keyframe.X_canon.clone().contiguous(), keyframe.C.clone().contiguous()
            fuse_pointmap(job.result, Xkf, Ckf, X_new, C_new)
        if o.lost:
            print(f"Insufficient match {frame.frame_id}")
            return True, False, False
        if o.failed:
            print(f"Cholesky failed {frame.frame_id}")
            return True, False, False
        frame.T_WC = wrap_pose(frame.T_WC, o.T_WCf.clone())
        if self.point_fusion_frontend:
            keyframe.X_canon, keyframe.C = X_new, C_new.reshape(keyframe.C.shape)
            keyframe.N += 1
            keyframe.N_updates += 1
            self.keyframes[len(self.keyframes) - 1] = keyframe
        is_keyframe, is_keyframe_map, last_dist = keyframe_decisions(o, n, self.cfg["match_frac_thresh"], self.min_displacement,
                                                                      self.last_dist)
        if is_keyframe:
            self.reset_idx_f2k()
            self.last_embedding = [featf, posf]
        self.last_dist = last_dist
        return False, is_keyframe, is_keyframe_map


def install_tracker() -> None:
    """Make `from VSLAM.CameraTracker import CameraTracker` (VSLAM/Frontend.py:9) resolve to this class."""
    import sys
    import types
    mod = types.ModuleType("VSLAM.CameraTracker")
    mod.CameraTracker = CameraTracker
    mod.__doc__ = "artdeco_amd drop-in for VSLAM/CameraTracker.py"
    sys.modules["VSLAM.CameraTracker"] = mod
