"""Source pins: which version of ARTDECO's host methods the fused paths mirror.

`artdeco_amd.fused` replaces bodies of ARTDECO's own classes (SceneModel.render / render_from_id / optimization_step /
update_voxel / weed_out_gaussians / add_new_gaussians / rigid_transform_gs, SparseGaussianAdam.step / add_and_prune) and re-implements what
Keyframe.step / get_Rt do inside them.  A replacement is only valid for the source it was written against: if ARTDECO's
`optimization_step` gains a loss term or a constant changes, a silently installed fused step would train something else at
full speed.  So the SHA-256 of `inspect.getsource` of every mirrored method is recorded (`reference_pins.json`, written by
`tools/make_reference_pins.py` from /root/reference) and compared with the classes that are actually live in the process
before anything is swapped; a group with a mismatch is NOT installed (the natives under ARTDECO's own code still are) and one
line on stderr names the methods that moved.

Groups (a group is installed only when all of its methods match):
  step     the per-iteration path  h3dgsv3.py:401-469, 595-700; optimizers.py:77-161; keyframe.py:150-154, 186-191
  densify  the important-frame path  h3dgsv3.py:227-316, 766-953; optimizers.py:163-219; utils.py:93-108, 188-216
  pose     Keyframe.get_Rt / set_Rt  keyframe.py:144-159; utils.py:223-229
"""
from __future__ import annotations

import hashlib
import inspect
import json
import os
import sys
import textwrap

PIN_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_pins.json")

# group -> [(where, attribute)]: where = "scene" (the SceneModel class), "optimizer" (class of scene.optimizer),
# "keyframe" / "utils" (names bound in the scene-model module: `Keyframe`, and the free functions it imported),
# "kfutils" (a free function bound in the module that defines Keyframe).  A method may sit in both groups.
GROUPS = {
    "step": [("scene", "render"), ("scene", "render_from_id"), ("scene", "optimization_step"), ("optimizer", "step"),
             ("keyframe", "step"), ("keyframe", "get_Rt"), ("utils", "radial_decay_kernel")],
    "densify": [("scene", "update_voxel"), ("scene", "weed_out_gaussians"), ("scene", "add_new_gaussians"),
                ("scene", "make_dummy_ext_tensor"), ("scene", "rigid_transform_gs"), ("optimizer", "add_and_prune"),
                ("utils", "update_gaussians"), ("utils", "get_lapla_norm"),
                ("utils", "sample"), ("utils", "make_torch_sampler"), ("utils", "depth2points"), ("utils", "RGB2SH"), ("utils", "inverse_sigmoid"),
                # fused_add_new_gaussians uses artdeco_amd's PoseRt where the reference calls Keyframe.get_R / get_t / get_Rt -> sixD2mtx
                # (keyframe.py:144-154, utils.py:223), and replicates make_torch_sampler's uv * 2 / (size - 1) - 1 (utils.py:203-212)
                ("keyframe", "get_R"), ("keyframe", "get_t"), ("keyframe", "get_Rt"), ("kfutils", "sixD2mtx")],
    # round 5: Keyframe.get_Rt / set_Rt themselves as one launch each (fused.patch_keyframe_class), for run_system.py's SLAM-keyframe loop
    "pose": [("keyframe", "get_R"), ("keyframe", "get_t"), ("keyframe", "get_Rt"), ("keyframe", "set_Rt"), ("kfutils", "sixD2mtx")],
}


def source_sha(obj) -> str | None:
    """SHA-256 of the dedented source with trailing blanks removed; None when the source is unavailable."""
    try:
        obj = inspect.unwrap(getattr(obj, "__func__", obj))
        src = inspect.getsource(obj)
    except (OSError, TypeError):
        return None
    src = "\n".join(line.rstrip() for line in textwrap.dedent(src).strip().splitlines())
    return hashlib.sha256(src.encode()).hexdigest()


def load_pins() -> dict:
    try:
        with open(PIN_FILE) as f:
            return json.load(f)["pins"]
    except (OSError, KeyError, ValueError):
        return {}


def _resolve(scene_cls, optimizer_cls, where: str, attr: str):
    if where == "scene":
        return getattr(scene_cls, attr, None)
    if where == "optimizer":
        return getattr(optimizer_cls, attr, None) if optimizer_cls is not None else None
    mod = sys.modules.get(scene_cls.__module__)
    if where in ("keyframe", "kfutils"):
        kf = getattr(mod, "Keyframe", None)
        if kf is None:
            return None
        if where == "keyframe":
            # a class this package has already patched (fused.patch_keyframe_class) is verified by the body it kept, not by the replacement
            return getattr(kf, "_unfused_" + attr, None) or getattr(kf, attr, None)
        return getattr(sys.modules.get(kf.__module__), attr, None)
    return getattr(mod, attr, None)


def pin_name(where: str, attr: str) -> str:
    return {"scene": "SceneModel", "optimizer": "SparseGaussianAdam", "keyframe": "Keyframe", "utils": "utils", "kfutils": "utils"}[where] + "." + attr


def collect(scene_cls, optimizer_cls) -> dict:
    """{pin name: sha} of the live classes (what tools/make_reference_pins.py stores)."""
    out = {}
    for members in GROUPS.values():
        for where, attr in members:
            obj = _resolve(scene_cls, optimizer_cls, where, attr)
            out[pin_name(where, attr)] = source_sha(obj) if obj is not None else None
    return out


_warned: set = set()


def verify(scene, pins: dict | None = None) -> dict:
    """{group: [names whose live source differs from the pin]} for the classes behind this scene-model instance.
    A method that is absent from both the pin file and the class does not count; one that is pinned but has no retrievable
    source (or the other way round) does."""
    pins = load_pins() if pins is None else pins
    scene_cls = type(scene)
    opt = getattr(scene, "optimizer", None)
    optimizer_cls = type(opt) if opt is not None else getattr(sys.modules.get(scene_cls.__module__), "SparseGaussianAdam", None)
    live = collect(scene_cls, optimizer_cls)
    bad: dict = {}
    for group, members in GROUPS.items():
        moved = [pin_name(w, a) for w, a in members if live.get(pin_name(w, a)) != pins.get(pin_name(w, a))]
        if moved:
            bad[group] = moved
    return bad


def warn_once(bad: dict) -> None:
    """One line on stderr per (group, methods) -- and, with ARTDECO_AMD_REQUIRE_FUSED=1, an exception instead: a deployment that counts on
    the fused path (5 frames/s without it, 36 with it at 1 M / 1080p) should stop at start-up when an upstream edit has unpinned a method,
    not run eight times slower behind a warning."""
    if bad and os.environ.get("ARTDECO_AMD_REQUIRE_FUSED", "0") == "1":
        raise RuntimeError("artdeco_amd: ARTDECO_AMD_REQUIRE_FUSED=1 and the host methods "
                           + "; ".join(f"{g}: {', '.join(n)}" for g, n in bad.items())
                           + " differ from the pinned ARTDECO sources (artdeco_amd/reference_pins.json): the fused path(s) would not be "
                             "installed.  Re-validate and re-pin with tools/make_reference_pins.py, or unset the variable to run ARTDECO's own code "
                             "on the native operators.")
    for group, names in bad.items():
        key = (group, tuple(names))
        if key in _warned:
            continue
        _warned.add(key)
        # not warnings.warn(): SceneModel.__init__ itself runs warnings.filterwarnings("ignore") (h3dgsv3.py:97) before the
        # post-import hook gets to speak
        print(f"[artdeco_amd] WARNING: the host methods {', '.join(names)} differ from the ARTDECO sources the fused '{group}' "
              f"path was written against (artdeco_amd/reference_pins.json); that path is NOT installed -- ARTDECO's own code runs "
              f"on the native operators instead.  Re-validate and re-pin with tools/make_reference_pins.py.", file=sys.stderr, flush=True)
