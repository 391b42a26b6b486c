"""What happens automatically when ARTDECO imports one of the drop-in modules (nothing in ARTDECO's files is edited).

`run_system.py:17` imports `Reconstruct.scene.keyframe` -> `Reconstruct.scene.optimizers` -> `diff_gaussian_rasterization`
(this package's drop-in) long before it imports the scene-model module by name (`run_system.py:113`,
`__import__('Reconstruct.scene.scene_models.' + args.base_model)`).  The first drop-in import therefore installs a
POST-IMPORT hook: once a module `...scene.scene_models.<name>` has been executed, its `SceneModel.__init__` is wrapped so
that every constructed scene model gets `artdeco_amd.fused.patch_scene_model(self)` -- the fused HIP paths for `render`,
`render_from_id`, `optimization_step`, `optimizer.step` / `add_and_prune` and `weed_out_gaussians`, same signatures and
results (tests/test_fused_glue.py), installed only when the model has the supported shapes (`fused.supported`) AND the
source of every ARTDECO method a fused path mirrors still hashes to the version it was written against
(`artdeco_amd/pins.py`, `reference_pins.json`; a moved method disables its group with one warning).  So an
UNCHANGED run_system.py with `PYTHONPATH=<repo>:<repo>/artdeco_amd/dropin` runs the fused mapper step.

    ARTDECO_AMD_AUTOFUSE=0      natives only: ARTDECO's own torch glue around them (the "unchanged host code" bench lines)
    ARTDECO_AMD_AUTOTRACKER=1   also route `from VSLAM.CameraTracker import CameraTracker` (VSLAM/Frontend.py:9) to the device
                                tracker (artdeco_amd.tracker.install_tracker); off by default, it replaces a whole class
"""
from __future__ import annotations

import functools
import importlib.abc
import importlib.util
import os
import re
import sys

_SCENE_MODULE = re.compile(r"(^|\.)scene\.scene_models\.\w+$")
_installed = False


def _autofuse_enabled() -> bool:
    return os.environ.get("ARTDECO_AMD_AUTOFUSE", "1") != "0"


def _wrap_scene_model(module) -> None:
    cls = getattr(module, "SceneModel", None)
    if cls is None or getattr(cls, "_artdeco_amd_autofuse", False):
        return
    orig_init = cls.__init__

    @functools.wraps(orig_init)
    def __init__(self, *args, **kwargs):
        orig_init(self, *args, **kwargs)
        if _autofuse_enabled():
            from artdeco_amd import fused
            self._artdeco_amd_fused = bool(fused.patch_scene_model(self, verify=True))

    cls.__init__ = __init__
    cls._artdeco_amd_autofuse = True


class _LoaderProxy(importlib.abc.Loader):
    def __init__(self, inner, callback):
        self._inner, self._callback = inner, callback

    def create_module(self, spec):
        return self._inner.create_module(spec)

    def exec_module(self, module):
        self._inner.exec_module(module)
        self._callback(module)

    def __getattr__(self, name):
        return getattr(self._inner, name)


class _PostImportFinder(importlib.abc.MetaPathFinder):
    """Finds nothing itself: lets the regular finders locate a scene-model module and wraps its loader."""

    def __init__(self):
        self._busy: set[str] = set()

    def find_spec(self, fullname, path=None, target=None):
        if fullname in self._busy or not _SCENE_MODULE.search(fullname):
            return None
        self._busy.add(fullname)
        try:
            spec = importlib.util.find_spec(fullname)
        except (ImportError, ValueError):
            spec = None
        finally:
            self._busy.discard(fullname)
        if spec is None or spec.loader is None:
            return None
        spec.loader = _LoaderProxy(spec.loader, _wrap_scene_model)
        return spec


def on_dropin_import() -> None:
    """Called by every drop-in module when it is imported (idempotent)."""
    global _installed
    if _installed:
        return
    _installed = True
    if _autofuse_enabled():
        for name, mod in list(sys.modules.items()):   # scene-model modules that are already in (tests, notebooks)
            if mod is not None and _SCENE_MODULE.search(name) and hasattr(mod, "SceneModel"):
                _wrap_scene_model(mod)
        sys.meta_path.insert(0, _PostImportFinder())
    if os.environ.get("ARTDECO_AMD_AUTOTRACKER", "0") == "1":
        from artdeco_amd import tracker
        tracker.install_tracker()
