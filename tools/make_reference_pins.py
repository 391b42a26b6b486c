#!/usr/bin/env python
"""Write artdeco_amd/reference_pins.json: SHA-256 of the source of every ARTDECO host method the fused paths mirror
(artdeco_amd/pins.py), taken from the reference tree at /root/reference.  Re-run after re-validating the fused paths against a
changed ARTDECO (tests/test_reference_scene_model.py, tests/test_densify.py, tests/test_fused_glue.py).

    ARTDECO_AMD_AUTOFUSE=0 python tools/make_reference_pins.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ARTDECO_AMD_AUTOFUSE"] = "0"

from artdeco_amd import pins  # noqa: E402
from harness import ref_env  # noqa: E402


def main():
    mod = ref_env.import_scene_module()
    live = pins.collect(mod.SceneModel, mod.SparseGaussianAdam)
    missing = [k for k, v in live.items() if v is None]
    if missing:
        raise SystemExit(f"no source for {missing}")
    out = {"source": "InternRobotics/ARTDECO at /root/reference: Reconstruct/scene/scene_models/h3dgsv3.py, scene/optimizers.py, "
                     "scene/keyframe.py, Reconstruct/utils.py", "hash": "sha256(dedented source, trailing blanks stripped)", "pins": live}
    with open(pins.PIN_FILE, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print(f"wrote {pins.PIN_FILE}: {len(live)} pins")


if __name__ == "__main__":
    main()
