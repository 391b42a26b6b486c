"""Generate tests/golden/rope_*.npz with the REFERENCE's pure-torch RoPE2D
(VSLAM/thirdparty/mast3r/dust3r/croco/models/pos_embed.py:112-158; the class the reference itself
falls back to when the CUDA extension is missing).  Build container only."""
import importlib.util
import os

import numpy as np
import torch

REF = "/root/reference/VSLAM/thirdparty/mast3r/dust3r/croco/models/pos_embed.py"
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    spec = importlib.util.spec_from_file_location("ref_pos_embed", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)  # prints the "cannot find cuda-compiled version" warning: expected
    rope = mod.RoPE2D(freq=100.0, F0=1.0)
    for name, (B, Hh, N, D, gh, gw, seed) in {"rope_vitl": (1, 4, 24 * 32 // 8, 64, 12, 8, 0), "rope_small": (2, 3, 15, 32, 5, 3, 1)}.items():
        g = torch.Generator().manual_seed(seed)
        tokens = torch.randn(B, Hh, N, D, generator=g)            # [B, heads, N, D] as the module expects
        ys, xs = torch.meshgrid(torch.arange(gh), torch.arange(gw), indexing="ij")
        pos = torch.stack([ys.reshape(-1), xs.reshape(-1)], -1)[None].repeat(B, 1, 1)[:, :N].contiguous()
        out = rope(tokens.clone(), pos)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), tokens=tokens.numpy(), positions=pos.numpy(), out=out.numpy())
        print(name, tokens.shape, float(out.abs().mean()))


if __name__ == "__main__":
    main()
