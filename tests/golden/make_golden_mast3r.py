"""Generate tests/golden/mast3r_tiny.npz with the REFERENCE's AsymmetricMASt3R
(VSLAM/thirdparty/mast3r/mast3r/model.py:31, dust3r/dust3r/model.py:45, croco blocks, DPT head) in a tiny
configuration.  Weights are not stored: every parameter `name` is filled from a generator seeded with
crc32(name) (see `fill_by_name`), and the test fills artdeco_amd.mast3r_model the same way -- which pins both
the architecture and the state-dict naming.  Build container only (needs /root/reference)."""
import os
import sys
import zlib

import numpy as np
import torch

R = "/root/reference/VSLAM/thirdparty/mast3r"
for p in (R, R + "/dust3r", R + "/dust3r/croco"):
    sys.path.insert(0, p)
OUT = os.path.dirname(os.path.abspath(__file__))
CFG = dict(enc_embed_dim=64, enc_depth=2, enc_num_heads=4, dec_embed_dim=48, dec_depth=12, dec_num_heads=4)


def fill_by_name(model, scale=0.05):
    with torch.no_grad():
        seen = set()
        for name, p in list(model.named_parameters(remove_duplicate=False)) + list(model.named_buffers()):
            if id(p) in seen:
                continue
            seen.add(id(p))
        for name, p in model.state_dict().items():
            g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
            v = torch.randn(p.shape, generator=g) * (0.2 * scale if ".dpt." in name else scale)  # keep exp() heads finite
            if name.endswith("weight") and p.dim() == 1:  # LayerNorm weights around 1
                v = 1.0 + v
            p.copy_(v.to(p.dtype))


# Second configuration: 64-wide attention heads (the released model's head size, and the only one csrc/attention.hip runs) at toy
# widths -- 2 heads in the encoder, 1 in the decoder -- on a 96x128 image (48 tokens: less than one 64-key tile), so that the
# reference model pins the hand-written attention / LayerNorm path of the TF32-class mode and not only the torch fallbacks.
CFG_D64 = dict(enc_embed_dim=128, enc_depth=2, enc_num_heads=2, dec_embed_dim=64, dec_depth=12, dec_num_heads=1)


def main():
    generate("mast3r_tiny.npz", CFG, (48, 64))
    generate("mast3r_d64.npz", CFG_D64, (96, 128))


def generate(fname, cfg, hw):
    from mast3r.model import AsymmetricMASt3R
    inf = float("inf")
    net = AsymmetricMASt3R(pos_embed="RoPE100", patch_embed_cls="PatchEmbedDust3R", img_size=hw, head_type="catmlp+dpt",
                           output_mode="pts3d+desc24", depth_mode=("exp", -inf, inf), conf_mode=("exp", 1, inf),
                           two_confs=True, desc_conf_mode=("exp", 0, inf), landscape_only=False, **cfg).eval()
    fill_by_name(net)
    g = torch.Generator().manual_seed(0)
    img1 = torch.rand(1, 3, *hw, generator=g) * 2 - 1
    img2 = torch.rand(1, 3, *hw, generator=g) * 2 - 1
    shp = torch.tensor([list(hw)])
    with torch.no_grad():
        f1, pos1, _ = net._encode_image(img1, shp)
        f2, pos2, _ = net._encode_image(img2, shp)
        dec1, dec2 = net._decoder(f1, pos1, f2, pos2)
        dec1, dec2 = list(dec1), list(dec2)
        r1 = net._downstream_head(1, [t.float() for t in dec1], shp)
        r2 = net._downstream_head(2, [t.float() for t in dec2], shp)
    names = sorted(net.state_dict().keys())
    np.savez_compressed(os.path.join(OUT, fname), img1=img1.numpy(), img2=img2.numpy(), feat1=f1.numpy(),
                        dec1_last=dec1[-1].numpy(), dec2_last=dec2[-1].numpy(),
                        pts3d1=r1["pts3d"].numpy(), conf1=r1["conf"].numpy(), desc1=r1["desc"].numpy(), desc_conf1=r1["desc_conf"].numpy(),
                        pts3d2=r2["pts3d"].numpy(), conf2=r2["conf"].numpy(), desc2=r2["desc"].numpy(), desc_conf2=r2["desc_conf"].numpy(),
                        state_names=np.array(names), state_shapes=np.array([str(tuple(net.state_dict()[n].shape)) for n in names]))
    print("ok", fname, f1.shape, r1["pts3d"].shape, len(names))


if __name__ == "__main__":
    main()
