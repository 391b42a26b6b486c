"""MASt3R (AsymmetricMASt3R, ViT-L encoder + two 12-block decoders + catmlp/DPT heads) on PyTorch-ROCm.

SURVEY.md 8 a8: the frontend's one dense-contraction workload.  ARTDECO runs the vendored model code
(VSLAM/thirdparty/mast3r) as is; that code cannot travel to the GPU box and must not be copied, so this
is a from-scratch restatement of the ARCHITECTURE with the reference's parameter names, so that a real
`MASt3R_ViTLarge_BaseDecoder_512_catmlpdpt_metric` state dict loads with `load_state_dict` and the
entry points ARTDECO uses keep their names and return layout:

    _encode_image(img, true_shape)           dust3r/dust3r/model.py:127-140
    _decoder(f1, pos1, f2, pos2)             dust3r/dust3r/model.py:172-191
    _downstream_head(head_num, decout, shp)  dust3r/dust3r/model.py:193-197 -> mast3r/catmlp_dpt_head.py:71-96
    forward(view1, view2)                    dust3r/dust3r/model.py:199-211

Blocks follow croco/models/blocks.py:81-191 (pre-norm ViT blocks, RoPE2D on q/k, decoder blocks with
self + cross attention and a normalised memory), the DPT adapter croco/models/dpt_block.py +
dust3r/heads/dpt_head.py:20-57, post-processing dust3r/heads/postprocess.py and
mast3r/catmlp_dpt_head.py:19-40.  Hyper-parameters of the released checkpoint:
thirdparty/mast3r/README.md:277.

MI355X specifics: in the fp16-operand inference mode attention runs on the hand-written MFMA kernel of csrc/attention.hip
(one workgroup per 64 query rows and head, transposed products, LDS transpose reads; 768 tokens fill 192 of the 256 CUs
where the library flash kernel launches 96 workgroups), otherwise on torch's fused scaled_dot_product_attention; the
N x N softmax of blocks.py:105-109 is never materialised; RoPE runs on the HIP
`curope.rope_2d` kernel when the tensors are on the GPU (pure-torch rotation otherwise, pos_embed.py:112-158);
GEMMs are hipBLASLt (bf16/fp16 under autocast, fp32 otherwise).  Heads run in fp32 like the reference
(model.py:205).  Inference only (the frontend never trains).
"""
from __future__ import annotations

from functools import partial
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

inf = float("inf")


# ------------------------------------------------------------------------------------------ RoPE 2D
class RoPE2D(nn.Module):
    def __init__(self, freq=100.0, F0=1.0):
        super().__init__()
        self.base, self.F0 = freq, F0

    def forward(self, tokens, positions):
        """tokens [B,heads,N,D] (any strides with (N? ,D) dense per head as produced below), positions [B,N,2]."""
        if tokens.is_cuda and tokens.dtype in (torch.float32, torch.float16, torch.bfloat16):
            view = tokens.transpose(1, 2)  # [B,N,H,D]
            if view.stride(3) == 1 and view.stride(2) == view.shape[3]:
                import curope  # drop-in HIP kernel (artdeco_amd/dropin/curope.py); trigonometry cached per positions tensor
                curope.rope_2d_cached(view, positions.contiguous(), self.base, self.F0)
                return tokens
        return self._torch(tokens, positions)

    def rotate_qk_inplace(self, qkv5, positions) -> bool:
        """qkv5 [B,N,3,H,D] contiguous: rotate the q and k slabs in place in one kernel launch (GPU only)."""
        if not (qkv5.is_cuda and qkv5.is_contiguous() and qkv5.dtype in (torch.float32, torch.float16, torch.bfloat16)):
            return False
        B, N, _, H, D = qkv5.shape
        import curope
        qk = qkv5.as_strided((B, N, 2 * H, D), (N * 3 * H * D, 3 * H * D, D, 1))
        curope.rope_2d_cached(qk, positions.contiguous(), self.base, self.F0)
        return True

    def _torch(self, tokens, positions):
        B, Hh, N, D = tokens.shape
        Q = D // 4
        inv = self.F0 / (self.base ** (torch.arange(Q, device=tokens.device, dtype=torch.float32) / Q))
        out = torch.empty_like(tokens)
        for xh in range(2):
            ang = positions[:, :, xh].float()[:, None, :, None] * inv  # [B,1,N,Q]
            c, s = ang.cos().to(tokens.dtype), ang.sin().to(tokens.dtype)
            u = tokens[..., xh * 2 * Q: xh * 2 * Q + Q]
            v = tokens[..., xh * 2 * Q + Q: xh * 2 * Q + 2 * Q]
            out[..., xh * 2 * Q: xh * 2 * Q + Q] = u * c - v * s
            out[..., xh * 2 * Q + Q: xh * 2 * Q + 2 * Q] = v * c + u * s
        return out


# ------------------------------------------------------------------------------------------ blocks
def _attend(q, k, v):
    """softmax(q k^T / sqrt(D)) v, returned as [B, N, H*D] (blocks.py:105-109).  fp16 GPU tensors with 64-wide heads (the
    TF32-class inference mode, every block of the released model) run on the hand-written gfx950 kernel; anything else
    (CPU, fp32 / bf16 modes, other head sizes) on torch's scaled_dot_product_attention."""
    if q.is_cuda and q.dtype == torch.float16:
        from artdeco_amd import attention as _att
        if _att.supported(q, k, v):
            return _att.attention(q, k, v)
    x = F.scaled_dot_product_attention(q, k, v)
    B, H, N, D = x.shape
    return x.transpose(1, 2).reshape(B, N, H * D)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x.to(self.fc1.weight.dtype))))


class Attention(nn.Module):
    def __init__(self, dim, rope, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)
        self.rope = rope

    def forward(self, x, xpos):
        B, N, C = x.shape
        qkv5 = self.qkv(x.to(self.qkv.weight.dtype)).reshape(B, N, 3, self.num_heads, C // self.num_heads)
        # q and k are adjacent [H,D] slabs of every token: rotate both with ONE in-place rope launch over 2H "heads"
        if not self.rope.rotate_qk_inplace(qkv5, xpos):
            qkv = qkv5.transpose(1, 3)
            q, k, v = self.rope(qkv[:, :, 0], xpos), self.rope(qkv[:, :, 1], xpos), qkv[:, :, 2]
        else:
            qkv = qkv5.transpose(1, 3)  # [B,H,3,N,D]
            q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
        return self.proj(_attend(q, k, v))


class CrossAttention(nn.Module):
    def __init__(self, dim, rope, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.projq = nn.Linear(dim, dim, bias=True)
        self.projk = nn.Linear(dim, dim, bias=True)
        self.projv = nn.Linear(dim, dim, bias=True)
        self.proj = nn.Linear(dim, dim)
        self.rope = rope

    def _stacked_kv(self):
        """[projk; projv] weight and bias, rebuilt whenever either parameter was replaced, cast or edited in place (inference only)."""
        ps = (self.projk.weight, self.projk.bias, self.projv.weight, self.projv.bias)
        key = tuple((p.data_ptr(), -1 if p.is_inference() else p._version, p.dtype, p.device) for p in ps)
        if getattr(self, "_kv_key", None) != key:
            with torch.no_grad():
                self._kv = (torch.cat([ps[0], ps[2]], 0).contiguous(), torch.cat([ps[1], ps[3]], 0).contiguous())
            self._kv_key = key
        return self._kv

    def forward(self, query, key, value, qpos, kpos):
        B, Nq, C = query.shape
        H, D = self.num_heads, C // self.num_heads
        wd = self.projq.weight.dtype
        query, key, value = query.to(wd), key.to(wd), value.to(wd)
        q = self.projq(query).reshape(B, Nq, H, D).permute(0, 2, 1, 3)
        if key is value and not torch.is_grad_enabled() and key.is_cuda:
            # the decoder passes the same normalised memory as key and value (blocks.py:188): ONE GEMM against the stacked
            # projk / projv weights (at 768 tokens every launch costs ~12 us whatever its size), k and v as views of its output
            kv = F.linear(key, *self._stacked_kv()).reshape(B, key.shape[1], 2, H, D)
            k, v = kv[:, :, 0].permute(0, 2, 1, 3), kv[:, :, 1].permute(0, 2, 1, 3)
        else:
            k = self.projk(key).reshape(B, key.shape[1], H, D).permute(0, 2, 1, 3)
            v = self.projv(value).reshape(B, value.shape[1], H, D).permute(0, 2, 1, 3)
        q, k = self.rope(q, qpos), self.rope(k, kpos)
        return self.proj(_attend(q, k, v))


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, norm_layer, rope):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, rope, num_heads)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x, xpos):
        x = x + self.attn(self.norm1(x), xpos)
        return x + self.mlp(self.norm2(x))


class DecoderBlock(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, norm_layer, rope):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, rope, num_heads)
        self.cross_attn = CrossAttention(dim, rope, num_heads)
        self.norm2 = norm_layer(dim)
        self.norm3 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.norm_y = norm_layer(dim)

    def forward(self, x, y, xpos, ypos):
        x = x + self.attn(self.norm1(x), xpos)
        y_ = self.norm_y(y)
        x = x + self.cross_attn(self.norm2(x), y_, y_, xpos, ypos)
        return x + self.mlp(self.norm3(x)), y

    def fused_tail(self, x, x_n1, mem, xpos, ypos):
        """The block from its first sub-layer on, given x (fp32, complete), x_n1 = norm1(x) and mem = norm_y(y) as fp16
        operands: returns (x after the cross-attention residual, the MLP output still to be added) -- the caller folds that
        last add into the next LayerNorm launch (artdeco_amd/fused_norm.py)."""
        from artdeco_amd.fused_norm import add_layernorm
        x, x_n2 = add_layernorm(x, self.attn(x_n1, xpos), self.norm2)
        x, x_n3 = add_layernorm(x, self.cross_attn(x_n2, mem, mem, xpos, ypos), self.norm3)
        return x, self.mlp(x_n3)


class PatchEmbed(nn.Module):
    def __init__(self, patch_size, embed_dim):
        super().__init__()
        self.patch_size = (patch_size, patch_size)
        self.proj = nn.Conv2d(3, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.Identity()

    def forward(self, x, true_shape=None):
        # A stride-P PxP convolution IS a GEMM over non-overlapping patches: [B h w, 3 P P] x [3 P P, D].  As a Conv2d in
        # fp16 MIOpen's immediate mode fell back to `naive_conv_ab_nonpacked_fwd_nchw_half_double_half`, 1.07 ms per image
        # (profiles/r02_frontend_kernel_stats_tf32eq.csv); as a Linear it is one hipBLASLt call.  Same parameters
        # (proj.weight [D,3,P,P], proj.bias), same arithmetic.
        B, C, H, W = x.shape
        P = self.patch_size[0]
        h, w = H // P, W // P
        patches = x.reshape(B, C, h, P, w, P).permute(0, 2, 4, 1, 3, 5).reshape(B, h * w, C * P * P)
        tokens = F.linear(patches, self.proj.weight.reshape(self.proj.weight.shape[0], -1), self.proj.bias)
        ys, xs = torch.arange(h, device=x.device), torch.arange(w, device=x.device)
        pos = torch.cartesian_prod(ys, xs).view(1, h * w, 2).expand(B, -1, 2).clone()
        return tokens, pos


# ------------------------------------------------------------------------------------------ DPT head
class ResidualConvUnit(nn.Module):
    def __init__(self, features):
        super().__init__()
        self.conv1 = nn.Conv2d(features, features, 3, 1, 1, bias=True)
        self.conv2 = nn.Conv2d(features, features, 3, 1, 1, bias=True)

    def forward(self, x):
        out = self.conv1(F.relu(x))
        out = self.conv2(F.relu(out))
        return out + x


class FeatureFusionBlock(nn.Module):
    def __init__(self, features):
        super().__init__()
        self.out_conv = nn.Conv2d(features, features, 1, 1, 0, bias=True)
        self.resConfUnit1 = ResidualConvUnit(features)
        self.resConfUnit2 = ResidualConvUnit(features)

    def forward(self, *xs):
        output = xs[0]
        if len(xs) == 2:
            output = output + self.resConfUnit1(xs[1])
        output = self.resConfUnit2(output)
        output = F.interpolate(output, scale_factor=2, mode="bilinear", align_corners=True)
        return self.out_conv(output)


class Interpolate(nn.Module):
    def forward(self, x):
        return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)


class _Scratch(nn.Module):
    pass


class DPTOutputAdapter(nn.Module):
    """dpt_block.py:DPTOutputAdapter + the dust3r forward (dpt_head.py:31-57), regression head."""

    def __init__(self, num_channels, hooks, dim_tokens, layer_dims=(96, 192, 384, 768), feature_dim=256, last_dim=128,
                 patch_size=16):
        super().__init__()
        self.hooks, self.P = list(hooks), patch_size
        sc = _Scratch()
        sc.layer1_rn = nn.Conv2d(layer_dims[0], feature_dim, 3, 1, 1, bias=False)
        sc.layer2_rn = nn.Conv2d(layer_dims[1], feature_dim, 3, 1, 1, bias=False)
        sc.layer3_rn = nn.Conv2d(layer_dims[2], feature_dim, 3, 1, 1, bias=False)
        sc.layer4_rn = nn.Conv2d(layer_dims[3], feature_dim, 3, 1, 1, bias=False)
        sc.layer_rn = nn.ModuleList([sc.layer1_rn, sc.layer2_rn, sc.layer3_rn, sc.layer4_rn])
        sc.refinenet1, sc.refinenet2 = FeatureFusionBlock(feature_dim), FeatureFusionBlock(feature_dim)
        sc.refinenet3, sc.refinenet4 = FeatureFusionBlock(feature_dim), FeatureFusionBlock(feature_dim)
        self.scratch = sc
        self.head = nn.Sequential(nn.Conv2d(feature_dim, feature_dim // 2, 3, 1, 1), Interpolate(),
                                  nn.Conv2d(feature_dim // 2, last_dim, 3, 1, 1), nn.ReLU(True),
                                  nn.Conv2d(last_dim, num_channels, 1, 1, 0))
        d, L = dim_tokens, layer_dims
        self.act_postprocess = nn.ModuleList([
            nn.Sequential(nn.Conv2d(d[0], L[0], 1), nn.ConvTranspose2d(L[0], L[0], 4, 4, 0)),
            nn.Sequential(nn.Conv2d(d[1], L[1], 1), nn.ConvTranspose2d(L[1], L[1], 2, 2, 0)),
            nn.Sequential(nn.Conv2d(d[2], L[2], 1)),
            nn.Sequential(nn.Conv2d(d[3], L[3], 1), nn.Conv2d(L[3], L[3], 3, 2, 1)),
        ])

    def forward(self, tokens: List[torch.Tensor], image_size):
        H, W = image_size
        nh, nw = H // self.P, W // self.P
        wd = self.head[0].weight.dtype
        # The hook tokens are the RAW residual stream of decoder levels 6 and 9 (un-normalised; real checkpoints carry outlier
        # channels there): they are never narrowed.  The 1x1 convolution that reads them (act_postprocess[i][0], 1.7 GFLOP per
        # head) stays in the tokens' precision -- to_inference_dtype leaves it fp32 -- and only ITS output enters the fp16 part.
        layers = [tokens[h] for h in self.hooks]
        layers = [l.transpose(1, 2).reshape(l.shape[0], l.shape[2], nh, nw) for l in layers]
        first = [self.act_postprocess[i][0] for i in range(len(layers))]
        layers = [f(l.to(f.weight.dtype)).to(wd) for f, l in zip(first, layers)]
        layers = [self.act_postprocess[i][1:](l) for i, l in enumerate(layers)]
        layers = [self.scratch.layer_rn[i](l) for i, l in enumerate(layers)]
        p4 = self.scratch.refinenet4(layers[3])[:, :, :layers[2].shape[2], :layers[2].shape[3]]
        p3 = self.scratch.refinenet3(p4, layers[2])
        p2 = self.scratch.refinenet2(p3, layers[1])
        p1 = self.scratch.refinenet1(p2, layers[0])
        return self.head(p1)


def reg_dense_depth(xyz, mode):
    kind = mode[0]
    if kind == "linear":
        return xyz
    d = xyz.norm(dim=-1, keepdim=True)
    xyz = xyz / d.clip(min=1e-8)
    if kind == "square":
        return xyz * d.square()
    if kind == "exp":
        return xyz * torch.expm1(d)
    raise ValueError(f"bad mode {mode}")


def reg_dense_conf(x, mode):
    kind, vmin, vmax = mode
    if kind == "exp":
        return vmin + x.exp().clip(max=vmax - vmin)
    if kind == "sigmoid":
        return (vmax - vmin) * torch.sigmoid(x) + vmin
    raise ValueError(f"bad mode {mode}")


class CatMlpDptHead(nn.Module):
    """mast3r/catmlp_dpt_head.py:43-96: DPT for 3D points (+conf), MLP on [enc, dec] tokens for descriptors."""

    def __init__(self, net, local_feat_dim, has_conf):
        super().__init__()
        l2, ed, dd = net.dec_depth, net.enc_embed_dim, net.dec_embed_dim
        self.dpt = DPTOutputAdapter(3 + int(has_conf), [0, l2 * 2 // 4, l2 * 3 // 4, l2], [ed, dd, dd, dd],
                                    patch_size=net.patch_size)
        self.local_feat_dim, self.patch_size = local_feat_dim, net.patch_size
        self.two_confs, self.depth_mode, self.conf_mode = net.two_confs, net.depth_mode, net.conf_mode
        self.desc_conf_mode = net.desc_conf_mode
        idim = ed + dd
        self.head_local_features = Mlp(idim, int(4.0 * idim), (local_feat_dim + int(net.two_confs)) * self.patch_size ** 2)

    def forward(self, decout, img_shape):
        H, W = int(img_shape[0]), int(img_shape[1])
        pts3d = self.dpt(decout, (H, W)).float()
        cat = torch.cat([decout[0], decout[-1]], dim=-1)
        B = cat.shape[0]
        lf = self.head_local_features(cat).float().transpose(-1, -2).reshape(B, -1, H // self.patch_size, W // self.patch_size)
        lf = F.pixel_shuffle(lf, self.patch_size)
        fmap = torch.cat([pts3d, lf], dim=1).permute(0, 2, 3, 1)
        res = {"pts3d": reg_dense_depth(fmap[..., 0:3], self.depth_mode)}
        start = 3
        if self.conf_mode is not None:
            res["conf"] = reg_dense_conf(fmap[..., 3], self.conf_mode)
            start = 4
        desc = fmap[..., start:start + self.local_feat_dim]
        res["desc"] = desc / desc.norm(dim=-1, keepdim=True)
        if self.two_confs:
            res["desc_conf"] = reg_dense_conf(fmap[..., start + self.local_feat_dim], self.desc_conf_mode)
        else:
            res["desc_conf"] = res["conf"].clone()
        return res


# ------------------------------------------------------------------------------------------ model
class AsymmetricMASt3R(nn.Module):
    def __init__(self, img_size=(512, 512), patch_size=16, enc_embed_dim=1024, enc_depth=24, enc_num_heads=16,
                 dec_embed_dim=768, dec_depth=12, dec_num_heads=12, mlp_ratio=4, pos_embed="RoPE100",
                 output_mode="pts3d+desc24", head_type="catmlp+dpt", depth_mode=("exp", -inf, inf),
                 conf_mode=("exp", 1, inf), two_confs=True, desc_conf_mode=("exp", 0, inf), **_ignored):
        super().__init__()
        if not pos_embed.startswith("RoPE") or head_type != "catmlp+dpt" or not output_mode.startswith("pts3d+desc"):
            raise NotImplementedError("only the released MASt3R configuration family is implemented")
        norm = partial(nn.LayerNorm, eps=1e-6)
        self.patch_size, self.enc_embed_dim, self.dec_embed_dim, self.dec_depth = patch_size, enc_embed_dim, dec_embed_dim, dec_depth
        self.depth_mode, self.conf_mode, self.two_confs = depth_mode, conf_mode, two_confs
        self.desc_conf_mode = desc_conf_mode if desc_conf_mode is not None else conf_mode
        self.rope = RoPE2D(freq=float(pos_embed[len("RoPE"):]))
        self.patch_embed = PatchEmbed(patch_size, enc_embed_dim)
        self.enc_blocks = nn.ModuleList([Block(enc_embed_dim, enc_num_heads, mlp_ratio, norm, self.rope) for _ in range(enc_depth)])
        self.enc_norm = norm(enc_embed_dim)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, dec_embed_dim))
        self.decoder_embed = nn.Linear(enc_embed_dim, dec_embed_dim, bias=True)
        self.dec_blocks = nn.ModuleList([DecoderBlock(dec_embed_dim, dec_num_heads, mlp_ratio, norm, self.rope) for _ in range(dec_depth)])
        self.dec_blocks2 = nn.ModuleList([DecoderBlock(dec_embed_dim, dec_num_heads, mlp_ratio, norm, self.rope) for _ in range(dec_depth)])
        self.dec_norm = norm(dec_embed_dim)
        lfd = int(output_mode[len("pts3d+desc"):])
        self.downstream_head1 = CatMlpDptHead(self, lfd, bool(conf_mode))
        self.downstream_head2 = CatMlpDptHead(self, lfd, bool(conf_mode))

    def load_state_dict(self, ckpt, **kw):
        ckpt = dict(ckpt)
        if not any(k.startswith("dec_blocks2") for k in ckpt):  # dust3r/model.py:108-115
            for k, v in list(ckpt.items()):
                if k.startswith("dec_blocks"):
                    ckpt[k.replace("dec_blocks", "dec_blocks2")] = v
        ckpt = {k: v for k, v in ckpt.items() if not k.startswith("prediction_head")}
        return super().load_state_dict(ckpt, **kw)

    def _fused_norms(self, x) -> bool:
        """TF32-class mode on the GPU: residual add + LayerNorm + operand cast run as one HIP launch per sub-layer
        (ADK_MAST3R_FUSED_NORM=0 keeps the three torch kernels, for A/B measurements and the equivalence test)."""
        import os
        C = x.shape[-1]   # what adk_add_layernorm takes (fused_norm.supported); anything else keeps the torch kernels
        return bool(getattr(self, "_fp32_stream", False) and x.is_cuda and self.patch_embed.proj.weight.dtype == torch.float16
                    and C % 4 == 0 and C <= 2048 and os.environ.get("ADK_MAST3R_FUSED_NORM", "1") != "0")

    def _encode_image(self, image, true_shape=None):
        x, pos = self.patch_embed(image.to(self.patch_embed.proj.weight.dtype), true_shape)
        if getattr(self, "_fp32_stream", False):
            x = x.float()   # the residual stream, the LayerNorms and the softmax stay fp32; only GEMM operands are narrow
        if self._fused_norms(x):
            from artdeco_amd.fused_norm import add_layernorm
            x = x.contiguous()
            pend = None  # the previous sub-layer's output, added inside the next LayerNorm launch
            for blk in self.enc_blocks:
                x, xn = add_layernorm(x, pend, blk.norm1)
                x, xn = add_layernorm(x, blk.attn(xn, pos), blk.norm2)
                pend = blk.mlp(xn)
            _, feat = add_layernorm(x, pend, self.enc_norm, out_f16=False)
            return feat, pos, None
        for blk in self.enc_blocks:
            x = blk(x, pos)
        return self.enc_norm(x), pos, None

    def _decoder(self, f1, pos1, f2, pos2):
        final = [(f1, f2)]
        wd = self.decoder_embed.weight.dtype
        f1, f2 = self.decoder_embed(f1.to(wd)), self.decoder_embed(f2.to(wd))
        if getattr(self, "_fp32_stream", False):
            f1, f2 = f1.float(), f2.float()
        final.append((f1, f2))
        side = self._side_stream(f1)
        if self._fused_norms(f1):
            return self._decoder_fused(final, f1.contiguous(), pos1, f2.contiguous(), pos2, side)
        for blk1, blk2 in zip(self.dec_blocks, self.dec_blocks2):
            a, b = final[-1]
            if side is None:
                n1, _ = blk1(a, b, pos1, pos2)
                n2, _ = blk2(b, a, pos2, pos1)
            else:
                # the two branches of a decoder level are independent (dust3r/model.py:181-186) and each is a chain of small
                # launches at 768 tokens that leaves most of the chip idle: run them on two HIP streams
                main = torch.cuda.current_stream(a.device)
                side.wait_stream(main)
                n1, _ = blk1(a, b, pos1, pos2)
                with torch.cuda.stream(side):
                    n2, _ = blk2(b, a, pos2, pos1)
                main.wait_stream(side)
                n2.record_stream(main)
            final.append((n1, n2))
        del final[1]
        final[-1] = (self.dec_norm(final[-1][0]), self.dec_norm(final[-1][1]))
        return zip(*final)

    def _decoder_fused(self, final, a, pos1, b, pos2, side):
        """_decoder's loop with every residual add folded into the LayerNorm launch that follows it.  (a, pa) / (b, pb): the
        two branches' streams and the MLP outputs still to be added to them; a level's complete outputs exist from the next
        level's first launch on, which is also where `final` collects them for the heads' hooks."""
        from artdeco_amd.fused_norm import add_layernorm
        pa = pb = None
        main = torch.cuda.current_stream(a.device)
        for lvl, (blk1, blk2) in enumerate(zip(self.dec_blocks, self.dec_blocks2)):
            a, a_n1 = add_layernorm(a, pa, blk1.norm1)
            b, b_n1 = add_layernorm(b, pb, blk2.norm1)
            if lvl > 0:
                final.append((a, b))
            mem1 = add_layernorm(b, None, blk1.norm_y)[1]  # branch 1 attends to view 2's tokens
            mem2 = add_layernorm(a, None, blk2.norm_y)[1]
            if side is None:
                a, pa = blk1.fused_tail(a, a_n1, mem1, pos1, pos2)
                b, pb = blk2.fused_tail(b, b_n1, mem2, pos2, pos1)
            else:
                side.wait_stream(main)
                a, pa = blk1.fused_tail(a, a_n1, mem1, pos1, pos2)
                with torch.cuda.stream(side):
                    b, pb = blk2.fused_tail(b, b_n1, mem2, pos2, pos1)
                main.wait_stream(side)
                b.record_stream(main)
                pb.record_stream(main)
        o1 = add_layernorm(a, pa, self.dec_norm, out_f16=False)[1]
        o2 = add_layernorm(b, pb, self.dec_norm, out_f16=False)[1]
        final.append((o1, o2))
        del final[1]
        return zip(*final)

    def _side_stream(self, t):
        """Second HIP stream for the independent halves of the decoder / the two heads (GPU only; ADK_MAST3R_STREAMS=0 disables)."""
        import os
        if not t.is_cuda or os.environ.get("ADK_MAST3R_STREAMS", "1") == "0":
            return None
        st = getattr(self, "_side", None)
        if st is None or st.device != t.device:
            st = self._side = torch.cuda.Stream(device=t.device)
        return st

    def _downstream_head(self, head_num, decout, img_shape):
        head = self.downstream_head1 if head_num == 1 else self.downstream_head2
        shp = img_shape[0] if torch.is_tensor(img_shape) and img_shape.dim() == 2 else img_shape
        return head(decout, shp)

    def both_heads(self, dec1, dec2, shape1, shape2):
        """head 1 on view 1's decoder tokens, head 2 on view 2's (dust3r/model.py:205-208): independent, so on the GPU they run
        on two HIP streams."""
        dec1, dec2 = [t.float() for t in dec1], [t.float() for t in dec2]
        side = self._side_stream(dec1[0])
        if side is None:
            return self._downstream_head(1, dec1, shape1), self._downstream_head(2, dec2, shape2)
        main = torch.cuda.current_stream(dec1[0].device)
        side.wait_stream(main)
        res1 = self._downstream_head(1, dec1, shape1)
        with torch.cuda.stream(side):
            res2 = self._downstream_head(2, dec2, shape2)
        main.wait_stream(side)
        for v in res2.values():
            v.record_stream(main)
        return res1, res2

    def to_inference_dtype(self, dtype, fp32_stream=False, heads=False):
        """Cast encoder + decoders ONCE to bf16/fp16 (autocast would re-cast every weight on every call);
        the heads stay fp32 like the reference (dust3r/model.py:205).

        fp32_stream=True is the TF32-CLASS mode: only the GEMM operands (Linear / patch-embedding weights and the
        activations entering them) are narrowed -- fp16 has TF32's 10-bit mantissa -- while products accumulate in
        fp32 on the matrix cores and the residual stream, every LayerNorm and the attention softmax stay fp32.  The
        reference runs the model under `torch.backends.cuda.matmul.allow_tf32 = True` (run_system.py:73), i.e. with
        10-bit GEMM operands; tools/frontend_precision.py measures this mode against an emulated TF32 forward."""
        trunk = (self.patch_embed, self.enc_blocks, self.enc_norm, self.decoder_embed, self.dec_blocks, self.dec_blocks2, self.dec_norm)
        if not fp32_stream:
            for m in trunk:
                m.to(dtype)
            self._trunk_dtype, self._fp32_stream = dtype, False   # a narrowed trunk has no fp32 residual stream to keep
            return self
        for root in trunk:
            for m in root.modules():
                if isinstance(m, (nn.Linear, nn.Conv2d)):
                    m.to(dtype)
        self._trunk_dtype, self._fp32_stream = None, True
        if heads:
            # the reference's heads run "in fp32" (dust3r/model.py:205) -- under the same allow_tf32 setting, which applies to
            # their convolutions and Linear layers as well (cudnn.allow_tf32 defaults to True): the same operand narrowing
            for head in (self.downstream_head1, self.downstream_head2):
                raw_readers = {id(seq[0]) for seq in head.dpt.act_postprocess}   # read the un-normalised residual stream: stay fp32
                for m in head.modules():
                    if isinstance(m, (nn.Linear, nn.Conv2d, nn.ConvTranspose2d)) and id(m) not in raw_readers:
                        m.to(dtype)
                # the DPT adapter's inputs are token-major [B, N, C] activations viewed as [B, C, h, w], i.e. ALREADY channels-last
                # in memory: with channels-last weights MIOpen picks NHWC kernels and the layout transposes around every
                # convolution disappear (tracked frame 8.09 -> 7.71 ms)
                head.dpt.to(memory_format=torch.channels_last)
        return self

    @staticmethod
    def outputs_finite(*results) -> torch.Tensor:
        """0-dim bool tensor on the device (no host sync): every value of the given head outputs is finite.  The narrowed modes
        keep fp16 GEMM outputs inside the trunk (|x| <= 65504); a caller that runs real checkpoints reads this flag once per
        batch of frames and re-runs the offending frame with to_inference_dtype(torch.float32)."""
        ok = None
        for r in results:
            for v in r.values():
                f = torch.isfinite(v).all()
                ok = f if ok is None else ok & f
        return ok

    @torch.inference_mode()
    def forward(self, view1, view2):
        img1, img2 = view1["img"], view2["img"]
        td = getattr(self, "_trunk_dtype", None)
        if td is not None:
            img1, img2 = img1.to(td), img2.to(td)
        shape1 = view1.get("true_shape", torch.tensor(img1.shape[-2:])[None])
        shape2 = view2.get("true_shape", torch.tensor(img2.shape[-2:])[None])
        feat1, pos1, _ = self._encode_image(img1, shape1)
        feat2, pos2, _ = self._encode_image(img2, shape2)
        dec1, dec2 = self._decoder(feat1, pos1, feat2, pos2)
        with torch.autocast(device_type=img1.device.type, enabled=False):
            res1, res2 = self.both_heads(dec1, dec2, shape1, shape2)
        res2["pts3d_in_other_view"] = res2.pop("pts3d")
        return res1, res2


def vit_large(**kw):
    """The released checkpoint's hyper-parameters (thirdparty/mast3r/README.md:277)."""
    return AsymmetricMASt3R(img_size=(512, 512), enc_embed_dim=1024, enc_depth=24, enc_num_heads=16, dec_embed_dim=768,
                            dec_depth=12, dec_num_heads=12, two_confs=True, desc_conf_mode=("exp", 0, inf), **kw)
