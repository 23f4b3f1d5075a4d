"""ORACLE (test infrastructure, not product code) -- the ADM-style 2-D U-Net of BASELINE config 4
(SURVEY.md section 8f row 3), restated on CPU fp32 as pure functions over a ``{state_dict key: tensor}``
mapping.  See oracle/unet1d.py for the rules on who may import this package.

Parity status: PINNED against the reference itself (``src/models/backbones/unet2d_oai.py`` imported on CPU in
the build container by ``oracle/gen_golden_next.py``; fixtures in ``tests/golden/next_golden.npz``).  The HIP path
(audiodiffuser_amd/csrc/adf_conv2d.hip, ``adf_adm_create``) is held to these fixtures by tests/test_adm.py.

Every function cites the reference lines it restates (paths relative to the reference repo root; all line
numbers are in ``src/models/backbones/unet2d_oai.py`` unless another file is named).

Arithmetic modes as in oracle/unet1d.py.  ``storage="bf16"`` rounds where the HIP throughput mode of adf_conv2d.hip holds bf16:
every stored activation, the activated conv operand ``silu(a x + b)``, the GEMM weights; a conv with a residual stores
``r(r(conv + bias) + residual)`` (its output tile is rounded in LDS before the residual is added); GroupNorm statistics are those
of the stored tensor; the first conv (vector kernel, fp32 weights), the last conv (fp32 weights, fp32 activation, fp32 output),
the attention softmax and P V (vector kernel, fp32) and the embedding path are not rounded inside.  ``force`` / ``errs``: teacher
forcing per recorded tensor, as in oracle/unet1d.py.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from .unet1d import label_embedding, Storage, FP32, rel_l2, mfma_attention

P = Dict[str, torch.Tensor]
Spec = Tuple[Tuple[int, ...], str]


# configuration, module structure, state-dict layout and the weight generator live in the package (the plugin needs them too)
from audiodiffuser_amd.adm_config import (ADMConfig, config_c4, config_c4_small, structure, param_specs, generate_weights,  # noqa: E402,F401
                                          _Layer, _Structure)


# ------------------------------------------------------------------ pieces
def group_norm32(p: P, pre: str, x: torch.Tensor) -> torch.Tensor:
    """:10-21 -- GroupNorm(32, C), eps 1e-5, computed in fp32."""
    return F.group_norm(x.float(), 32, p[f"{pre}.weight"], p[f"{pre}.bias"], 1e-5)


def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    """:31-49 -- fixed sinusoidal features, cosines first."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def conv2d(p: P, pre: str, x: torch.Tensor, stride: int = 1, q: Storage = FP32, round_w: bool = True) -> torch.Tensor:
    w = p[f"{pre}.weight"]
    if q.bf16 and round_w:
        w = q.w(w)
    return F.conv2d(x, w, p[f"{pre}.bias"], stride=stride, padding=w.shape[-1] // 2)


def upsample(p: P, pre: str, x: torch.Tensor, use_conv: bool, q: Storage = FP32) -> torch.Tensor:
    """:122-127 -- nearest x2, then the 3x3 conv."""
    x = F.interpolate(x, scale_factor=2, mode="nearest")
    return q.r(conv2d(p, f"{pre}.conv", x, q=q)) if use_conv else x


def downsample(p: P, pre: str, x: torch.Tensor, use_conv: bool, q: Storage = FP32) -> torch.Tensor:
    """:146-158 -- 3x3 stride-2 conv (padding 1), or a 2x2 average pool."""
    return q.r(conv2d(p, f"{pre}.op", x, stride=2, q=q)) if use_conv else q.r(F.avg_pool2d(x, 2, 2))


def res_block(p: P, l: _Layer, x: torch.Tensor, emb: torch.Tensor, scale_shift: bool, q: Storage = FP32, rec=None) -> torch.Tensor:
    """:248-272.  ``emb`` is the 4*model_channels embedding; the block applies SiLU + Linear to it (:214-220).
    ``rec(suffix, tensor)`` records / forces the stored tensors of the device path (``.h1``, ``.skip``)."""
    pre = l.pre
    rec = rec or (lambda _n, v: v)
    if q.bf16:                              # rounding points of the device path (adf_net_adm.hip)
        h = F.silu(group_norm32(p, f"{pre}.in_layers.0", x))
        if l.down:                          # one pass writes avg_pool(in_rest(x)), another avg_pool(x): each rounds once, on its store
            h, x = q.r(F.avg_pool2d(h, 2, 2)), q.r(F.avg_pool2d(x, 2, 2))
        elif l.up:                          # the conv gathers the activated (rounded) operand through the x 2 index map; x is copied
            h, x = F.interpolate(q.r(h), scale_factor=2, mode="nearest"), F.interpolate(x, scale_factor=2, mode="nearest")
        else:
            h = q.r(h)
        e = F.linear(F.silu(emb), p[f"{pre}.emb_layers.1.weight"], p[f"{pre}.emb_layers.1.bias"])[:, :, None, None]
        if scale_shift:
            h = rec(".h1", q.r(conv2d(p, f"{pre}.in_layers.2", h, q=q)))
            scale, shift = torch.chunk(e, 2, dim=1)
            h2 = q.r(F.silu(group_norm32(p, f"{pre}.out_layers.0", h) * (1 + scale) + shift))
        else:                               # the device stores conv1 + emb_out (one rounding) and normalises that tensor
            h = rec(".h1", q.r(conv2d(p, f"{pre}.in_layers.2", h, q=q) + e))
            h2 = q.r(F.silu(group_norm32(p, f"{pre}.out_layers.0", h)))
        skip = rec(".skip", q.r(conv2d(p, f"{pre}.skip_connection", x, q=q))) if l.cin != l.cout else x
        return q.r(q.r(conv2d(p, f"{pre}.out_layers.3", h2, q=q)) + skip)
    h = F.silu(group_norm32(p, f"{pre}.in_layers.0", x))
    if l.up or l.down:                     # :249-254: resample the activated h and the raw x, then convolve
        if l.up:
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        else:
            h = F.avg_pool2d(h, 2, 2)
            x = F.avg_pool2d(x, 2, 2)
    h = rec(".h1", conv2d(p, f"{pre}.in_layers.2", h))
    e = F.linear(F.silu(emb), p[f"{pre}.emb_layers.1.weight"], p[f"{pre}.emb_layers.1.bias"])[:, :, None, None]
    if scale_shift:                        # :262-267
        scale, shift = torch.chunk(e, 2, dim=1)
        h = group_norm32(p, f"{pre}.out_layers.0", h) * (1 + scale) + shift
    else:                                  # :268-270
        h = group_norm32(p, f"{pre}.out_layers.0", h + e)
    h = conv2d(p, f"{pre}.out_layers.3", F.silu(h))       # dropout is the identity at inference
    skip = rec(".skip", conv2d(p, f"{pre}.skip_connection", x)) if l.cin != l.cout else x
    return skip + h


def qkv_attention(qkv: torch.Tensor, heads: int, legacy: bool, st: Optional["Storage"] = None) -> torch.Tensor:
    """:324-350 (legacy: heads split first, each head's rows are q|k|v) and :352-380 (new order: q|k|v split first).
    Both scale q and k by ch^-1/4 and take the softmax in fp32.  bf16 storage (``st``) on the shapes the device serves with its MFMA attention
    kernel: the un-normalised probabilities are rounded as the matrix operand of P V while the normaliser sums the unrounded ones (as
    oracle/unet1d.py self_attention)."""
    b, width, n = qkv.shape
    ch = width // (3 * heads)
    if legacy:
        q, k, v = qkv.reshape(b * heads, 3 * ch, n).split(ch, dim=1)
    else:
        q, k, v = (t.reshape(b * heads, ch, n) for t in qkv.chunk(3, dim=1))
    scale = 1 / math.sqrt(math.sqrt(ch))
    w = torch.einsum("bct,bcs->bts", q * scale, k * scale)
    if st is not None and st.bf16 and mfma_attention(ch, n):
        pr = torch.exp(w.float() - w.float().amax(dim=-1, keepdim=True))
        return (torch.einsum("bts,bcs->bct", st.r(pr), v) / pr.sum(dim=-1).unsqueeze(1)).reshape(b, -1, n)
    w = torch.softmax(w.float(), dim=-1)
    return torch.einsum("bts,bcs->bct", w, v).reshape(b, -1, n)


def attention_block(p: P, l: _Layer, x: torch.Tensor, heads: int, legacy: bool, q: Storage = FP32, rec=None) -> torch.Tensor:
    """:316-322.  The residual is added to the NORMALISED input (the reference reassigns ``x = self.norm(x)``).
    ``rec``: ``.xn``, ``.qkv`` (in the device's q | k | v row order), ``.att``."""
    b, c = x.shape[:2]
    rec = rec or (lambda _n, v: v)
    sp = x.shape
    xn = rec(".xn", q.r(group_norm32(p, f"{l.pre}.norm", x.reshape(b, c, -1))).reshape(sp)).reshape(b, c, -1)
    wq = q.w(p[f"{l.pre}.qkv.weight"]) if q.bf16 else p[f"{l.pre}.qkv.weight"]
    qkv = q.r(F.conv1d(xn, wq, p[f"{l.pre}.qkv.bias"]))
    if rec is not None:
        ch = c // heads
        if legacy:      # device rows: (which, head, c) <- reference rows (head, which, c)
            dev = qkv.reshape(b, heads, 3, ch, -1).permute(0, 2, 1, 3, 4).reshape(b, 3 * c, -1)
            dev = rec(".qkv", dev.reshape(b, 3 * c, *sp[2:])).reshape(b, 3, heads, ch, -1)
            qkv = dev.permute(0, 2, 1, 3, 4).reshape(b, 3 * c, -1)
        else:
            qkv = rec(".qkv", qkv.reshape(b, 3 * c, *sp[2:])).reshape(b, 3 * c, -1)
    h = rec(".att", q.r(qkv_attention(qkv, heads, legacy, q)).reshape(sp)).reshape(b, c, -1)
    wp = q.w(p[f"{l.pre}.proj_out.weight"]) if q.bf16 else p[f"{l.pre}.proj_out.weight"]
    h = q.r(F.conv1d(h, wp, p[f"{l.pre}.proj_out.bias"]))
    return q.r(xn + h).reshape(x.shape)


# ------------------------------------------------------------------ the network
def unet2d_forward(p: P, cfg: ADMConfig, x: torch.Tensor, t: torch.Tensor, classes: Optional[torch.Tensor] = None,
                   cond_drop_prob: float = 0.0, taps: Optional[Dict[str, torch.Tensor]] = None, storage: str = "fp32",
                   force: Optional[Dict[str, torch.Tensor]] = None, errs: Optional[Dict[str, float]] = None) -> torch.Tensor:
    """``UNetModel.forward`` :603-634.  x: [B, in_channels, H, W], t: [B] (the EDM wrapper passes c_noise).
    ``taps`` (optional) receives the output of every input / middle / output block under the block's module name, and under
    ``<block>.<j>`` (+ ``.h1 / .skip / .xn / .qkv / .att``) every tensor the device path stores.  ``force`` maps names to tensors
    shaped [B, C, H*W] (the device's tap copies) or [B, C, H, W]."""
    assert (classes is not None) == (cfg.num_classes is not None), "must specify y if and only if the model is class-conditional"
    s = structure(cfg)
    q = Storage(storage)

    def rec(name, v):
        if force is not None and name in force:
            f = force[name].reshape(v.shape)
            if errs is not None:
                errs[name] = rel_l2(v, f)
            v = f
        if taps is not None:
            taps[name] = v
        return v

    emb = F.linear(timestep_embedding(t, cfg.model_channels), p["time_embed.0.weight"], p["time_embed.0.bias"])
    emb = F.linear(F.silu(emb), p["time_embed.2.weight"], p["time_embed.2.bias"])
    if classes is not None:
        emb = emb + label_embedding(p, classes, cond_drop_prob)
    rec("emb", emb)

    def run(layers: List[_Layer], h: torch.Tensor) -> torch.Tensor:
        for l in layers:
            sub = lambda sfx, v, _pre=l.pre: rec(_pre + sfx, v)
            if l.kind == "conv":
                h = q.r(conv2d(p, l.pre, h, q=q, round_w=False))      # vector kernel on the device: fp32 weights
            elif l.kind == "res":
                h = res_block(p, l, h, emb, cfg.use_scale_shift_norm, q, sub)
            elif l.kind == "attn":
                h = attention_block(p, l, h, cfg.heads(l.cin), not cfg.use_new_attention_order, q, sub)
            elif l.kind == "down":
                h = downsample(p, l.pre, h, cfg.conv_resample, q)
            elif l.kind == "up":
                h = upsample(p, l.pre, h, cfg.conv_resample, q)
            h = rec(l.pre, h)
        return h

    hs = []
    h = x
    for i, blk in enumerate(s.input_blocks):
        h = rec(f"input_blocks.{i}", run(blk, h))
        hs.append(h)
    h = rec("middle_block", run(s.middle, h))
    for i, blk in enumerate(s.output_blocks):
        h = rec(f"output_blocks.{i}", run(blk, torch.cat([h, hs.pop()], dim=1)))
    h = F.silu(group_norm32(p, "out.0", h))
    return conv2d(p, "out.2", h, q=q, round_w=False)      # vector kernel on the device: fp32 activation and weights
