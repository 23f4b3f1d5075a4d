"""ORACLE (test infrastructure, not product code).

CPU fp32 restatement of the reference 1-D U-Net forward pass, written as pure
functions over a ``{state_dict key: tensor}`` mapping.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; the product path (``audiodiffuser_amd``) never does.

Parity status: PINNED against the reference itself, imported on CPU in the build
container by ``oracle/gen_golden.py`` (fixtures under ``tests/golden/``).  The
reference's own test-suite holds no vectors for this path.

Every function cites the reference lines it restates
(paths relative to the reference repo root).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from audiodiffuser_amd.config import UNet1dConfig

P = Dict[str, torch.Tensor]


# ---------------------------------------------------------------- small pieces
def time_embedding(p: P, t: torch.Tensor) -> torch.Tensor:
    """src/models/backbones/unet1d.py:128-148 (learned Fourier features + Linear),
    :678-684 (SiLU + Linear).  t: [B] -> [B, 4*channels]."""
    w = p["unet.to_time.0.0.weights"]
    tt = t[:, None]
    ang = tt * w[None, :] * 2 * math.pi
    feat = torch.cat((tt, ang.sin(), ang.cos()), dim=-1)
    h = F.linear(feat, p["unet.to_time.0.1.weight"], p["unet.to_time.0.1.bias"])
    return F.linear(F.silu(h), p["unet.to_time.2.weight"], p["unet.to_time.2.bias"])


def conv_block(p: P, pre: str, x: torch.Tensor, groups: int,
               scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None) -> torch.Tensor:
    """unet1d.py:193-207: GroupNorm -> optional x*(scale+1)+shift (:160-161) -> SiLU -> Conv1d k=3 p=1."""
    h = F.group_norm(x, groups, p[f"{pre}.groupnorm.weight"], p[f"{pre}.groupnorm.bias"], eps=1e-5)
    if scale is not None:
        h = h * (scale + 1) + shift
    return F.conv1d(F.silu(h), p[f"{pre}.project.weight"], p[f"{pre}.project.bias"], padding=1)


def label_embedding(p: P, classes: torch.Tensor, cond_drop_prob: float) -> torch.Tensor:
    """src/models/backbones/conditioner.py:92-111 with prob_mask_like (operator_utils.py:46-52) at its two
    deterministic settings: cond_drop_prob 0 keeps every label, 1 replaces every label by the null embedding.
    classes: int64 [B] -> [B, 4*channels]."""
    emb = F.embedding(classes, p["label_conditioner.label_emb.weight"])
    if cond_drop_prob > 0:
        if cond_drop_prob != 1:
            raise ValueError("the oracle covers cond_drop_prob 0 and 1 (inference); other values draw a random mask")
        emb = p["label_conditioner.null_classes_emb"].expand_as(emb)
    c = emb.shape[-1]
    h = F.layer_norm(emb, (c,), p["label_conditioner.class_to_cond.0.weight"], p["label_conditioner.class_to_cond.0.bias"], 1e-5)
    h = F.linear(h, p["label_conditioner.class_to_cond.1.weight"], p["label_conditioner.class_to_cond.1.bias"])
    return F.linear(F.silu(h), p["label_conditioner.class_to_cond.3.weight"], p["label_conditioner.class_to_cond.3.bias"])


def resnet_block(p: P, pre: str, x: torch.Tensor, temb: torch.Tensor, groups: int) -> torch.Tensor:
    """unet1d.py:297-316.  FiLM (scale, shift) = chunk(Linear(SiLU(cat(time_embed, class_embed)))) feeds block2 only
    (the caller passes the concatenation as ``temb``)."""
    cond = F.linear(F.silu(temb), p[f"{pre}.to_cond_embedding.1.weight"], p[f"{pre}.to_cond_embedding.1.bias"])
    scale, shift = cond[:, :, None].chunk(2, dim=1)
    h = conv_block(p, f"{pre}.block1", x, groups)
    h = conv_block(p, f"{pre}.block2", h, groups, scale, shift)
    key = f"{pre}.to_out.weight"
    res = F.conv1d(x, p[key], p[f"{pre}.to_out.bias"]) if key in p else x
    return h + res


def channel_layer_norm(x: torch.Tensor, g: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """unet1d.py:31-43 (LayerNorm1d over dim=1, biased variance, gain only)."""
    var = x.var(dim=1, unbiased=False, keepdim=True)
    mean = x.mean(dim=1, keepdim=True)
    return (x - mean) * (var + eps).rsqrt() * g


def self_attention(p: P, pre: str, x: torch.Tensor, heads: int) -> torch.Tensor:
    """src/models/backbones/attention_utils.py:113-184, plain self-attention branch
    (no context, no RoPE, no mask).  x: [B, N, C]."""
    b, n, c = x.shape
    d = c // heads
    q = F.linear(x, p[f"{pre}.to_q.weight"])
    k, v = F.linear(x, p[f"{pre}.to_kv.weight"]).chunk(2, dim=-1)
    q, k, v = (z.reshape(b, n, heads, d).permute(0, 2, 1, 3) for z in (q, k, v))
    sim = torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5)
    attn = sim.softmax(dim=-1, dtype=torch.float32)
    o = torch.matmul(attn, v).permute(0, 2, 1, 3).reshape(b, n, c)
    return F.linear(o, p[f"{pre}.to_out.weight"])


def transformer_block(p: P, pre: str, x: torch.Tensor, heads: int) -> torch.Tensor:
    """unet1d.py:106-122 with FeedForward1d :49-61."""
    c = x.shape[1]
    y = x.transpose(1, 2)
    y = self_attention(p, f"{pre}.attention", F.layer_norm(y, (c,), p[f"{pre}.norm.weight"], p[f"{pre}.norm.bias"], 1e-5), heads) + y
    x = y.transpose(1, 2)
    h = channel_layer_norm(x, p[f"{pre}.feed_forward.0.g"])
    h = F.conv1d(h, p[f"{pre}.feed_forward.1.weight"])
    h = F.gelu(h)
    h = channel_layer_norm(h, p[f"{pre}.feed_forward.3.g"])
    h = F.conv1d(h, p[f"{pre}.feed_forward.4.weight"])
    return h + x


def downsample_conv(p: P, pre: str, x: torch.Tensor, factor: int, kmult: int) -> torch.Tensor:
    """unet1d.py:214-225: Conv1d(k = factor*kmult+1, stride = factor, pad = factor*(kmult//2))."""
    return F.conv1d(x, p[f"{pre}.weight"], p[f"{pre}.bias"], stride=factor, padding=factor * (kmult // 2))


def upsample_conv(p: P, pre: str, x: torch.Tensor, factor: int) -> torch.Tensor:
    """unet1d.py:248-255: ConvTranspose1d(k = 2f, stride f, pad f//2 + f%2, output_padding f%2)."""
    return F.conv_transpose1d(x, p[f"{pre}.weight"], p[f"{pre}.bias"], stride=factor,
                              padding=factor // 2 + factor % 2, output_padding=factor % 2)


# ---------------------------------------------------------------- whole network
def unet1d_forward(p: P, cfg: UNet1dConfig, x: torch.Tensor, t: torch.Tensor,
                   taps: Optional[Dict[str, torch.Tensor]] = None, classes: Optional[torch.Tensor] = None,
                   cond_drop_prob: float = 0.0) -> torch.Tensor:
    """unet1d.py:864-893 -> :771-816 (no text context; ``classes`` = int64 labels for a class-conditional net).

    x: [B, in_channels, L], t: [B] (= c_noise).  ``taps`` optionally records
    intermediate activations by name (used by kernel-level parity tests)."""
    g, heads = cfg.resnet_groups, cfg.attention_heads
    n = cfg.num_layers
    pad = cfg.window_length // 2 - cfg.stride // 2

    def rec(name, v):
        if taps is not None:
            taps[name] = v
        return v

    h = rec("to_in", F.conv1d(x, p["unet.to_in.to_in.weight"], stride=cfg.stride, padding=pad))  # :584-591
    temb = rec("temb", time_embedding(p, t))
    if classes is not None:                                                       # :877, resblocks :306-308
        temb = torch.cat((temb, rec("class_emb", label_embedding(p, classes, cond_drop_prob))), dim=-1)
    skips_list: List[List[torch.Tensor]] = []
    for i in range(n):                                                            # :792-801, :441-468
        pre = f"unet.downsamples.{i}"
        h = rec(f"down{i}.conv", downsample_conv(p, f"{pre}.downsample", h, cfg.factors[i], cfg.kernel_multiplier_downsample))
        skips = []
        for j in range(cfg.num_blocks[i]):
            h = rec(f"down{i}.block{j}", resnet_block(p, f"{pre}.blocks.{j}", h, temb, g))
            skips.append(h)
        if cfg.attentions[i]:
            h = rec(f"down{i}.attn", transformer_block(p, f"{pre}.transformer", h, heads))
            skips.append(h)
        skips_list.append(skips)
    h = rec("mid.pre", resnet_block(p, "unet.bottleneck.pre_block", h, temb, g))    # :374-379
    if cfg.use_attention_bottleneck:
        h = rec("mid.attn", transformer_block(p, "unet.bottleneck.transformer", h, heads))
    h = rec("mid.post", resnet_block(p, "unet.bottleneck.post_block", h, temb, g))
    skip_scale = 2 ** -0.5 if cfg.use_skip_scale else 1.0
    for u, i in enumerate(reversed(range(n))):                                    # :807-812, :542-566
        pre = f"unet.upsamples.{u}"
        skips = skips_list.pop()
        nb = cfg.num_blocks[i] + (1 if cfg.attentions[i] else 0)
        for j in range(nb):
            h = torch.cat([h, skips.pop() * skip_scale], dim=1)                   # :539-540
            h = rec(f"up{u}.block{j}", resnet_block(p, f"{pre}.blocks.{j}", h, temb, g))
        if cfg.attentions[i]:
            h = rec(f"up{u}.attn", transformer_block(p, f"{pre}.transformer", h, heads))
        h = rec(f"up{u}.conv", upsample_conv(p, f"{pre}.upsample", h, cfg.factors[i]))
    return F.conv_transpose1d(h, p["unet.to_out.to_out.weight"], stride=cfg.stride, padding=pad)  # :611-622
