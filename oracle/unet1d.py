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

Two arithmetic modes.  ``storage="fp32"`` (default) is the reference's own arithmetic and the one pinned
against the imported reference.  ``storage="bf16"`` is the SAME restatement with a round-to-nearest-even
bf16 rounding inserted at exactly the points where the HIP throughput mode holds a value in bf16
("bf16-storage oracle"): every stored activation, every MFMA operand (the activated
``silu(a x + b)`` input of a conv, the GEMM weights, attention probabilities) -- accumulation, norms,
softmax statistics, the sigma embedding and the FiLM projections stay fp32, as on the device.  What is
left between this mode and the device is summation order inside the fp32 accumulations (and the rare
bf16 roundings that flip because of it), so the GPU tests can hold the bf16 kernels to a bound ~50x
tighter than bf16-vs-fp32.  ``force`` ("teacher forcing"): a mapping tap-name -> tensor; each recorded
activation is compared with the forced value (error into ``errs``) and REPLACED by it, so every layer is
checked in isolation on exactly the inputs the device layer saw.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from audiodiffuser_amd.config import UNet1dConfig

P = Dict[str, torch.Tensor]


class Storage:
    """Where a value is rounded.  ``fp32``: nowhere (the reference's arithmetic).  ``bf16``: ``r`` rounds a stored
    activation / MFMA operand to bf16 (round-to-nearest-even, like v_cvt_pk_bf16_f32) and ``w`` rounds a GEMM weight."""

    def __init__(self, kind: str = "fp32"):
        if kind not in ("fp32", "bf16"):
            raise ValueError("storage must be 'fp32' or 'bf16'")
        self.kind = kind
        self.bf16 = kind == "bf16"
        self._wcache: Dict[int, torch.Tensor] = {}

    def r(self, x: torch.Tensor) -> torch.Tensor:
        return x.to(torch.bfloat16).to(torch.float32) if self.bf16 else x

    def w(self, x: torch.Tensor) -> torch.Tensor:
        if not self.bf16:
            return x
        k = id(x)
        if k not in self._wcache:
            self._wcache[k] = x.to(torch.bfloat16).to(torch.float32)
        return self._wcache[k]


FP32 = Storage("fp32")


# ---------------------------------------------------------------- small pieces
def time_embedding(p: P, t: torch.Tensor) -> torch.Tensor:
    """src/models/backbones/unet1d.py:128-148 (learned Fourier features + Linear),
    :678-684 (SiLU + Linear).  t: [B] -> [B, 4*channels].  fp32 in both storage modes."""
    w = p["unet.to_time.0.0.weights"]
    tt = t[:, None]
    ang = tt * w[None, :] * 2 * math.pi
    feat = torch.cat((tt, ang.sin(), ang.cos()), dim=-1)
    h = F.linear(feat, p["unet.to_time.0.1.weight"], p["unet.to_time.0.1.bias"])
    return F.linear(F.silu(h), p["unet.to_time.2.weight"], p["unet.to_time.2.bias"])


def conv_block(p: P, pre: str, x: torch.Tensor, groups: int,
               scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None, q: Storage = FP32) -> torch.Tensor:
    """unet1d.py:193-207: GroupNorm -> optional x*(scale+1)+shift (:160-161) -> SiLU -> Conv1d k=3 p=1.
    bf16 storage: the activated tensor is the MFMA A operand (rounded), the weight the B operand (rounded); the
    result (bias included) is returned UNROUNDED -- the caller rounds once after adding the residual."""
    h = F.group_norm(x, groups, p[f"{pre}.groupnorm.weight"], p[f"{pre}.groupnorm.bias"], eps=1e-5)
    if scale is not None:
        h = h * (scale + 1) + shift
    return F.conv1d(q.r(F.silu(h)), q.w(p[f"{pre}.project.weight"]), p[f"{pre}.project.bias"], padding=1)


def label_embedding(p: P, classes: torch.Tensor, cond_drop_prob: float) -> torch.Tensor:
    """src/models/backbones/conditioner.py:92-111 with prob_mask_like (operator_utils.py:46-52) at its two
    deterministic settings: cond_drop_prob 0 keeps every label, 1 replaces every label by the null embedding.
    classes: int64 [B] -> [B, 4*channels].  fp32 in both storage modes."""
    emb = F.embedding(classes, p["label_conditioner.label_emb.weight"])
    if cond_drop_prob > 0:
        if cond_drop_prob != 1:
            raise ValueError("the oracle covers cond_drop_prob 0 and 1 (inference); other values draw a random mask")
        emb = p["label_conditioner.null_classes_emb"].expand_as(emb)
    c = emb.shape[-1]
    h = F.layer_norm(emb, (c,), p["label_conditioner.class_to_cond.0.weight"], p["label_conditioner.class_to_cond.0.bias"], 1e-5)
    h = F.linear(h, p["label_conditioner.class_to_cond.1.weight"], p["label_conditioner.class_to_cond.1.bias"])
    return F.linear(F.silu(h), p["label_conditioner.class_to_cond.3.weight"], p["label_conditioner.class_to_cond.3.bias"])


def resnet_block(p: P, pre: str, x: torch.Tensor, temb: torch.Tensor, groups: int, q: Storage = FP32,
                 x_raw: Optional[torch.Tensor] = None, rec_h1=None) -> torch.Tensor:
    """unet1d.py:297-316.  FiLM (scale, shift) = chunk(Linear(SiLU(cat(time_embed, class_embed)))) feeds block2 only
    (the caller passes the concatenation as ``temb``).  ``x_raw``: the input as the residual 1x1 conv reads it
    (bf16 storage rounds the scaled skip half of a concatenation there; the GroupNorm path folds the scale into
    its affine and never materialises it).  ``rec_h1``: callback recording (and possibly forcing) the stored intermediate
    h1 = conv1 output, the tap "<block>.h1" of the device's two-launch resblocks."""
    cond = F.linear(F.silu(temb), p[f"{pre}.to_cond_embedding.1.weight"], p[f"{pre}.to_cond_embedding.1.bias"])
    scale, shift = cond[:, :, None].chunk(2, dim=1)
    h = q.r(conv_block(p, f"{pre}.block1", x, groups, q=q))
    if rec_h1 is not None:
        h = rec_h1(h)
    h = conv_block(p, f"{pre}.block2", h, groups, scale, shift, q=q)
    key = f"{pre}.to_out.weight"
    xr = x if x_raw is None else x_raw
    # device: the 1x1 residual conv is a second K segment of the same fp32 accumulator (its bias joins conv2's)
    res = F.conv1d(xr, q.w(p[key]), p[f"{pre}.to_out.bias"]) if key in p else xr
    return q.r(h + res)


def channel_layer_norm(x: torch.Tensor, g: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """unet1d.py:31-43 (LayerNorm1d over dim=1, biased variance, gain only)."""
    var = x.var(dim=1, unbiased=False, keepdim=True)
    mean = x.mean(dim=1, keepdim=True)
    return (x - mean) * (var + eps).rsqrt() * g


def mfma_attention(d: int, n: int) -> bool:
    """The shapes whose probabilities launch_attention (adf_kernels.hip) feeds to the matrix cores as ONE bf16: head dim 32 up to 1024 tokens
    (head dim 64 goes through the same kernel with bf16 hi + lo probabilities, i.e. the fp32 arithmetic of this restatement)."""
    return d == 32 and n <= 1024


def self_attention(p: P, pre: str, x: torch.Tensor, heads: int, q: Storage = FP32, rec=None) -> torch.Tensor:
    """src/models/backbones/attention_utils.py:113-184, plain self-attention branch
    (no context, no RoPE, no mask).  x: [B, N, C].
    bf16 storage: q | k | v are stored tensors (rounded); scores and the softmax statistics are fp32; the
    un-normalised probabilities are rounded as the MFMA operand of P V while the normaliser sums the unrounded
    ones; the attention output is a stored tensor (rounded); the projection result is returned unrounded (the
    caller adds the residual in fp32 and rounds once)."""
    b, n, c = x.shape
    d = c // heads
    qq = q.r(F.linear(x, q.w(p[f"{pre}.to_q.weight"])))
    kv = q.r(F.linear(x, q.w(p[f"{pre}.to_kv.weight"])))
    if rec is not None:                       # the device stores q | k | v as one [B, N, 3C] tensor (tap "<block>.qkv")
        qkv = rec("qkv", torch.cat((qq, kv), dim=-1).transpose(1, 2)).transpose(1, 2)
        qq, kv = qkv[..., :c], qkv[..., c:]
    k, v = kv.chunk(2, dim=-1)
    qq, k, v = (z.reshape(b, n, heads, d).permute(0, 2, 1, 3) for z in (qq, k, v))
    sim = torch.matmul(qq, k.transpose(-1, -2)) * (d ** -0.5)
    if q.bf16:
        # the device's MFMA kernel at head dim 32 (<= 1024 tokens) feeds the probabilities to the matrix cores as one bf16; the other paths keep
        # (or, at head dim 64, reconstruct) their fp32 value
        pr = torch.exp(sim - sim.amax(dim=-1, keepdim=True))
        pv = q.r(pr) if mfma_attention(d, n) else pr
        o = torch.matmul(pv, v) / pr.sum(dim=-1, keepdim=True)
    else:
        attn = sim.softmax(dim=-1, dtype=torch.float32)
        o = torch.matmul(attn, v)
    o = q.r(o.permute(0, 2, 1, 3).reshape(b, n, c))
    if rec is not None:
        o = rec("att", o.transpose(1, 2)).transpose(1, 2)
    return F.linear(o, q.w(p[f"{pre}.to_out.weight"]))


def transformer_block(p: P, pre: str, x: torch.Tensor, heads: int, q: Storage = FP32, rec=None) -> torch.Tensor:
    """unet1d.py:106-122 with FeedForward1d :49-61.  bf16 storage rounds every tensor the device stores: both
    LayerNorm outputs, the attention residual sum, the GELU output, the block output.  ``rec(suffix, tensor [B, C, N])``
    records (and may force) the block's stored intermediates: ln, qkv, att, x1, n1, f1, n2 -- the taps "<block>.<suffix>" of
    the device's nine-launch path."""
    r_ = rec if rec is not None else (lambda name, v: v)
    c = x.shape[1]
    y = x.transpose(1, 2)
    ln = q.r(F.layer_norm(y, (c,), p[f"{pre}.norm.weight"], p[f"{pre}.norm.bias"], 1e-5))
    ln = r_("ln", ln.transpose(1, 2)).transpose(1, 2)
    y = q.r(self_attention(p, f"{pre}.attention", ln, heads, q, rec=rec) + y)
    x = r_("x1", y.transpose(1, 2))
    h = r_("n1", q.r(channel_layer_norm(x, p[f"{pre}.feed_forward.0.g"])))
    h = F.conv1d(h, q.w(p[f"{pre}.feed_forward.1.weight"]))
    h = r_("f1", q.r(F.gelu(h)))
    h = r_("n2", q.r(channel_layer_norm(h, p[f"{pre}.feed_forward.3.g"])))
    h = F.conv1d(h, q.w(p[f"{pre}.feed_forward.4.weight"]))
    return q.r(h + x)


def downsample_conv(p: P, pre: str, x: torch.Tensor, factor: int, kmult: int, q: Storage = FP32) -> torch.Tensor:
    """unet1d.py:214-225: Conv1d(k = factor*kmult+1, stride = factor, pad = factor*(kmult//2))."""
    return q.r(F.conv1d(x, q.w(p[f"{pre}.weight"]), p[f"{pre}.bias"], stride=factor, padding=factor * (kmult // 2)))


def upsample_conv(p: P, pre: str, x: torch.Tensor, factor: int, q: Storage = FP32, nearest: bool = False) -> torch.Tensor:
    """unet1d.py:248-255: ConvTranspose1d(k = 2f, stride f, pad f//2 + f%2, output_padding f%2); with ``nearest`` unet1d.py:236-246:
    nn.Upsample(scale_factor=f, mode="nearest") -> nn.ReflectionPad1d(1) -> Conv1d(k = 3, padding 0) (the copies round nothing)."""
    if nearest:
        u = F.pad(F.interpolate(x, scale_factor=factor, mode="nearest"), (1, 1), mode="reflect")
        return q.r(F.conv1d(u, q.w(p[f"{pre}.2.weight"]), p[f"{pre}.2.bias"]))
    return q.r(F.conv_transpose1d(x, q.w(p[f"{pre}.weight"]), p[f"{pre}.bias"], stride=factor,
                                  padding=factor // 2 + factor % 2, output_padding=factor % 2))


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    """||a - b||_2 / ||b||_2 in fp64: the metric of the bf16-storage comparisons (a max-norm metric would sit at one
    bf16 ulp = 2^-8 as soon as a single rounding flips)."""
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


# ---------------------------------------------------------------- whole network
def unet1d_forward(p: P, cfg: UNet1dConfig, x: torch.Tensor, t: torch.Tensor,
                   taps: Optional[Dict[str, torch.Tensor]] = None, classes: Optional[torch.Tensor] = None,
                   cond_drop_prob: float = 0.0, storage: str = "fp32",
                   force: Optional[Dict[str, torch.Tensor]] = None, errs: Optional[Dict[str, float]] = None) -> torch.Tensor:
    """unet1d.py:864-893 -> :771-816 (no text context; ``classes`` = int64 labels for a class-conditional net).

    x: [B, in_channels, L], t: [B] (= c_noise).  ``taps`` optionally records intermediate activations by name (used
    by kernel-level parity tests).  ``storage``: see the module docstring.  ``force`` / ``errs``: teacher forcing --
    every recorded activation whose name is in ``force`` is compared with it (relative L2 error into ``errs``) and
    replaced by it before the next layer runs."""
    q = Storage(storage)
    g, heads = cfg.resnet_groups, cfg.attention_heads
    n = cfg.num_layers
    pad = cfg.window_length // 2 - cfg.stride // 2

    def rec(name, v):
        if taps is not None:
            taps[name] = v
        if force is not None and name in force:
            if errs is not None:
                errs[name] = rel_l2(force[name], v)
            return force[name].to(torch.float32)
        return v

    def h1rec(block):
        return lambda v: rec(block + ".h1", v)

    def subrec(block):
        return lambda suffix, v: rec(block + "." + suffix, v)

    h = rec("to_in", q.r(F.conv1d(x, p["unet.to_in.to_in.weight"], stride=cfg.stride, padding=pad)))  # :584-591 (fp32 weights on the device too)
    temb = rec("temb", time_embedding(p, t))
    if classes is not None:                                                       # :877, resblocks :306-308
        temb = torch.cat((temb, rec("class_emb", label_embedding(p, classes, cond_drop_prob))), dim=-1)
    skips_list: List[List[torch.Tensor]] = []
    for i in range(n):                                                            # :792-801, :441-468
        pre = f"unet.downsamples.{i}"
        h = rec(f"down{i}.conv", downsample_conv(p, f"{pre}.downsample", h, cfg.factors[i], cfg.kernel_multiplier_downsample, q))
        skips = []
        for j in range(cfg.num_blocks[i]):
            h = rec(f"down{i}.block{j}", resnet_block(p, f"{pre}.blocks.{j}", h, temb, g, q, rec_h1=h1rec(f"down{i}.block{j}")))
            skips.append(h)
        if cfg.attentions[i]:
            h = rec(f"down{i}.attn", transformer_block(p, f"{pre}.transformer", h, heads, q, rec=subrec(f"down{i}.attn")))
            skips.append(h)
        skips_list.append(skips)
    h = rec("mid.pre", resnet_block(p, "unet.bottleneck.pre_block", h, temb, g, q, rec_h1=h1rec("mid.pre")))    # :374-379
    if cfg.use_attention_bottleneck:
        h = rec("mid.attn", transformer_block(p, "unet.bottleneck.transformer", h, heads, q, rec=subrec("mid.attn")))
    h = rec("mid.post", resnet_block(p, "unet.bottleneck.post_block", h, temb, g, q, rec_h1=h1rec("mid.post")))
    skip_scale = 2 ** -0.5 if cfg.use_skip_scale else 1.0
    for u, i in enumerate(reversed(range(n))):                                    # :807-812, :542-566
        pre = f"unet.upsamples.{u}"
        skips = skips_list.pop()
        nb = cfg.num_blocks[i] + (1 if cfg.attentions[i] else 0)
        for j in range(nb):
            sk = skips.pop() * skip_scale
            hx = torch.cat([h, sk], dim=1)                                        # :539-540
            hraw = torch.cat([h, q.r(sk)], dim=1) if q.bf16 else None
            h = rec(f"up{u}.block{j}", resnet_block(p, f"{pre}.blocks.{j}", hx, temb, g, q, x_raw=hraw, rec_h1=h1rec(f"up{u}.block{j}")))
        if cfg.attentions[i]:
            h = rec(f"up{u}.attn", transformer_block(p, f"{pre}.transformer", h, heads, q, rec=subrec(f"up{u}.attn")))
        h = rec(f"up{u}.conv", upsample_conv(p, f"{pre}.upsample", h, cfg.factors[i], q, nearest=cfg.use_nearest_upsample))
    # :611-622; the device's bf16 MFMA route (num_filters a multiple of 16, <= 128) rounds the weight as an operand
    w_out = p["unet.to_out.to_out.weight"]
    if q.bf16 and cfg.num_filters % 16 == 0 and cfg.num_filters <= 128:
        w_out = q.w(w_out)
    return F.conv_transpose1d(h, w_out, stride=cfg.stride, padding=pad)
