"""State-dict layout of the reference ``UNet1dBase`` and a repo-owned deterministic
weight / noise generator.

* ``param_specs(cfg)`` lists every ``state_dict()`` key of the reference module with
  its shape (reference: src/models/backbones/unet1d.py).  The
  key set is the checkpoint-compatibility contract of the ``model.net`` plugin.
* ``generate_weights(cfg, seed)`` fills those tensors from a generator keyed by the
  parameter *name*, so the GPU box regenerates bit-identical weights without the
  reference being present.  The zero-initialised output layer
  (reference: unet1d.py:619) is re-randomised, otherwise every output is 0 and a
  parity check would be vacuous.
* ``generate_noise`` keys the initial noise by the *global sample index*, so
  sharding a batch over ranks does not change any sample.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import torch

from .config import UNet1dConfig, WaveNetConfig

Spec = Tuple[Tuple[int, ...], str]  # (shape, kind)


def _resblock_specs(out: "OrderedDict[str, Spec]", pre: str, cin: int, cout: int, temb: int) -> None:
    out[f"{pre}.to_cond_embedding.1.weight"] = ((2 * cout, temb), "linear_w")
    out[f"{pre}.to_cond_embedding.1.bias"] = ((2 * cout,), "bias")
    out[f"{pre}.block1.groupnorm.weight"] = ((cin,), "norm_w")
    out[f"{pre}.block1.groupnorm.bias"] = ((cin,), "norm_b")
    out[f"{pre}.block1.project.weight"] = ((cout, cin, 3), "conv_w")
    out[f"{pre}.block1.project.bias"] = ((cout,), "bias")
    out[f"{pre}.block2.groupnorm.weight"] = ((cout,), "norm_w")
    out[f"{pre}.block2.groupnorm.bias"] = ((cout,), "norm_b")
    out[f"{pre}.block2.project.weight"] = ((cout, cout, 3), "conv_w")
    out[f"{pre}.block2.project.bias"] = ((cout,), "bias")
    if cin != cout:
        out[f"{pre}.to_out.weight"] = ((cout, cin, 1), "conv_w")
        out[f"{pre}.to_out.bias"] = ((cout,), "bias")


def _transformer_specs(out: "OrderedDict[str, Spec]", pre: str, c: int, mult: int) -> None:
    out[f"{pre}.norm.weight"] = ((c,), "norm_w")
    out[f"{pre}.norm.bias"] = ((c,), "norm_b")
    out[f"{pre}.attention.to_q.weight"] = ((c, c), "linear_w")
    out[f"{pre}.attention.to_kv.weight"] = ((2 * c, c), "linear_w")
    out[f"{pre}.attention.to_out.weight"] = ((c, c), "linear_w")
    mid = int(c * mult)
    out[f"{pre}.feed_forward.0.g"] = ((1, c, 1), "norm_w")
    out[f"{pre}.feed_forward.1.weight"] = ((mid, c, 1), "conv_w")
    out[f"{pre}.feed_forward.3.g"] = ((1, mid, 1), "norm_w")
    out[f"{pre}.feed_forward.4.weight"] = ((c, mid, 1), "conv_w")


def param_specs(cfg: UNet1dConfig) -> "OrderedDict[str, Spec]":
    """All ``UNet1dBase.state_dict()`` keys -> (shape, kind), reference order."""
    cfg.validate()
    ch, temb = cfg.channels, cfg.time_embed_dim
    n = cfg.num_layers
    s: "OrderedDict[str, Spec]" = OrderedDict()
    if cfg.class_cond:      # LabelEmbedder is registered before the U-Net (unet1d.py:841-847, conditioner.py:64-90)
        cdim = cfg.classes_dim
        s["label_conditioner.null_classes_emb"] = ((1, ch), "embed")
        s["label_conditioner.label_emb.weight"] = ((cfg.num_classes, ch), "embed")
        s["label_conditioner.class_to_cond.0.weight"] = ((ch,), "norm_w")
        s["label_conditioner.class_to_cond.0.bias"] = ((ch,), "norm_b")
        s["label_conditioner.class_to_cond.1.weight"] = ((cdim, ch), "linear_w")
        s["label_conditioner.class_to_cond.1.bias"] = ((cdim,), "bias")
        s["label_conditioner.class_to_cond.3.weight"] = ((cdim, cdim), "linear_w")
        s["label_conditioner.class_to_cond.3.bias"] = ((cdim,), "bias")
    temb = temb + cfg.classes_dim   # every FiLM projection reads cat(time_embed, class_embed) (unet1d.py:272, :306-308)
    tdim = cfg.time_embed_dim
    s["unet.to_in.to_in.weight"] = ((cfg.num_filters, cfg.in_channels, cfg.window_length), "conv_w")
    s["unet.to_out.to_out.weight"] = ((cfg.num_filters, cfg.out_channels, cfg.window_length), "convT_w")
    s["unet.to_time.0.0.weights"] = ((ch // 2,), "fourier")
    s["unet.to_time.0.1.weight"] = ((tdim, ch + 1), "linear_w")
    s["unet.to_time.0.1.bias"] = ((tdim,), "bias")
    s["unet.to_time.2.weight"] = ((tdim, tdim), "linear_w")
    s["unet.to_time.2.bias"] = ((tdim,), "bias")
    for i in range(n):
        cin, cout = ch * cfg.multipliers[i], ch * cfg.multipliers[i + 1]
        f = cfg.factors[i]
        pre = f"unet.downsamples.{i}"
        s[f"{pre}.downsample.weight"] = ((cout, cin, f * cfg.kernel_multiplier_downsample + 1), "conv_w")
        s[f"{pre}.downsample.bias"] = ((cout,), "bias")
        for j in range(cfg.num_blocks[i]):
            _resblock_specs(s, f"{pre}.blocks.{j}", cout, cout, temb)
        if cfg.attentions[i]:
            _transformer_specs(s, f"{pre}.transformer", cout, cfg.attention_multiplier)
    cb = ch * cfg.multipliers[-1]
    _resblock_specs(s, "unet.bottleneck.pre_block", cb, cb, temb)
    if cfg.use_attention_bottleneck:
        _transformer_specs(s, "unet.bottleneck.transformer", cb, cfg.attention_multiplier)
    _resblock_specs(s, "unet.bottleneck.post_block", cb, cb, temb)
    for u, i in enumerate(reversed(range(n))):
        cin, cout = ch * cfg.multipliers[i + 1], ch * cfg.multipliers[i]
        f = cfg.factors[i]
        pre = f"unet.upsamples.{u}"
        nb = cfg.num_blocks[i] + (1 if cfg.attentions[i] else 0)
        for j in range(nb):
            _resblock_specs(s, f"{pre}.blocks.{j}", 2 * cin, cin, temb)
        if cfg.attentions[i]:
            _transformer_specs(s, f"{pre}.transformer", cin, cfg.attention_multiplier)
        if cfg.use_nearest_upsample:          # nn.Sequential(Upsample, ReflectionPad1d, Conv1d): the conv is module 2 (unet1d.py:236-246)
            s[f"{pre}.upsample.2.weight"] = ((cout, cin, 3), "conv_w")
            s[f"{pre}.upsample.2.bias"] = ((cout,), "bias")
        else:
            s[f"{pre}.upsample.weight"] = ((cin, cout, 2 * f), "convT_w")
            s[f"{pre}.upsample.bias"] = ((cout,), "bias")
    return s


def _seed_for(name: str, seed: int) -> int:
    return (zlib.crc32(name.encode()) + 0x9E3779B1 * (seed + 1)) % (2 ** 62)


def _fan_in(shape: Tuple[int, ...], kind: str) -> int:
    if kind == "conv_w":       # (Cout, Cin, k)
        return shape[1] * shape[2]
    if kind == "convT_w":      # (Cin, Cout, k): each output sees Cin * ceil(k/stride) taps
        return shape[0] * max(1, shape[2] // 2)
    if kind == "linear_w":     # (out, in)
        return shape[1]
    return 1


def generate_tensor(name: str, shape: Tuple[int, ...], kind: str, seed: int = 0) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed(_seed_for(name, seed))
    z = torch.randn(shape, generator=g, dtype=torch.float32)
    if kind in ("conv_w", "convT_w", "linear_w"):
        return z * (1.0 / _fan_in(shape, kind) ** 0.5)
    if kind == "bias":
        return z * 0.1
    if kind == "norm_w":
        return 1.0 + 0.1 * z
    if kind == "norm_b":
        return 0.1 * z
    if kind in ("fourier", "embed"):
        return z
    raise ValueError(kind)


def generate_weights(cfg: UNet1dConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Name-keyed deterministic fp32 weights for every reference state_dict key."""
    return OrderedDict((k, generate_tensor(k, shp, kind, seed))
                       for k, (shp, kind) in param_specs(cfg).items())


def generate_noise(first_sample: int, count: int, length: int, channels: int = 1,
                   base_seed: int = 1234) -> torch.Tensor:
    """Unit-variance initial noise ``[count, channels, length]``; sample ``i`` of the
    global batch always gets generator seed ``base_seed + i``."""
    out = torch.empty(count, channels, length, dtype=torch.float32)
    for j in range(count):
        g = torch.Generator(device="cpu")
        g.manual_seed(base_seed + first_sample + j)
        out[j] = torch.randn(channels, length, generator=g, dtype=torch.float32)
    return out


def count_parameters(cfg: UNet1dConfig) -> int:
    n = 0
    for shp, _ in param_specs(cfg).values():
        k = 1
        for d in shp:
            k *= d
        n += k
    return n


# ---------------------------------------------------------------------------------------------- WaveNetNoise (BASELINE configs[4])
def wavenet_param_specs(cfg: WaveNetConfig) -> "OrderedDict[str, Spec]":
    """Every ``WaveNetNoise.state_dict()`` key in registration order (reference: src/models/backbones/wavenet.py:153-167).  The
    custom ``WeightNorm`` (:15-55) deletes ``weight`` and registers a 0-dim ``weight_g`` (the norm of the WHOLE tensor, :29)
    and ``weight_v`` after ``bias``."""
    out: "OrderedDict[str, Spec]" = OrderedDict()
    c = cfg.residual_channels

    def wn_conv(pre, cin, cout, k):
        out[f"{pre}.conv.module.bias"] = ((cout,), "bias")
        out[f"{pre}.conv.module.weight_g"] = ((), "wn_g")
        out[f"{pre}.conv.module.weight_v"] = ((cout, cin, k), "conv_w")

    wn_conv("input_projection", 1, c, 1)
    out["residual_layer.fc_t1.weight"] = ((cfg.dim_mid, cfg.dim_in), "linear_w")
    out["residual_layer.fc_t1.bias"] = ((cfg.dim_mid,), "bias")
    out["residual_layer.fc_t2.weight"] = ((cfg.dim_out, cfg.dim_mid), "linear_w")
    out["residual_layer.fc_t2.bias"] = ((cfg.dim_out,), "bias")
    for n in range(cfg.residual_layers):
        pre = f"residual_layer.residual_blocks.{n}"
        wn_conv(f"{pre}.dilated_conv", c, 2 * c, 3)
        out[f"{pre}.diffusion_projection.weight"] = ((c, cfg.dim_out), "linear_w")
        out[f"{pre}.diffusion_projection.bias"] = ((c,), "bias")
        wn_conv(f"{pre}.output_projection", c, 2 * c, 1)
    wn_conv("skip_projection", c, c, 1)
    out["output_projection.conv.weight"] = ((1, c, 1), "conv_w")
    out["output_projection.conv.bias"] = ((1,), "bias")
    return out


def generate_wavenet_weights(cfg: WaveNetConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Name-keyed deterministic weights.  ``weight_g`` is drawn so that the effective weight ``v * g / ||v||`` has the usual
    1/sqrt(fan_in) scale per element (g = sqrt(Cout) times a name-keyed factor in [0.8, 1.2]); the zero-initialised output
    conv (``ZeroConv1d`` wavenet.py:57-66) is random here, otherwise every output is 0 and a parity check vacuous."""
    specs = wavenet_param_specs(cfg)
    out = OrderedDict()
    for k, (shape, kind) in specs.items():
        if kind == "wn_g":
            vshape = specs[k[:-1] + "v"][0]
            u = generate_tensor(k, (1,), "embed", seed).clamp(-2, 2)[0]
            out[k] = (vshape[0] ** 0.5) * (1.0 + 0.1 * u)
        else:
            out[k] = generate_tensor(k, shape, kind, seed)
    return out
