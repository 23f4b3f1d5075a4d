"""ORACLE (test infrastructure, not product code) -- the ADM-style 2-D U-Net of BASELINE config 4
(SURVEY.md section 8f row 3), restated on CPU fp32 as pure functions over a ``{state_dict key: tensor}``
mapping.  See oracle/unet1d.py for the rules on who may import this package.

Parity status: PINNED against the reference itself (``src/models/backbones/unet2d_oai.py`` imported on CPU in
the build container by ``oracle/gen_golden_next.py``; fixtures in ``tests/golden/next_golden.npz``).  There is no
device path for this network yet: this file and its fixtures are the groundwork the HIP conv2d path will be
held to.

Every function cites the reference lines it restates (paths relative to the reference repo root; all line
numbers are in ``src/models/backbones/unet2d_oai.py`` unless another file is named).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from audiodiffuser_amd.weights import generate_tensor
from .unet1d import label_embedding

P = Dict[str, torch.Tensor]
Spec = Tuple[Tuple[int, ...], str]


@dataclass
class ADMConfig:
    """Constructor arguments of ``UNetModel`` (:410-430), same names and defaults."""
    image_size: int = 256
    in_channels: int = 2
    model_channels: int = 128
    out_channels: int = 2
    num_res_blocks: int = 2
    attention_resolutions: str = "16"
    channel_mult: Tuple[int, ...] = (1, 2, 2, 4)
    conv_resample: bool = True
    num_classes: Optional[int] = None
    num_heads: int = 8
    num_head_channels: int = -1
    use_scale_shift_norm: bool = True
    resblock_updown: bool = False
    use_new_attention_order: bool = False

    def to_kwargs(self) -> dict:
        return dict(self.__dict__)

    @property
    def attention_ds(self) -> Tuple[int, ...]:
        """:433-436 -- the constructor turns resolutions into downsample factors."""
        return tuple(self.image_size // int(r) for r in self.attention_resolutions.split(","))

    def heads(self, ch: int) -> int:
        """:296-302"""
        return self.num_heads if self.num_head_channels == -1 else ch // self.num_head_channels


def config_c4() -> ADMConfig:
    """BASELINE config 4: 1 x 80 x 256 mel input; every other argument is the constructor default (attention only in
    the middle block: ds = 16 is never reached with four levels)."""
    return ADMConfig(in_channels=1, out_channels=1)


def config_c4_small() -> ADMConfig:
    """Fixture size: two levels, attention at the second level and in the middle block, 32 groups still divide."""
    return ADMConfig(image_size=32, in_channels=1, model_channels=32, out_channels=1, num_res_blocks=1,
                     attention_resolutions="16", channel_mult=(1, 2), num_heads=2)


# ------------------------------------------------------------------ structure (shared by specs and forward)
@dataclass
class _Layer:
    kind: str                  # "conv" | "res" | "attn" | "down" | "up"
    pre: str
    cin: int = 0
    cout: int = 0
    up: bool = False
    down: bool = False


@dataclass
class _Structure:
    input_blocks: List[List[_Layer]] = field(default_factory=list)
    middle: List[_Layer] = field(default_factory=list)
    output_blocks: List[List[_Layer]] = field(default_factory=list)
    final_ch: int = 0
    input_ch: int = 0


def structure(cfg: ADMConfig) -> _Structure:
    """The module list ``UNetModel.__init__`` builds (:467-594), as data."""
    s = _Structure()
    mc = cfg.model_channels
    att = cfg.attention_ds
    ch = s.input_ch = int(cfg.channel_mult[0] * mc)
    s.input_blocks.append([_Layer("conv", "input_blocks.0.0", cfg.in_channels, ch)])
    chans = [ch]
    ds = 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            i = len(s.input_blocks)
            layers = [_Layer("res", f"input_blocks.{i}.0", ch, int(mult * mc))]
            ch = int(mult * mc)
            if ds in att:
                layers.append(_Layer("attn", f"input_blocks.{i}.1", ch, ch))
            s.input_blocks.append(layers)
            chans.append(ch)
        if level != len(cfg.channel_mult) - 1:
            i = len(s.input_blocks)
            if cfg.resblock_updown:
                s.input_blocks.append([_Layer("res", f"input_blocks.{i}.0", ch, ch, down=True)])
            else:
                s.input_blocks.append([_Layer("down", f"input_blocks.{i}.0", ch, ch)])
            chans.append(ch)
            ds *= 2
    s.middle = [_Layer("res", "middle_block.0", ch, ch), _Layer("attn", "middle_block.1", ch, ch),
                _Layer("res", "middle_block.2", ch, ch)]
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            k = len(s.output_blocks)
            layers = [_Layer("res", f"output_blocks.{k}.0", ch + ich, int(mc * mult))]
            ch = int(mc * mult)
            if ds in att:
                layers.append(_Layer("attn", f"output_blocks.{k}.{len(layers)}", ch, ch))
            if level and i == cfg.num_res_blocks:
                j = len(layers)
                if cfg.resblock_updown:
                    layers.append(_Layer("res", f"output_blocks.{k}.{j}", ch, ch, up=True))
                else:
                    layers.append(_Layer("up", f"output_blocks.{k}.{j}", ch, ch))
                ds //= 2
            s.output_blocks.append(layers)
    s.final_ch = ch
    return s


def param_specs(cfg: ADMConfig) -> "OrderedDict[str, Spec]":
    """Every ``UNetModel.state_dict()`` key with its shape, in the module's registration order."""
    out: "OrderedDict[str, Spec]" = OrderedDict()
    mc = cfg.model_channels
    ted = 4 * mc
    out["time_embed.0.weight"] = ((ted, mc), "linear_w")
    out["time_embed.0.bias"] = ((ted,), "bias")
    out["time_embed.2.weight"] = ((ted, ted), "linear_w")
    out["time_embed.2.bias"] = ((ted,), "bias")
    if cfg.num_classes is not None:        # conditioner.py:64-90
        out["label_conditioner.null_classes_emb"] = ((1, mc), "embed")
        out["label_conditioner.label_emb.weight"] = ((cfg.num_classes, mc), "embed")
        out["label_conditioner.class_to_cond.0.weight"] = ((mc,), "norm_w")
        out["label_conditioner.class_to_cond.0.bias"] = ((mc,), "norm_b")
        out["label_conditioner.class_to_cond.1.weight"] = ((ted, mc), "linear_w")
        out["label_conditioner.class_to_cond.1.bias"] = ((ted,), "bias")
        out["label_conditioner.class_to_cond.3.weight"] = ((ted, ted), "linear_w")
        out["label_conditioner.class_to_cond.3.bias"] = ((ted,), "bias")

    def conv(pre, cin, cout, k):
        out[f"{pre}.weight"] = ((cout, cin, k, k), "conv2d_w")
        out[f"{pre}.bias"] = ((cout,), "bias")

    def norm(pre, c):
        out[f"{pre}.weight"] = ((c,), "norm_w")
        out[f"{pre}.bias"] = ((c,), "norm_b")

    def layer(l: _Layer):
        if l.kind == "conv":
            conv(l.pre, l.cin, l.cout, 3)
        elif l.kind == "res":              # :194-235
            norm(f"{l.pre}.in_layers.0", l.cin)
            conv(f"{l.pre}.in_layers.2", l.cin, l.cout, 3)
            out[f"{l.pre}.emb_layers.1.weight"] = (((2 if cfg.use_scale_shift_norm else 1) * l.cout, ted), "linear_w")
            out[f"{l.pre}.emb_layers.1.bias"] = (((2 if cfg.use_scale_shift_norm else 1) * l.cout,), "bias")
            norm(f"{l.pre}.out_layers.0", l.cout)
            conv(f"{l.pre}.out_layers.3", l.cout, l.cout, 3)
            if l.cin != l.cout:
                conv(f"{l.pre}.skip_connection", l.cin, l.cout, 1)
        elif l.kind == "attn":             # :303-315
            norm(f"{l.pre}.norm", l.cin)
            out[f"{l.pre}.qkv.weight"] = ((3 * l.cin, l.cin, 1), "conv_w")
            out[f"{l.pre}.qkv.bias"] = ((3 * l.cin,), "bias")
            out[f"{l.pre}.proj_out.weight"] = ((l.cin, l.cin, 1), "conv_w")
            out[f"{l.pre}.proj_out.bias"] = ((l.cin,), "bias")
        elif l.kind == "down":             # :147-151
            if cfg.conv_resample:
                conv(f"{l.pre}.op", l.cin, l.cout, 3)
        elif l.kind == "up":               # :118-119
            if cfg.conv_resample:
                conv(f"{l.pre}.conv", l.cin, l.cout, 3)

    s = structure(cfg)
    for blk in s.input_blocks:
        for l in blk:
            layer(l)
    for l in s.middle:
        layer(l)
    for blk in s.output_blocks:
        for l in blk:
            layer(l)
    norm("out.0", s.final_ch)
    conv("out.2", s.input_ch, cfg.out_channels, 3)
    return out


def generate_weights(cfg: ADMConfig, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Name-keyed deterministic weights (same generator as audiodiffuser_amd/weights.py).  The reference zero-initialises
    every ResBlock's second conv, every attention projection and the output conv (``zero_module``, :227,309,599): a
    random-init net is then the identity-plus-nothing and a parity check vacuous, so these are random here too."""
    out = OrderedDict()
    for k, (shape, kind) in param_specs(cfg).items():
        if kind == "conv2d_w":
            g = generate_tensor(k, shape, "embed", seed)
            out[k] = g * (1.0 / (shape[1] * shape[2] * shape[3]) ** 0.5)
        else:
            out[k] = generate_tensor(k, shape, kind, seed)
    return out


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


def conv2d(p: P, pre: str, x: torch.Tensor, stride: int = 1) -> torch.Tensor:
    w = p[f"{pre}.weight"]
    return F.conv2d(x, w, p[f"{pre}.bias"], stride=stride, padding=w.shape[-1] // 2)


def upsample(p: P, pre: str, x: torch.Tensor, use_conv: bool) -> torch.Tensor:
    """:122-127 -- nearest x2, then the 3x3 conv."""
    x = F.interpolate(x, scale_factor=2, mode="nearest")
    return conv2d(p, f"{pre}.conv", x) if use_conv else x


def downsample(p: P, pre: str, x: torch.Tensor, use_conv: bool) -> torch.Tensor:
    """:146-158 -- 3x3 stride-2 conv (padding 1), or a 2x2 average pool."""
    return conv2d(p, f"{pre}.op", x, stride=2) if use_conv else F.avg_pool2d(x, 2, 2)


def res_block(p: P, l: _Layer, x: torch.Tensor, emb: torch.Tensor, scale_shift: bool) -> torch.Tensor:
    """:248-272.  ``emb`` is the 4*model_channels embedding; the block applies SiLU + Linear to it (:214-220)."""
    pre = l.pre
    h = F.silu(group_norm32(p, f"{pre}.in_layers.0", x))
    if l.up or l.down:                     # :249-254: resample the activated h and the raw x, then convolve
        if l.up:
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        else:
            h = F.avg_pool2d(h, 2, 2)
            x = F.avg_pool2d(x, 2, 2)
    h = conv2d(p, f"{pre}.in_layers.2", h)
    e = F.linear(F.silu(emb), p[f"{pre}.emb_layers.1.weight"], p[f"{pre}.emb_layers.1.bias"])[:, :, None, None]
    if scale_shift:                        # :262-267
        scale, shift = torch.chunk(e, 2, dim=1)
        h = group_norm32(p, f"{pre}.out_layers.0", h) * (1 + scale) + shift
    else:                                  # :268-270
        h = group_norm32(p, f"{pre}.out_layers.0", h + e)
    h = conv2d(p, f"{pre}.out_layers.3", F.silu(h))       # dropout is the identity at inference
    skip = conv2d(p, f"{pre}.skip_connection", x) if l.cin != l.cout else x
    return skip + h


def qkv_attention(qkv: torch.Tensor, heads: int, legacy: bool) -> torch.Tensor:
    """:324-350 (legacy: heads split first, each head's rows are q|k|v) and :352-380 (new order: q|k|v split first).
    Both scale q and k by ch^-1/4 and take the softmax in fp32."""
    b, width, n = qkv.shape
    ch = width // (3 * heads)
    if legacy:
        q, k, v = qkv.reshape(b * heads, 3 * ch, n).split(ch, dim=1)
    else:
        q, k, v = (t.reshape(b * heads, ch, n) for t in qkv.chunk(3, dim=1))
    scale = 1 / math.sqrt(math.sqrt(ch))
    w = torch.einsum("bct,bcs->bts", q * scale, k * scale)
    w = torch.softmax(w.float(), dim=-1)
    return torch.einsum("bts,bcs->bct", w, v).reshape(b, -1, n)


def attention_block(p: P, l: _Layer, x: torch.Tensor, heads: int, legacy: bool) -> torch.Tensor:
    """:316-322.  The residual is added to the NORMALISED input (the reference reassigns ``x = self.norm(x)``)."""
    b, c = x.shape[:2]
    xn = group_norm32(p, f"{l.pre}.norm", x.reshape(b, c, -1))
    qkv = F.conv1d(xn, p[f"{l.pre}.qkv.weight"], p[f"{l.pre}.qkv.bias"])
    h = qkv_attention(qkv, heads, legacy)
    h = F.conv1d(h, p[f"{l.pre}.proj_out.weight"], p[f"{l.pre}.proj_out.bias"])
    return (xn + h).reshape(x.shape)


# ------------------------------------------------------------------ the network
def unet2d_forward(p: P, cfg: ADMConfig, x: torch.Tensor, t: torch.Tensor, classes: Optional[torch.Tensor] = None,
                   cond_drop_prob: float = 0.0, taps: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """``UNetModel.forward`` :603-634.  x: [B, in_channels, H, W], t: [B] (the EDM wrapper passes c_noise).
    ``taps`` (optional) receives the output of every input / middle / output block under the block's module name."""
    assert (classes is not None) == (cfg.num_classes is not None), "must specify y if and only if the model is class-conditional"
    s = structure(cfg)

    def rec(name, v):
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
            if l.kind == "conv":
                h = conv2d(p, l.pre, h)
            elif l.kind == "res":
                h = res_block(p, l, h, emb, cfg.use_scale_shift_norm)
            elif l.kind == "attn":
                h = attention_block(p, l, h, cfg.heads(l.cin), not cfg.use_new_attention_order)
            elif l.kind == "down":
                h = downsample(p, l.pre, h, cfg.conv_resample)
            elif l.kind == "up":
                h = upsample(p, l.pre, h, cfg.conv_resample)
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
    return conv2d(p, "out.2", h)
