"""Configuration, module structure and ``state_dict`` layout of the reference's ADM-style 2-D U-Net ``UNetModel``
(src/models/backbones/unet2d_oai.py:382-635; BASELINE configs[3]), plus the repo-owned deterministic weight
generator for it.  Shared by the plugin (audiodiffuser_amd/adm.py), the C-ABI registry order and the oracle."""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

from .weights import generate_tensor

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
    def class_cond(self) -> bool:
        """What the shared denoise / sampler host code asks of a network config."""
        return self.num_classes is not None

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


