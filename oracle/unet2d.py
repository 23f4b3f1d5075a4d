"""ORACLE (test infrastructure, not product code) -- the Imagen-style 2-D U-Net ``UNet2dBase`` that every shipped
experiment config of the reference instantiates (``src/models/backbones/unet2d.py:622``; e.g.
``configs/experiment/sc09_inference/diffunet_complex_sc09_eval_dpm.yaml:38-50``), restated on CPU fp32 as pure functions over a
``{state_dict key: tensor}`` mapping.  See oracle/unet1d.py for the rules on who may import this package.

Status: an ORACLE-FIRST START (SURVEY.md section 2 row "2-D U-Net (Imagen-style)", not a section-8 row): there is
no device path behind it yet.  Parity status of the restatement itself: PINNED against the reference, imported on CPU in the build
container by ``oracle/gen_golden_unet2d.py`` (state_dict key order + shapes for four constructor variants, the forward output and the
output of every down / middle / up block; fixtures in ``tests/golden/unet2d_golden.npz``, held by tests/test_oracle_unet2d.py).

Every function cites the reference lines it restates; line numbers are in ``src/models/backbones/unet2d.py`` unless another file is
named.  What the shipped configs never reach raises instead of being guessed: text conditioning (``cond_on_text``), linear attention,
cross-embed downsampling, the condition encoder (``use_condition_block`` / ``inj_channels``), the upsample combiner and the
init-conv-to-final-conv residual.  Modules that exist in the ``state_dict`` but that the forward never runs with ``c=None`` and no text
(``ResnetBlock.cross_attn``, ``Attention.to_context``, ``to_time_tokens``) are in ``param_specs`` and nowhere else.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field, asdict
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from .unet1d import label_embedding

P = Dict[str, torch.Tensor]
Spec = Tuple[Tuple[int, ...], str]


# ------------------------------------------------------------------ configuration (the constructor's arguments, :623-667)
@dataclass
class UNet2dConfig:
    dim: int = 128
    num_classes: int = 0
    num_resnet_blocks: int = 1
    cond_dim: Optional[int] = None
    num_time_tokens: int = 2
    learned_sinu_pos_emb_dim: int = 16
    dim_mults: Tuple[int, ...] = (1, 2, 4, 8)
    channels: int = 3
    channels_out: Optional[int] = None
    attn_heads: int = 8
    ff_mult: float = 2.0
    layer_attns: Tuple[bool, ...] = (True, True, True, True)
    layer_attns_depth: int = 1
    layer_mid_attns_depth: int = 1
    attend_at_middle: bool = True
    layer_cross_attns: Tuple[bool, ...] = (True, True, True, True)
    init_dim: Optional[int] = None
    resnet_groups: int = 8
    init_conv_kernel_size: int = 7
    init_cross_embed: bool = True
    init_cross_embed_kernel_sizes: Tuple[int, ...] = (3, 7, 15)
    memory_efficient: bool = False
    use_global_context_attn: bool = True
    scale_skip_connection: bool = True
    final_resnet_block: bool = True
    final_conv_kernel_size: int = 3
    pixel_shuffle_upsample: bool = True

    def to_kwargs(self) -> dict:
        kw = asdict(self)
        for k in ("dim_mults", "layer_attns", "layer_cross_attns"):
            kw[k] = list(kw[k])
        kw["init_cross_embed_kernel_sizes"] = tuple(kw["init_cross_embed_kernel_sizes"])
        return kw

    # derived quantities (:672-676, :692-693, :749)
    @property
    def dims(self) -> List[int]:
        init_dim = self.init_dim if self.init_dim is not None else self.dim
        return [init_dim] + [self.dim * m for m in self.dim_mults]

    @property
    def in_out(self) -> List[Tuple[int, int]]:
        d = self.dims
        return list(zip(d[:-1], d[1:]))

    @property
    def cdim(self) -> int:
        return self.cond_dim if self.cond_dim is not None else self.dim

    @property
    def time_cond_dim(self) -> int:
        return self.cdim * 4

    @property
    def skip_scale(self) -> float:
        return 2 ** -0.5 if self.scale_skip_connection else 1.0

    def check(self) -> None:
        n = len(self.dim_mults)
        assert self.attn_heads > 1 and self.dim > 100                                      # :670-671
        assert len(self.layer_attns) == n and len(self.layer_cross_attns) == n            # :753
        assert self.num_classes == 0 or self.cdim == self.dim, "t + classes_emb needs 4 * cond_dim == 4 * dim (:718-726, :902)"


def config_sc09(num_classes: int = 10) -> UNet2dConfig:
    """``net:`` of configs/experiment/sc09_inference/diffunet_complex_sc09_eval_dpm.yaml:38-50 (SC09: ten spoken digits; the network sees the
    2-channel real / imaginary spectrogram, 256 x 128 at n_fft 510 / hop 128 / 16384 samples)."""
    return UNet2dConfig(dim=128, num_classes=num_classes, dim_mults=(1, 2, 2, 2), channels=2, num_resnet_blocks=2, resnet_groups=8,
                        layer_attns=(False, False, True, True), layer_cross_attns=(False, False, True, True), attn_heads=2, ff_mult=2.0,
                        memory_efficient=True)


def config_sc09_small(num_classes: int = 10) -> UNet2dConfig:
    """The same structure at the smallest width the constructor accepts (``dim > 100``, a multiple of the 8 groups and the 2 heads) and
    three levels: the fixture net."""
    return UNet2dConfig(dim=104, num_classes=num_classes, dim_mults=(1, 2, 2), channels=2, num_resnet_blocks=2, resnet_groups=8,
                        layer_attns=(False, True, True), layer_cross_attns=(False, True, True), attn_heads=2, ff_mult=2.0,
                        memory_efficient=True)


def fixture_variants() -> "Dict[str, Tuple[UNet2dConfig, Tuple[int, int, int]]]":
    """tag -> (constructor arguments, (batch, H, W)) of the nets oracle/gen_golden_unet2d.py runs through the reference and tests/test_oracle_unet2d.py
    through this restatement."""
    return {
        "small": (config_sc09_small(), (2, 32, 16)),
        "nomem": (UNet2dConfig(dim=104, num_classes=0, dim_mults=(1, 2), channels=3, num_resnet_blocks=1, resnet_groups=8,
                               layer_attns=(False, True), layer_cross_attns=(False, False), attn_heads=4, ff_mult=2.0,
                               memory_efficient=False, pixel_shuffle_upsample=False, init_cross_embed=False, scale_skip_connection=False),
                  (2, 16, 16)),
        "nogca": (UNet2dConfig(dim=104, num_classes=4, dim_mults=(1, 2), channels=2, num_resnet_blocks=1, resnet_groups=4,
                               layer_attns=(True, True), layer_cross_attns=(True, True), layer_attns_depth=2, attn_heads=2, ff_mult=1.5,
                               memory_efficient=True, use_global_context_attn=False, final_resnet_block=False, attend_at_middle=False),
                  (2, 16, 8)),
        "sc09": (config_sc09(), (1, 64, 32)),
    }


# ------------------------------------------------------------------ state_dict layout (the registration order of __init__)
def _resnet_specs(out: "OrderedDict[str, Spec]", pre: str, din: int, dout: int, tcd: Optional[int], cond_dim: Optional[int], gca: bool) -> None:
    """ResnetBlock.__init__ :106-144: time_mlp, cross_attn, block1, block2, gca, res_conv in that order."""
    if tcd is not None:
        out[f"{pre}.time_mlp.1.weight"] = ((dout * 2, tcd), "linear_w")
        out[f"{pre}.time_mlp.1.bias"] = ((dout * 2,), "bias")
    if cond_dim is not None:                                             # Attention(dim = dout, context_dim = cond_dim): attention_utils.py:96-110
        out[f"{pre}.cross_attn.to_q.weight"] = ((dout, dout), "linear_w")
        out[f"{pre}.cross_attn.to_kv.weight"] = ((2 * dout, dout), "linear_w")
        out[f"{pre}.cross_attn.to_context.weight"] = ((2 * dout, cond_dim), "linear_w")
        out[f"{pre}.cross_attn.to_out.weight"] = ((dout, dout), "linear_w")
    for blk, ci in (("block1", din), ("block2", dout)):                  # Block :83-94
        out[f"{pre}.{blk}.groupnorm.weight"] = ((ci,), "norm_w")
        out[f"{pre}.{blk}.groupnorm.bias"] = ((ci,), "norm_b")
        out[f"{pre}.{blk}.project.weight"] = ((dout, ci, 3, 3), "conv2d_w")
        out[f"{pre}.{blk}.project.bias"] = ((dout,), "bias")
    if gca:                                                              # GlobalContext :173-188
        hid = max(3, dout // 2)
        out[f"{pre}.gca.to_k.weight"] = ((1, dout, 1, 1), "conv2d_w")
        out[f"{pre}.gca.to_k.bias"] = ((1,), "bias")
        out[f"{pre}.gca.net.0.weight"] = ((hid, dout, 1, 1), "conv2d_w")
        out[f"{pre}.gca.net.0.bias"] = ((hid,), "bias")
        out[f"{pre}.gca.net.2.weight"] = ((dout, hid, 1, 1), "conv2d_w")
        out[f"{pre}.gca.net.2.bias"] = ((dout,), "bias")
    if din != dout:
        out[f"{pre}.res_conv.weight"] = ((dout, din, 1, 1), "conv2d_w")
        out[f"{pre}.res_conv.bias"] = ((dout,), "bias")


def _transformer_specs(out: "OrderedDict[str, Spec]", pre: str, dim: int, depth: int, ff_mult: float, context_dim: Optional[int]) -> None:
    """TransformerBlock.__init__ :198-217 (the ModuleList is assigned before ``norm``); Attention attention_utils.py:96-110; FeedForward :186-194."""
    hid = int(dim * ff_mult)
    for d in range(depth):
        out[f"{pre}.layers.{d}.0.to_q.weight"] = ((dim, dim), "linear_w")
        out[f"{pre}.layers.{d}.0.to_kv.weight"] = ((2 * dim, dim), "linear_w")
        if context_dim is not None:
            out[f"{pre}.layers.{d}.0.to_context.weight"] = ((2 * dim, context_dim), "linear_w")
        out[f"{pre}.layers.{d}.0.to_out.weight"] = ((dim, dim), "linear_w")
        out[f"{pre}.layers.{d}.1.0.g"] = ((dim,), "norm_w")
        out[f"{pre}.layers.{d}.1.1.weight"] = ((hid, dim), "linear_w")
        out[f"{pre}.layers.{d}.1.3.g"] = ((hid,), "norm_w")
        out[f"{pre}.layers.{d}.1.4.weight"] = ((dim, hid), "linear_w")
    out[f"{pre}.norm.g"] = ((dim,), "norm_w")


def param_specs(cfg: UNet2dConfig) -> "OrderedDict[str, Spec]":
    """state_dict key -> (shape, init kind), in the reference's registration order (checked against the imported module by the generator)."""
    cfg.check()
    o: "OrderedDict[str, Spec]" = OrderedDict()
    dims, in_out, tcd, cd = cfg.dims, cfg.in_out, cfg.time_cond_dim, cfg.cdim
    init_dim = dims[0]
    n = len(in_out)
    # init_conv :679-686 (CrossEmbedLayer :261-282, stride 1)
    if cfg.init_cross_embed:
        ks = sorted(cfg.init_cross_embed_kernel_sizes)
        scales = [int(init_dim / (2 ** i)) for i in range(1, len(ks))]
        scales = scales + [init_dim - sum(scales)]
        for i, (k, ds) in enumerate(zip(ks, scales)):
            o[f"init_conv.convs.{i}.weight"] = ((ds, cfg.channels, k, k), "conv2d_w")
            o[f"init_conv.convs.{i}.bias"] = ((ds,), "bias")
    else:
        k = cfg.init_conv_kernel_size
        o["init_conv.weight"] = ((init_dim, cfg.channels, k, k), "conv2d_w")
        o["init_conv.bias"] = ((init_dim,), "bias")
    # time conditioning :695-714
    o["to_time_hiddens.0.weights"] = ((cfg.learned_sinu_pos_emb_dim // 2,), "fourier")
    o["to_time_hiddens.1.weight"] = ((tcd, cfg.learned_sinu_pos_emb_dim + 1), "linear_w")
    o["to_time_hiddens.1.bias"] = ((tcd,), "bias")
    o["to_time_cond.0.weight"] = ((tcd, tcd), "linear_w")
    o["to_time_cond.0.bias"] = ((tcd,), "bias")
    o["to_time_tokens.0.weight"] = ((cd * cfg.num_time_tokens, tcd), "linear_w")
    o["to_time_tokens.0.bias"] = ((cd * cfg.num_time_tokens,), "bias")
    # LabelEmbedder :717-727 (conditioner.py:65-90)
    if cfg.num_classes != 0:
        cdm = cfg.dim * 4
        o["label_conditioner.null_classes_emb"] = ((1, cfg.dim), "embed")
        o["label_conditioner.label_emb.weight"] = ((cfg.num_classes, cfg.dim), "embed")
        o["label_conditioner.class_to_cond.0.weight"] = ((cfg.dim,), "norm_w")
        o["label_conditioner.class_to_cond.0.bias"] = ((cfg.dim,), "norm_b")
        o["label_conditioner.class_to_cond.1.weight"] = ((cdm, cfg.dim), "linear_w")
        o["label_conditioner.class_to_cond.1.bias"] = ((cdm,), "bias")
        o["label_conditioner.class_to_cond.3.weight"] = ((cdm, cdm), "linear_w")
        o["label_conditioner.class_to_cond.3.bias"] = ((cdm,), "bias")
    # initial resnet block :756-762
    if cfg.memory_efficient:
        _resnet_specs(o, "init_resnet_block", init_dim, init_dim, tcd, None, cfg.use_global_context_attn)
    # downsampling layers :783-815 (DownsamplingBlock :322-401)
    skip_dims = []
    for i, (din, dout) in enumerate(in_out):
        pre = f"downs.{i}.ds_block"
        last = i >= n - 1
        lcd = cd if cfg.layer_cross_attns[i] else None
        if cfg.memory_efficient:
            o[f"{pre}.0.1.weight"] = ((dout, din * 4, 1, 1), "conv2d_w")                 # Downsample :57-64
            o[f"{pre}.0.1.bias"] = ((dout,), "bias")
            cur = dout
        else:
            cur = din
        _resnet_specs(o, f"{pre}.1", cur, cur, tcd, lcd, False)
        for j in range(cfg.num_resnet_blocks):
            _resnet_specs(o, f"{pre}.2.{j}", cur, cur, tcd, None, cfg.use_global_context_attn)
        if cfg.layer_attns[i]:
            _transformer_specs(o, f"{pre}.3", cur, cfg.layer_attns_depth, cfg.ff_mult, cd)
        if not cfg.memory_efficient:
            if not last:
                o[f"{pre}.4.1.weight"] = ((dout, cur * 4, 1, 1), "conv2d_w")
                o[f"{pre}.4.1.bias"] = ((dout,), "bias")
            else:                                                                          # Parallel(3x3, 1x1) :379-380
                o[f"{pre}.4.fns.0.weight"] = ((dout, din, 3, 3), "conv2d_w")
                o[f"{pre}.4.fns.0.bias"] = ((dout,), "bias")
                o[f"{pre}.4.fns.1.weight"] = ((dout, din, 1, 1), "conv2d_w")
                o[f"{pre}.4.fns.1.bias"] = ((dout,), "bias")
        skip_dims.append(cur)
    # middle :817-823 (MiddleBlock :438-459: its ResnetBlocks get cond_dim, hence an unused cross_attn; its transformer no context)
    mid = dims[-1]
    _resnet_specs(o, "mid_block.mid_block1", mid, mid, tcd, cd, False)
    if cfg.attend_at_middle:
        _transformer_specs(o, "mid_block.mid_attn", mid, cfg.layer_mid_attns_depth, 2, None)
    _resnet_specs(o, "mid_block.mid_block2", mid, mid, tcd, cd, False)
    # upsampling layers :831-851 (UpsamplingBlock :471-522)
    for i, (din, dout) in enumerate(reversed(in_out)):
        li = n - 1 - i
        pre = f"ups.{i}.us_block"
        skip = skip_dims.pop()
        last = i == n - 1
        lcd = cd if cfg.layer_cross_attns[li] else None
        _resnet_specs(o, f"{pre}.0", dout + skip, dout, tcd, lcd, False)
        for j in range(cfg.num_resnet_blocks):
            _resnet_specs(o, f"{pre}.1.{j}", dout + skip, dout, tcd, None, cfg.use_global_context_attn)
        if cfg.layer_attns[li]:
            _transformer_specs(o, f"{pre}.2", dout, cfg.layer_attns_depth, cfg.ff_mult, cd)
        if not last or cfg.memory_efficient:
            if cfg.pixel_shuffle_upsample:                                                 # PixelShuffleUpsample :27-55
                o[f"{pre}.3.net.0.weight"] = ((din * 4, dout, 1, 1), "conv2d_w")
                o[f"{pre}.3.net.0.bias"] = ((din * 4,), "bias")
            else:                                                                          # Upsample :19-25
                o[f"{pre}.3.1.weight"] = ((din, dout, 3, 3), "conv2d_w")
                o[f"{pre}.3.1.bias"] = ((din,), "bias")
    # final :866-872
    if cfg.final_resnet_block:
        _resnet_specs(o, "final_res_block", cfg.dim, cfg.dim, tcd, None, True)
    k = cfg.final_conv_kernel_size
    co = cfg.channels_out if cfg.channels_out is not None else cfg.channels
    o["final_conv.weight"] = ((co, cfg.dim, k, k), "conv2d_w")
    o["final_conv.bias"] = ((co,), "bias")
    return o


def generate_weights(cfg: UNet2dConfig, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Name-keyed deterministic weights (the generator of audiodiffuser_amd/weights.py).  The reference zero-initialises ``final_conv``
    (:874-876): with that a random-init net outputs zeros and a parity check is vacuous, so it is random here too."""
    from audiodiffuser_amd.weights import generate_tensor
    out = OrderedDict()
    for k, (shape, kind) in param_specs(cfg).items():
        if kind == "conv2d_w":
            out[k] = generate_tensor(k, shape, "embed", seed) * (1.0 / (shape[1] * shape[2] * shape[3]) ** 0.5)
        else:
            out[k] = generate_tensor(k, shape, kind, seed)
    return out


# ------------------------------------------------------------------ pieces
def conv2d(p: P, pre: str, x: torch.Tensor, padding: int = 0) -> torch.Tensor:
    return F.conv2d(x, p[f"{pre}.weight"], p[f"{pre}.bias"], padding=padding)


def time_conditioning(p: P, time: torch.Tensor) -> torch.Tensor:
    """LearnedSinusoidalPosEmb :66-81 (x | sin | cos), ``to_time_hiddens`` :702-706 (Linear, SiLU), ``to_time_cond`` :708-710 -> t [B, 4 cond_dim]."""
    x = time[:, None]
    freqs = x * p["to_time_hiddens.0.weights"][None, :] * 2 * math.pi
    f = torch.cat((x, freqs.sin(), freqs.cos()), dim=-1)
    hid = F.silu(F.linear(f, p["to_time_hiddens.1.weight"], p["to_time_hiddens.1.bias"]))
    return F.linear(hid, p["to_time_cond.0.weight"], p["to_time_cond.0.bias"])


def block(p: P, pre: str, x: torch.Tensor, groups: int, scale_shift=None) -> torch.Tensor:
    """Block.forward :96-104: GroupNorm -> x (scale + 1) + shift -> SiLU -> Conv2d 3x3."""
    h = F.group_norm(x, groups, p[f"{pre}.groupnorm.weight"], p[f"{pre}.groupnorm.bias"], 1e-5)
    if scale_shift is not None:
        scale, shift = scale_shift
        h = h * (scale + 1) + shift
    return conv2d(p, f"{pre}.project", F.silu(h), padding=1)


def global_context(p: P, pre: str, x: torch.Tensor) -> torch.Tensor:
    """GlobalContext.forward :190-195: softmax over the positions of a 1-channel key map pools x to [B, C, 1, 1]; 1x1 conv, SiLU, 1x1 conv, sigmoid."""
    b, c = x.shape[:2]
    ctx = conv2d(p, f"{pre}.to_k", x).reshape(b, 1, -1)
    out = torch.einsum("bin,bcn->bci", ctx.softmax(dim=-1), x.reshape(b, c, -1))[..., None]
    h = F.silu(conv2d(p, f"{pre}.net.0", out))
    return torch.sigmoid(conv2d(p, f"{pre}.net.2", h))


def resnet_block(p: P, pre: str, x: torch.Tensor, t: Optional[torch.Tensor], groups: int) -> torch.Tensor:
    """ResnetBlock.forward :147-168 with ``cond=None`` (the cross-attention branch :157-162 is never taken by UNet2dBase.forward, which
    passes ``c=None`` everywhere :928-946): time_mlp -> (scale, shift) for block2 only; gca gate (or the constant 1); 1x1 residual conv
    when the widths differ."""
    scale_shift = None
    if t is not None and f"{pre}.time_mlp.1.weight" in p:
        te = F.linear(F.silu(t), p[f"{pre}.time_mlp.1.weight"], p[f"{pre}.time_mlp.1.bias"])[:, :, None, None]
        scale_shift = te.chunk(2, dim=1)
    h = block(p, f"{pre}.block1", x, groups)
    h = block(p, f"{pre}.block2", h, groups, scale_shift)
    if f"{pre}.gca.to_k.weight" in p:
        h = h * global_context(p, f"{pre}.gca", h)
    res = conv2d(p, f"{pre}.res_conv", x) if f"{pre}.res_conv.weight" in p else x
    return h + res


def layer_norm_g(x: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """layer_utils.py:14-32 (dim = -1, gain only, biased variance, eps 1e-5 in fp32)."""
    var = torch.var(x, dim=-1, unbiased=False, keepdim=True)
    mean = torch.mean(x, dim=-1, keepdim=True)
    return (x - mean) * (var + 1e-5).rsqrt() * g


def attention(p: P, pre: str, x: torch.Tensor, heads: int) -> torch.Tensor:
    """attention_utils.py:113-184, the branch without context (:157-158: no rotary embedding there), no mask, no qk l2norm.  x: [B, N, C]."""
    b, n, c = x.shape
    d = c // heads
    q = F.linear(x, p[f"{pre}.to_q.weight"])
    k, v = F.linear(x, p[f"{pre}.to_kv.weight"]).chunk(2, dim=-1)
    q, k, v = (z.reshape(b, n, heads, d).permute(0, 2, 1, 3) for z in (q, k, v))
    sim = torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5)
    out = torch.matmul(sim.softmax(dim=-1, dtype=torch.float32), v)
    return F.linear(out.permute(0, 2, 1, 3).reshape(b, n, c), p[f"{pre}.to_out.weight"])


def feed_forward(p: P, pre: str, x: torch.Tensor) -> torch.Tensor:
    """attention_utils.py:186-194: LayerNorm -> Linear -> GELU -> LayerNorm -> Linear (no biases)."""
    h = F.linear(layer_norm_g(x, p[f"{pre}.0.g"]), p[f"{pre}.1.weight"])
    return F.linear(layer_norm_g(F.gelu(h), p[f"{pre}.3.g"]), p[f"{pre}.4.weight"])


def transformer_block(p: P, pre: str, x: torch.Tensor, heads: int) -> torch.Tensor:
    """TransformerBlock.forward :219-232 without context: tokens = pixels; per layer x = attn(norm(x)) + x (ONE shared norm), x = ff(x) + x."""
    b, c, hh, ww = x.shape
    tok = x.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
    d = 0
    while f"{pre}.layers.{d}.0.to_q.weight" in p:
        tok = attention(p, f"{pre}.layers.{d}.0", layer_norm_g(tok, p[f"{pre}.norm.g"]), heads) + tok
        tok = feed_forward(p, f"{pre}.layers.{d}.1", tok) + tok
        d += 1
    return tok.reshape(b, hh, ww, c).permute(0, 3, 1, 2)


def downsample(p: P, pre: str, x: torch.Tensor) -> torch.Tensor:
    """Downsample :57-64: 'b c (h s1) (w s2) -> b (c s1 s2) h w' (= pixel_unshuffle by 2), then a 1x1 conv."""
    return conv2d(p, f"{pre}.1", F.pixel_unshuffle(x, 2))


def upsample(p: P, pre: str, x: torch.Tensor, pixel_shuffle: bool) -> torch.Tensor:
    """PixelShuffleUpsample.forward :54-55 (1x1 conv to 4x the channels, SiLU, PixelShuffle(2)) or Upsample :19-25 (nearest x2, 3x3 conv)."""
    if pixel_shuffle:
        return F.pixel_shuffle(F.silu(conv2d(p, f"{pre}.net.0", x)), 2)
    return conv2d(p, f"{pre}.1", F.interpolate(x, scale_factor=2, mode="nearest"), padding=1)


# ------------------------------------------------------------------ the network
def unet2d_forward(p: P, cfg: UNet2dConfig, x: torch.Tensor, time: torch.Tensor, classes: Optional[torch.Tensor] = None,
                   cond_drop_prob: float = 0.0, taps: Optional[dict] = None) -> torch.Tensor:
    """UNet2dBase.forward :879-970 for ``text_embeds=None`` and ``inj_channels=None``.  x: [B, channels, H, W] with H and W multiples of
    2^levels; time: [B] (the EDM wrapper passes c_noise); classes: int64 [B] when ``num_classes != 0``.
    ``taps``: receives the output of every module whose output the generator hooks in the reference ("init_conv", "init_resnet_block",
    "downs.i", "mid_block", "ups.i", "final_res_block")."""
    cfg.check()
    rec = (lambda k, v: taps.__setitem__(k, v)) if taps is not None else (lambda k, v: None)
    g = cfg.resnet_groups
    heads = cfg.attn_heads
    n = len(cfg.in_out)
    # :890-892 initial convolution (CrossEmbedLayer.forward :284-286: one conv per kernel size at stride 1, concatenated)
    if cfg.init_cross_embed:
        ks = sorted(cfg.init_cross_embed_kernel_sizes)
        x = torch.cat([conv2d(p, f"init_conv.convs.{i}", x, padding=(k - 1) // 2) for i, k in enumerate(ks)], dim=1)
    else:
        x = conv2d(p, "init_conv", x, padding=cfg.init_conv_kernel_size // 2)
    rec("init_conv", x)
    # :898-908 conditioning vector
    t = time_conditioning(p, time)
    if cfg.num_classes != 0:
        assert classes is not None
        t = t + label_embedding(p, classes, cond_drop_prob)
    # :918-921
    if cfg.memory_efficient:
        x = resnet_block(p, "init_resnet_block", x, t, g)
        rec("init_resnet_block", x)
    # :924-946 down path (DownsamplingBlock.forward :404-436)
    hiddens: List[torch.Tensor] = []
    for i in range(n):
        pre = f"downs.{i}.ds_block"
        if cfg.memory_efficient:
            x = downsample(p, f"{pre}.0", x)
        x = resnet_block(p, f"{pre}.1", x, t, g)
        for j in range(cfg.num_resnet_blocks):
            x = resnet_block(p, f"{pre}.2.{j}", x, t, g)
            hiddens.append(x)
        if cfg.layer_attns[i]:
            x = transformer_block(p, f"{pre}.3", x, heads)
        hiddens.append(x)
        if not cfg.memory_efficient:
            if i < n - 1:
                x = downsample(p, f"{pre}.4", x)
            else:
                x = conv2d(p, f"{pre}.4.fns.0", x, padding=1) + conv2d(p, f"{pre}.4.fns.1", x)
        rec(f"downs.{i}", x)
    # :948 (MiddleBlock.forward :461-469)
    x = resnet_block(p, "mid_block.mid_block1", x, t, g)
    if cfg.attend_at_middle:
        x = transformer_block(p, "mid_block.mid_attn", x, heads)
    x = resnet_block(p, "mid_block.mid_block2", x, t, g)
    rec("mid_block", x)
    # :950-958 up path (UpsamplingBlock.forward :524-538)
    s = cfg.skip_scale
    for i in range(n):
        li = n - 1 - i
        pre = f"ups.{i}.us_block"
        x = resnet_block(p, f"{pre}.0", torch.cat((x, hiddens.pop() * s), dim=1), t, g)
        for j in range(cfg.num_resnet_blocks):
            x = resnet_block(p, f"{pre}.1.{j}", torch.cat((x, hiddens.pop() * s), dim=1), t, g)
        if cfg.layer_attns[li]:
            x = transformer_block(p, f"{pre}.2", x, heads)
        if i < n - 1 or cfg.memory_efficient:
            x = upsample(p, f"{pre}.3", x, cfg.pixel_shuffle_upsample)
        rec(f"ups.{i}", x)
    assert not hiddens                                                     # :960
    # :969-972
    if cfg.final_resnet_block:
        x = resnet_block(p, "final_res_block", x, t, g)
        rec("final_res_block", x)
    return conv2d(p, "final_conv", x, padding=cfg.final_conv_kernel_size // 2)
