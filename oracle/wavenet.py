"""ORACLE (test infrastructure, not product code) -- the DiffWave-style ``WaveNetNoise`` of BASELINE config 5
(SURVEY.md section 8f row 4), restated on CPU as pure functions over a ``{state_dict key: tensor}`` mapping.
See oracle/unet1d.py for the rules on who may import this package.

Parity status: PINNED against the reference itself (``src/models/backbones/wavenet.py`` imported on CPU in the build
container by ``oracle/gen_golden_next.py``; fixtures in ``tests/golden/next_golden.npz``) at the level the reference can
be called at all, ``WaveNetNoise.forward(audio, diffusion_step)``: the class takes no conditioning input and rejects the
keyword arguments ``Diffusion.denoise_fn`` passes, so there is no reference caller above it.
``wavenet_net`` below is the adapter this build defines for the EDM wrapper (and says so).

All line numbers are in ``src/models/backbones/wavenet.py``.

Arithmetic modes as in oracle/unet1d.py.  ``storage="bf16"`` rounds exactly where the HIP throughput path holds bf16:
the stored stream ``y_n = h_n + e_n`` (the layer input INCLUDING the diffusion-step addend, see ``residual_block``),
the gated activation, the GEMM weights, and the activated skip projection; the skip sum, the diffusion-step
embedding and all accumulations stay fp32.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from audiodiffuser_amd.config import WaveNetConfig, config_c5, config_c5_small
from audiodiffuser_amd.weights import wavenet_param_specs, generate_wavenet_weights
from .unet1d import Storage, FP32, rel_l2

P = Dict[str, torch.Tensor]
Spec = Tuple[Tuple[int, ...], str]


# configuration, state-dict layout and the name-keyed weight generator live in the package (the plugin needs them too)
param_specs = wavenet_param_specs
generate_weights = generate_wavenet_weights


# ------------------------------------------------------------------ pieces
def wn_weight(p: P, pre: str) -> torch.Tensor:
    """:44-51 -- w = v * g / ||v||, the norm over the whole tensor."""
    v = p[f"{pre}.conv.module.weight_v"]
    return v * (p[f"{pre}.conv.module.weight_g"] / torch.norm(v))


def wn_conv(p: P, pre: str, x: torch.Tensor, dilation: int = 1, q: Storage = FP32) -> torch.Tensor:
    """:68-82 -- Conv1d with padding = dilation * (k - 1) / 2 and the weight-normed weight."""
    w = q.r(wn_weight(p, pre))      # not q.w: its cache is keyed by tensor identity and this tensor is a temporary
    k = w.shape[-1]
    return F.conv1d(x, w, p[f"{pre}.conv.module.bias"], dilation=dilation, padding=dilation * (k - 1) // 2)


def diffusion_embedding(step: torch.Tensor, dim_in: int) -> torch.Tensor:
    """:88-92 -- sines first, frequencies exp(-4 i / (half - 1))."""
    half = dim_in // 2
    vec = torch.arange(half)
    table = step.unsqueeze(1) * torch.exp(-vec * 4.0 / (half - 1))
    return torch.cat([torch.sin(table), torch.cos(table)], dim=1)


def step_embedding(p: P, cfg: WaveNetConfig, step: torch.Tensor) -> torch.Tensor:
    """:141-143 -- two Linear layers, swish (x * sigmoid(x), :84-86) after each.  fp32 in both storage modes."""
    e = diffusion_embedding(step, cfg.dim_in)
    e = F.linear(e, p["residual_layer.fc_t1.weight"], p["residual_layer.fc_t1.bias"])
    e = e * torch.sigmoid(e)
    e = F.linear(e, p["residual_layer.fc_t2.weight"], p["residual_layer.fc_t2.bias"])
    return e * torch.sigmoid(e)


def layer_addend(p: P, n: int, emb: torch.Tensor) -> torch.Tensor:
    """:109 -- the per-layer projection of the step embedding, [B, C, 1]."""
    pre = f"residual_layer.residual_blocks.{n}.diffusion_projection"
    return F.linear(emb, p[f"{pre}.weight"], p[f"{pre}.bias"]).unsqueeze(-1)


def wavenet_forward(p: P, cfg: WaveNetConfig, audio: torch.Tensor, step: torch.Tensor,
                    taps: Optional[Dict[str, torch.Tensor]] = None, storage: str = "fp32",
                    force: Optional[Dict[str, torch.Tensor]] = None, errs: Optional[Dict[str, float]] = None) -> torch.Tensor:
    """``WaveNetNoise.forward`` :169-180.  audio: [B, T]; step: [B]; returns [B, 1, T].

    Taps: ``y<n>`` = input of residual layer n including its step addend (``x + diffusion_embed``, :110), ``g<n>`` = the
    gated activation (:113), ``skip`` = the normalised skip sum (:150), ``sp`` = the activated skip projection (:177-178).
    In fp32 storage the arithmetic is the reference's, operation for operation.  In bf16 storage the stream between
    layers is the rounded ``y<n>`` and the layer recovers its residual input as ``y<n> - e<n>``, as the device does."""
    q = Storage(storage)

    def rec(name, v):
        if force is not None and name in force:
            if errs is not None:
                errs[name] = rel_l2(v, force[name])
            v = force[name]
        if taps is not None:
            taps[name] = v
        return v

    emb = step_embedding(p, cfg, step)
    x = F.relu(wn_conv(p, "input_projection", audio.unsqueeze(1)))        # :171-173 (fp32: one input channel)
    skip = 0
    nl = cfg.residual_layers
    if not q.bf16:
        h = x
        for n in range(nl):                                               # :146-150 / :108-116
            pre = f"residual_layer.residual_blocks.{n}"
            y = rec(f"y{n}", h + layer_addend(p, n, emb))
            gate, filt = torch.chunk(wn_conv(p, f"{pre}.dilated_conv", y, cfg.dilation(n)), 2, dim=1)
            g = rec(f"g{n}", torch.sigmoid(gate) * torch.tanh(filt))
            res, sk = torch.chunk(wn_conv(p, f"{pre}.output_projection", g), 2, dim=1)
            h = (h + res) / math.sqrt(2.0)
            skip = skip + sk
    else:
        y = rec("y0", q.r(x + layer_addend(p, 0, emb)))
        for n in range(nl):
            pre = f"residual_layer.residual_blocks.{n}"
            e = layer_addend(p, n, emb)
            gate, filt = torch.chunk(wn_conv(p, f"{pre}.dilated_conv", y, cfg.dilation(n), q), 2, dim=1)
            g = rec(f"g{n}", q.r(torch.sigmoid(gate) * torch.tanh(filt)))
            res, sk = torch.chunk(wn_conv(p, f"{pre}.output_projection", g, 1, q), 2, dim=1)
            skip = skip + sk
            if n + 1 < nl:                 # the last layer's residual output is never used (:148-150)
                y = rec(f"y{n + 1}", q.r(((y - e) + res) / math.sqrt(2.0) + layer_addend(p, n + 1, emb)))
    s = rec("skip", skip * math.sqrt(1.0 / nl))                           # :152
    sp = rec("sp", q.r(F.relu(wn_conv(p, "skip_projection", q.r(s), 1, q))))   # :177-178
    return F.conv1d(sp, p["output_projection.conv.weight"], p["output_projection.conv.bias"])   # :179


def wavenet_net(p: P, cfg: WaveNetConfig, storage: str = "fp32"):
    """The adapter THIS BUILD defines so the EDM wrapper can call the network (no reference counterpart, see the module
    docstring): ``net(x[B, 1, T], t[B], **ignored) -> [B, 1, T]``."""
    def net(x, t, **_ignored):
        return wavenet_forward(p, cfg, x[:, 0], t, storage=storage)
    return net
