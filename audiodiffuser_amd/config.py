"""Hyper-parameters of the 1-D waveform U-Net on the sampling hot path.

Mirrors the constructor kwargs of the reference ``UNet1dBase`` / ``UNet1d``
(reference: src/models/backbones/unet1d.py:624-649, :818-862) restricted to the
unconditional EDM path.  The presets follow the only
shipped 1-D experiment config
(reference: configs/experiment/sc09/reflowunet_sc09_cfg.yaml:30-50) with
``channels`` / ``num_filters`` scaled as BASELINE.json's configs say.
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict
from typing import List, Optional, Sequence


@dataclass
class UNet1dConfig:
    channels: int = 16
    num_filters: int = 16
    window_length: int = 8
    stride: int = 2
    in_channels: int = 1
    out_channels: int = 1
    resnet_groups: int = 8
    kernel_multiplier_downsample: int = 2
    multipliers: List[int] = field(default_factory=lambda: [1, 2, 2, 4, 4, 4, 4])
    factors: List[int] = field(default_factory=lambda: [2, 2, 2, 4, 4, 4])
    num_blocks: List[int] = field(default_factory=lambda: [2, 2, 2, 2, 2, 2])
    attentions: List[bool] = field(default_factory=lambda: [False, False, False, True, True, True])
    attention_heads: int = 8
    attention_multiplier: int = 2
    use_nearest_upsample: bool = False
    use_skip_scale: bool = True
    use_attention_bottleneck: bool = True
    cond_drop_prob: float = 0.0
    # class conditioning for classifier-free guidance (reference: unet1d.py:824-847, conditioner.py:59-111)
    class_cond: bool = False
    num_classes: Optional[int] = None

    # ---- derived -----------------------------------------------------
    @property
    def num_layers(self) -> int:
        return len(self.multipliers) - 1

    @property
    def time_embed_dim(self) -> int:
        return self.channels * 4

    @property
    def classes_dim(self) -> int:
        """Width of the class embedding that joins the time embedding in every FiLM projection (0 = unconditional)."""
        return self.channels * 4 if self.class_cond else 0

    @property
    def total_downsample(self) -> int:
        f = self.stride
        for x in self.factors:
            f *= x
        return f

    def validate(self) -> None:
        n = self.num_layers
        if not (len(self.factors) == n and len(self.num_blocks) == n and len(self.attentions) == n):
            raise ValueError("factors / num_blocks / attentions must have len(multipliers)-1 entries")
        if self.use_nearest_upsample and any(f < 2 for f in self.factors):
            raise NotImplementedError("use_nearest_upsample=True with a factor of 1 (a plain conv in the reference, unet1d.py:231-234) is not built")
        if self.num_filters != self.channels * self.multipliers[0]:
            raise ValueError("num_filters must equal channels*multipliers[0] (to_in feeds downsamples[0])")
        if self.channels % 2:
            raise ValueError("channels must be even (LearnedPositionalEmbedding)")
        if self.class_cond and not (isinstance(self.num_classes, int) and self.num_classes > 0):
            raise ValueError("class_cond=True needs num_classes (label conditioning; class_embed_dim inputs are outside the path)")

    def to_kwargs(self) -> dict:
        """kwargs accepted by the reference ``UNet1dBase`` constructor."""
        d = asdict(self)
        d.pop("out_channels")
        if not self.class_cond:
            d.pop("class_cond"); d.pop("num_classes")
        return d


def config_c1() -> UNet1dConfig:
    """BASELINE config 1: 16-channel net (1,510,040 parameters)."""
    return UNet1dConfig(channels=16, num_filters=16)


def config_c2() -> UNet1dConfig:
    """BASELINE config 2: 64-channel net (23,937,632 parameters)."""
    return UNet1dConfig(channels=64, num_filters=64)


def config_c3() -> UNet1dConfig:
    """BASELINE config 3: config 2 with attention from the 16x-downsampled level."""
    return UNet1dConfig(channels=64, num_filters=64,
                        attentions=[False, False, True, True, True, True])


def config_tiny() -> UNet1dConfig:
    """Small 3-level net used by unit tests (total downsample 2*2*4*4 = 64)."""
    return UNet1dConfig(channels=16, num_filters=16,
                        multipliers=[1, 2, 4, 4], factors=[2, 4, 4],
                        num_blocks=[1, 2, 1], attentions=[False, True, True])


def config_tiny_cc() -> UNet1dConfig:
    """``config_tiny`` with class conditioning (10 labels, the sc09 digit count) for the CFG tests."""
    c = config_tiny()
    c.class_cond, c.num_classes = True, 10
    return c


def config_tiny_nearest() -> UNet1dConfig:
    """``config_tiny`` with ``use_nearest_upsample=True`` (reference: unet1d.py:236-246: nearest upsampling, reflection pad, 3-tap conv)."""
    c = config_tiny()
    c.use_nearest_upsample = True
    return c


PRESETS = {"c1": config_c1, "c2": config_c2, "c3": config_c3, "tiny": config_tiny, "tiny_cc": config_tiny_cc, "tiny_nearest": config_tiny_nearest}


@dataclass
class WaveNetConfig:
    """Constructor arguments of the reference ``WaveNetNoise`` (src/models/backbones/wavenet.py:154-157), same names and
    defaults.  The embedding widths (128 -> 512 -> 512) are the defaults of ``ResidualGroup`` (:120) and the literal 512 of
    ``ResidualBlock`` (:103)."""
    residual_channels: int = 256
    residual_layers: int = 36
    dilation_cycle: int = 12
    dim_in: int = 128
    dim_mid: int = 512
    dim_out: int = 512
    # what the shared sampler / denoise host code reads from a network config
    in_channels: int = 1
    out_channels: int = 1
    class_cond: bool = False

    def to_kwargs(self) -> dict:
        return dict(residual_channels=self.residual_channels, residual_layers=self.residual_layers,
                    dilation_cycle=self.dilation_cycle)

    def dilation(self, n: int) -> int:
        """wavenet.py:131-134"""
        return 2 ** (n % self.dilation_cycle)


def config_c5() -> WaveNetConfig:
    """BASELINE configs[4]: the reference's default WaveNetNoise (36 layers, 256 channels, dilation cycle 12)."""
    return WaveNetConfig()


def config_c5_small() -> WaveNetConfig:
    return WaveNetConfig(residual_channels=32, residual_layers=6, dilation_cycle=3)
