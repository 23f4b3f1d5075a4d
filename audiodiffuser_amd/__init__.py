"""MI355X-native EDM sampling hot path of AudioDiffuser.

Plugin surface (hydra ``_target_`` s): ``audiodiffuser_amd.UNet1dBase`` / ``audiodiffuser_amd.WaveNetNoise`` (model.net),
``audiodiffuser_amd.EluDiffusion`` (model.diffusion), ``audiodiffuser_amd.EDMSampler`` /
``EDMAlphaSampler`` / ``DPMSampler`` / ``DPM2Sampler`` / ``DPM2MSampler`` / ``ADPM2Sampler`` / ``ADPMPP2SSampler`` / ``LMSSampler`` / ``UniPCSampler`` (model.sampler), ``audiodiffuser_amd.KarrasSchedule``
(model.noise_scheduler).
"""
from .config import UNet1dConfig, config_c1, config_c2, config_c3, config_tiny, config_tiny_cc, PRESETS  # noqa: F401
from .config import WaveNetConfig, config_c5, config_c5_small  # noqa: F401
from .scheduler import KarrasSchedule  # noqa: F401
from .net import UNet1dBase  # noqa: F401
from .wavenet import WaveNetNoise  # noqa: F401
from .adm import UNetModel  # noqa: F401
from .adm_config import ADMConfig, config_c4, config_c4_small  # noqa: F401
from .diffusion import EluDiffusion  # noqa: F401
from .samplers import EDMSampler, EDMAlphaSampler, DPMSampler, DPM2Sampler, DPM2MSampler, ADPM2Sampler, ADPMPP2SSampler, LMSSampler, UniPCSampler  # noqa: F401
