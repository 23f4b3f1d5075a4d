"""MI355X-native EDM sampling hot path of AudioDiffuser."""
from .config import UNet1dConfig, config_c1, config_c2, config_c3, config_tiny, PRESETS  # noqa: F401
