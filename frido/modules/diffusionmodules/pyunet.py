"""`frido.modules.diffusionmodules.pyunet` import path (unet_config.target) -> HIP-backed denoiser."""
from frido_amd.models import PyUNetModel  # noqa: F401
