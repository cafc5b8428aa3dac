"""`frido.models.diffusion.frido` import path (configs/frido/**/*.yaml `target:`) -> HIP-backed classes."""
from frido_amd.models import FridoDiffusion, DiffusionWrapper  # noqa: F401
