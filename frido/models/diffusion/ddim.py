"""`frido.models.diffusion.ddim` import path (scripts/sample_diffusion.py) -> HIP-backed sampler."""
from frido_amd.samplers import DDIMSampler  # noqa: F401
