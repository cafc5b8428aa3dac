"""`frido.models.diffusion.plms` import path (scripts/sample_diffusion.py) -> HIP-backed sampler."""
from frido_amd.samplers import PLMSSampler  # noqa: F401
