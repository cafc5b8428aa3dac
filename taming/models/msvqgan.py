"""`taming.models.msvqgan` import path (first_stage_config.target) -> HIP-backed MS-VQGAN."""
from frido_amd.models import VQModelInterface  # noqa: F401
