"""Stale `ldm.models.diffusion.msldm.MSLatentDiffusion` target used by configs/frido/t2i/frido_f16f8_coco_clip.yaml
and layout2i/frido_f8f4_vg.yaml (the module does not exist in the reference) -> FridoDiffusion."""
from frido_amd.models import MSLatentDiffusion  # noqa: F401
