"""`frido.modules.encoders.modules` import path (cond_stage_config.target) -> HIP-backed BERTEmbedder; CLIP stand-in."""
from frido_amd.models import BERTEmbedder, FrozenCLIPTextEmbedder  # noqa: F401
