"""`frido.modules.encoders.modules` import path (cond_stage_config.target) -> HIP-backed BERTEmbedder."""
from frido_amd.models import BERTEmbedder  # noqa: F401
