"""Stale `ldm.modules.encoders.modules.BERTEmbedder` target of configs/frido/layout2i/frido_f8f4_vg.yaml:80 -> BERTEmbedder."""
from frido_amd.models import BERTEmbedder  # noqa: F401
