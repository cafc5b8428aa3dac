"""Stale `ldm.modules.diffusionmodules.openaimodel.UNetModel` target of two shipped configs -> PyUNetModel."""
from frido_amd.models import UNetModel  # noqa: F401
