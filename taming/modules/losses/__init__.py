"""`taming.modules.losses.DummyLoss` (lossconfig.target of every shipped first-stage config)."""
from frido_amd.models import DummyLoss  # noqa: F401
