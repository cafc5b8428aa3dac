"""`frido.util` import path of the reference (frido/util.py:74-95) -> frido_amd factory."""
from frido_amd.models import instantiate_from_config, instantiate_from_config_main, get_obj_from_str  # noqa: F401
