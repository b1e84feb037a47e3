"""``from im2mesh import config`` -> arah_release_amd.config (reference im2mesh/config.py:7-75 surface)."""
from arah_release_amd.config import (load_config, get_model, method_dict, builtin_config,  # noqa: F401
                                     build_synthetic_model)
