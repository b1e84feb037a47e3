"""reference im2mesh/metaavatar_render/config.py:147-302 surface."""
from arah_release_amd.config import get_render_model as get_model  # noqa: F401
