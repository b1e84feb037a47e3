"""``im2mesh.metaavatar_render.lightning_model`` -> arah_release_amd.config (reference lightning_model.py:37-653)."""
from arah_release_amd.config import LightningModel  # noqa: F401
from arah_release_amd.smpl import get_transforms_02v  # noqa: F401
