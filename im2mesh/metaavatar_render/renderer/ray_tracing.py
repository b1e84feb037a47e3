"""reference renderer/ray_tracing.py:13 surface."""
from arah_release_amd.renderer import BodyRayTracing  # noqa: F401
