"""reference renderer/implicit_differentiable_renderer.py:15 surface."""
from arah_release_amd.renderer import IDHRNetwork  # noqa: F401
