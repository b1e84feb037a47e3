"""arah_release_amd -- MI355X-native (gfx950) implementation of ARAH's articulated-SDF
volume-rendering hot path (reference: taconite/arah-release, im2mesh/metaavatar_render).

Layout
    csrc/        hand-written HIP kernels + the C-ABI (include/arah_hip.h)
    hip.py       ctypes binding of the C-ABI (raw device pointers + HIP stream)
    nets.py      PyTorch parameter containers with the reference's state-dict names
    renderer.py  BodyRayTracing / IDHRNetwork / MetaAvatarRender drop-ins (dict in / dict out)
    config.py    load_config / get_model factories (reference im2mesh/config.py surface)
    synthetic.py seeded synthetic body + input-dict generator
"""

__all__ = ["nets", "synthetic"]
