"""reference im2mesh/metaavatar/models/__init__.py:3-8 surface (the two decoders ARAH configs use)."""
from arah_release_amd.nets import decoder_dict, HyperBVPNet, Deformer  # noqa: F401
