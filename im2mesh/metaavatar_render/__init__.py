from im2mesh.metaavatar_render import config, models  # noqa: F401
