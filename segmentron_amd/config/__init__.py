from .settings import cfg, reset_cfg  # noqa: F401
