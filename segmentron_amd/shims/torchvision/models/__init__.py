from . import _utils  # noqa: F401
