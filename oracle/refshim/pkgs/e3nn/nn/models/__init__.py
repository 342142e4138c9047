from . import v2106  # noqa: F401
