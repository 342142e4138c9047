from . import jit, _argtools  # noqa: F401
