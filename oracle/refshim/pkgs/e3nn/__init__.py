"""Stand-in for the ~20 symbols of e3nn 0.4.4 that the reference's nets/ package touches (see oracle/refshim/__init__.py).
Arithmetic is delegated to oracle/e3.py (restated e3nn semantics, pinned by tests/test_independent_kat.py)."""
__version__ = "0.4.4"
from . import o3, math, util, nn  # noqa: F401,E402
