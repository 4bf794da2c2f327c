"""Drop-in ``tome`` package (reference: tome/__init__.py) whose merge path runs on hand-written
gfx950 kernels.  ``tome.vis`` (offline CPU plotting) is out of scope of this build."""
from . import merge, patch, utils

__all__ = ["utils", "merge", "patch"]
