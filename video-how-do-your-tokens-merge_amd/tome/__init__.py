"""Drop-in ``tome`` package (reference: tome/__init__.py) whose merge path runs on hand-written
gfx950 kernels.  The visualisation helpers are imported on first use (they need PIL, which the merge path
does not)."""
from . import merge, patch, utils

_VIS = ("make_visualization", "make_spatial_video_visualization", "make_spatiotemporal_video_visualization",
        "concatenate_images")
__all__ = ["utils", "merge", "patch", *_VIS]


def __getattr__(name):
    if name in _VIS or name == "vis":
        from importlib import import_module
        vis = import_module(__name__ + ".vis")
        return vis if name == "vis" else getattr(vis, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
