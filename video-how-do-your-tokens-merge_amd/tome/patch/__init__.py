"""Same export names as the reference's tome/patch/__init__.py:1-11."""
from .videomae import apply_patch as videomae
from .videomae import apply_duplicate_patch as duplicate_videomae

__all__ = ["videomae", "duplicate_videomae"]
