"""The eight entry points a driver picks from (`tome.patch.<family>` installs the patch on a model,
`tome.patch.duplicate_<family>` inserts attend-and-merge-only copies of a layer) -- the public names of the
reference's tome/patch package, which tools/test_net.py:259-283 and slowfast/utils/model_benchmark.py:82-103 select by
MODEL.MODEL_NAME."""
from importlib import import_module

FAMILIES = ("videomae", "timesformer", "motionformer", "vivit")
__all__ = []
for _family in FAMILIES:
    _mod = import_module(f"{__name__}.{_family}")
    _exports = {_family: _mod.apply_patch, f"duplicate_{_family}": _mod.apply_duplicate_patch}
    # the family names are rebound from the submodules to the functions, as the reference's package does
    globals().update(_exports)
    __all__ += list(_exports)
del _family, _mod, _exports
