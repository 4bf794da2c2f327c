"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol that
include/tome_hip.h declares, and the host logic that needs no GPU (clamping, workspace sizing, argument
validation, loud failure on CPU tensors) behaves.  No kernel is launched here."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(measurement_build=False):
    """Entry points include/tome_hip.h declares: the product ABI, or (measurement_build) what its
    `#ifdef TOME_PROFILE_HOOKS` section adds for lib/libtome_hip_prof.so."""
    text = open(os.path.join(ROOT, "include", "tome_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    hooks = re.findall(r"#ifdef TOME_PROFILE_HOOKS(.*?)#endif", text, flags=re.S)
    product = re.sub(r"#ifdef TOME_PROFILE_HOOKS.*?#endif", "", text, flags=re.S)
    pick = "".join(hooks) if measurement_build else product
    return sorted(set(re.findall(r"\b(tome_[a-z_]+)\s*\(", pick)))


def test_header_symbols_are_exported():
    from tome import _abi
    L = _abi.lib()
    names = _declared_symbols()
    assert len(names) >= 11
    for name in names:
        assert hasattr(L, name), f"{name} declared in include/tome_hip.h but not exported"
    assert set(names) == set(_abi.SYMBOLS)
    assert L.tome_abi_version() == _abi.ABI_VERSION


def test_measurement_hooks_live_in_their_own_library_only():
    """tome_profile_enable / tome_profile_read (stage events + repeated launches for bench.py's roofline figures) are
    not part of the product: libtome_hip.so does not export them, lib/libtome_hip_prof.so (-DTOME_PROFILE_HOOKS,
    bound by bench.py's stage-timing leg alone) exports them beside the whole product ABI."""
    from tome import _abi
    hooks = _declared_symbols(measurement_build=True)
    assert hooks == ["tome_profile_enable", "tome_profile_read"]
    product = ctypes.CDLL(_abi.LIB_PATH)
    prof = ctypes.CDLL(os.path.join(os.path.dirname(_abi.LIB_PATH), "libtome_hip_prof.so"))
    for name in hooks:
        assert not hasattr(product, name) and hasattr(prof, name), name
        assert name not in _abi.SYMBOLS
    for name in _declared_symbols():
        assert hasattr(prof, name), name


def test_effective_r_matches_reference_clamp():
    from tome import _abi
    for T in range(0, 14):
        for r in (-2, 0, 1, 3, 7, 1000):
            for cls in (False, True):
                for dist in (False, True):
                    want = max(0, min(r, (T - cls - dist) // 2))  # merge.py:36-47
                    assert _abi.effective_r(T, r, cls, dist) == want
                    assert _abi.lib().tome_effective_r(T, r, int(cls), int(dist)) == want


def test_workspace_bytes_monotone_and_aligned():
    from tome import _abi
    L = _abi.lib()
    a = L.tome_match_workspace_bytes(8, 1568, 64)
    b = L.tome_match_workspace_bytes(16, 1568, 64)
    assert a % 256 == 0 and b > a
    assert a >= 4 * 8 * 1568 * 64
    assert L.tome_match_workspace_bytes(0, 10, 10) == 0


def test_argument_validation_without_gpu():
    """Bad arguments are rejected before any launch, with a message."""
    from tome import _abi
    L = _abi.lib()
    rc = L.tome_merge(None, 0, 2, 16, 8, 4, None, None, None, 0, 0, None, None, None)
    assert rc == 1 and b"tome_merge" in L.tome_last_error()
    rc = L.tome_match(None, 0, 2, 16, 8, 128, 8, 4, 0, 0, None, None, None, None, None, None, 0, None)
    assert rc == 1
    buf = ctypes.create_string_buffer(64)
    rc = L.tome_merge_wavg(buf, 7, None, 0, 1, 8, 4, 9, buf, buf, buf, 0, None, buf, buf, None, None)
    assert rc == 1  # r outside (0, T/2]


def test_cpu_tensors_fail_loudly():
    from tome import _abi, merge as tm
    with pytest.raises(_abi.TomeHipError, match="no CPU path"):
        tm.bipartite_soft_matching(torch.randn(1, 8, 4), 2)


def test_missing_library_fails_loudly(monkeypatch):
    from tome import _abi
    monkeypatch.setattr(_abi, "_lib", None)
    monkeypatch.setattr(_abi, "LIB_PATH", "/nonexistent/libtome_hip.so")
    with pytest.raises(_abi.TomeHipError, match="no fallback"):
        _abi.lib()


def test_parse_r_golden(golden_dir):
    import json
    from tome.utils import parse_r
    for e in json.load(open(os.path.join(golden_dir, "parse_r.json"))):
        r = tuple(e["r"]) if e["r_type"] == "tuple" else e["r"]
        assert parse_r(e["num_layers"], r) == e["out"]


def test_product_never_imports_oracle():
    """The product package must not reference the oracle (it is test infrastructure)."""
    pkg = os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M), (dirpath, f)
