"""Helpers shared by the golden-vector tests: load the manifest / arrays and rebuild the inputs of
a case from its seed (tests/synth.py), exactly as tests/golden/generate.py made them."""
from __future__ import annotations

import json
import os

import numpy as np

import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

_manifest = None
_arrays = {}


def manifest():
    global _manifest
    if _manifest is None:
        with open(os.path.join(GOLDEN, "manifest.json")) as f:
            _manifest = json.load(f)
    return _manifest


def arrays(name):
    if name not in _arrays:
        _arrays[name] = np.load(os.path.join(GOLDEN, name + ".npz"))
    return _arrays[name]


def metric_of(case) -> np.ndarray:
    shape = (case["n"], case["T"], case["D"])
    if case["kind"] == "normal":
        return synth.normal_like(shape, case["seed"])
    return synth.clustered(shape, case["seed"])


def x_of(case) -> np.ndarray:
    return synth.normal_like((case["n"], case["T"], case["C"]), case["seed"] ^ 0xABCDEF)


def size_of(case):
    if case.get("sizes") == "ints":
        return synth.small_ints((case["n"], case["T"], 1), case["seed"] ^ 0x51235)
    return None


def metric2_of(case, r_eff) -> np.ndarray:
    """second-layer metric of the `source` cases"""
    shape = (case["n"], case["T"] - r_eff, case["D"])
    return synth.normal_like(shape, case["seed"] ^ 0x2222)


def match_cases():
    return manifest()["match"]


def value_cases(op=None):
    return [c for c in manifest()["values"] if op is None or c["op"] == op]


def check_indices(case, got_src, got_dst, got_unm, want_src, want_dst, want_unm):
    """The comparison contract of SURVEY 7.1: src_idx and dst_idx are always bit-exact; unm_idx is
    bit-exact when the fixture certified its order (margins > tau), otherwise it must be the same
    SET of rows (its order is undefined in the reference itself: unstable argsort on near-ties)."""
    cert = case.get("cert", {"src": True, "dst": True, "unm": True})
    assert cert["src"] and cert["dst"], "fixture without certified src/dst"
    np.testing.assert_array_equal(np.asarray(got_src).reshape(want_src.shape), want_src)
    np.testing.assert_array_equal(np.asarray(got_dst).reshape(want_dst.shape), want_dst)
    got_unm = np.asarray(got_unm).reshape(want_unm.shape)
    if cert["unm"]:
        np.testing.assert_array_equal(got_unm, want_unm)
    else:
        np.testing.assert_array_equal(np.sort(got_unm, axis=1), np.sort(want_unm, axis=1))
