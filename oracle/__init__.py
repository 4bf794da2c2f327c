"""CPU oracle for the ToMe merge hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this package, and only as the checker.  The shipped product
(``video-how-do-your-tokens-merge_amd/``) never imports it and has no CPU fallback.

Two restatements live here:

* ``tome_oracle.c`` (loaded below through ctypes): fixed-order fp32 arithmetic that the HIP
  kernels must reproduce bit for bit.  Follows ``tome/merge.py:17-102,215-352,355-369`` of the
  reference; every function cites its lines in the C source.
* ``torch_port.py``: the same path as the PyTorch-CPU op sequence the reference executes; it is
  what ``bench.py`` times as the ``cpu_baseline`` ("port").

Parity pinning: both are checked against golden vectors produced from the real reference by
``tests/golden/generate.py`` (see ``tests/test_oracle_golden.py``).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libtome_oracle.so")

MODES = {"sum": 0, "mean": 1, "amax": 2, "max": 2, "prod": 3, "amin": 4, "min": 4}


def build(force: bool = False) -> str:
    """Compile tome_oracle.c with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "tome_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-s", "-B"], check=True)
    return _LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        i64, i32, vp = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p
        L.oracle_effective_r.restype = i64
        L.oracle_effective_r.argtypes = [i64, i64, i32, i32]
        L.oracle_head_mean.restype = None
        L.oracle_head_mean.argtypes = [vp, i64, i64, i64, i64, vp]
        L.oracle_match.restype = i64
        L.oracle_match.argtypes = [vp, i64, i64, i64, i64, i32, i32, vp, vp, vp, vp, vp]
        L.oracle_match_scores.restype = i64
        L.oracle_match_scores.argtypes = [vp, i64, i64, i64, i32, i32, vp, vp, vp, vp]
        L.oracle_merge.restype = None
        L.oracle_merge.argtypes = [vp, i64, i64, i64, i64, vp, vp, vp, i32, i32, vp, vp]
        L.oracle_merge_wavg.restype = None
        L.oracle_merge_wavg.argtypes = [vp, vp, i64, i64, i64, i64, vp, vp, vp, i32, vp, vp, vp]
        L.oracle_unmerge.restype = None
        L.oracle_unmerge.argtypes = [vp, i64, i64, i64, i64, vp, vp, vp, vp]
        L.oracle_drop.restype = None
        L.oracle_drop.argtypes = [vp, i64, i64, i64, i64, vp, i32, vp]
        _lib = L
    return _lib


def _f32(a) -> np.ndarray:
    if hasattr(a, "detach"):  # torch tensor
        a = a.detach().to("cpu").float().numpy()
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a) -> np.ndarray:
    if hasattr(a, "detach"):
        a = a.detach().to("cpu").numpy()
    return np.ascontiguousarray(a, dtype=np.int64)


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def effective_r(T: int, r: int, class_token: bool = False, distill_token: bool = False) -> int:
    return int(lib().oracle_effective_r(T, r, int(class_token), int(distill_token)))


def head_mean(keys) -> np.ndarray:
    """[n,H,T,D] -> [n,T,D]: fp32 sum over heads in order, divided by H (torch CPU `k.mean(1)`)."""
    k = _f32(keys)
    n, H, T, D = k.shape
    out = np.empty((n, T, D), np.float32)
    lib().oracle_head_mean(_p(k), n, H, T, D, _p(out))
    return out


class Plan:
    """Index tensors of one bipartite matching (merge.py:64-73), numpy int64, shapes as the
    reference's closure variables: src_idx/dst_idx [n,r,1], unm_idx [n,T1-r,1]."""

    def __init__(self, n, T, r, src, dst, unm, node_max, node_idx, class_token, distill_token):
        self.n, self.T, self.r = n, T, r
        self.src_idx, self.dst_idx, self.unm_idx = src, dst, unm
        self.node_max, self.node_idx = node_max, node_idx
        self.class_token, self.distill_token = class_token, distill_token
        self.edge_keep = None  # hybrid: uint8 [n,r]

    def with_threshold(self, threshold: float) -> "Plan":
        """merge.py:326 -- flags (sorted node_max >= threshold) of the first r edges."""
        p = Plan(self.n, self.T, self.r, self.src_idx, self.dst_idx, self.unm_idx, self.node_max,
                 self.node_idx, self.class_token, self.distill_token)
        nm = np.take_along_axis(self.node_max, self.src_idx[..., 0], axis=1)
        p.edge_keep = np.ascontiguousarray((nm >= np.float32(threshold)).astype(np.uint8))
        return p


def match(metric, r: int, class_token: bool = False, distill_token: bool = False) -> Optional[Plan]:
    m = _f32(metric)
    n, T, D = m.shape
    re = effective_r(T, r, class_token, distill_token)
    if re <= 0:
        return None
    T1 = (T + 1) // 2
    src = np.empty((n, re, 1), np.int64)
    dst = np.empty((n, re, 1), np.int64)
    unm = np.empty((n, T1 - re, 1), np.int64)
    nmax = np.empty((n, T1), np.float32)
    nidx = np.empty((n, T1), np.int32)
    got = lib().oracle_match(_p(m), n, T, D, r, int(class_token), int(distill_token), _p(src), _p(dst),
                             _p(unm), _p(nmax), _p(nidx))
    assert got == re
    return Plan(n, T, re, src, dst, unm, nmax, nidx, class_token, distill_token)


def match_scores(scores, T: int, r: int, class_token: bool = False, distill_token: bool = False) -> Optional[Plan]:
    s = _f32(scores)
    n, T1, T2 = s.shape
    assert T1 == (T + 1) // 2 and T2 == T // 2
    re = effective_r(T, r, class_token, distill_token)
    if re <= 0:
        return None
    src = np.empty((n, re, 1), np.int64)
    dst = np.empty((n, re, 1), np.int64)
    unm = np.empty((n, T1 - re, 1), np.int64)
    nmax = np.empty((n, T1), np.float32)
    lib().oracle_match_scores(_p(s), n, T, r, int(class_token), int(distill_token), _p(src), _p(dst), _p(unm),
                              _p(nmax))
    return Plan(n, T, re, src, dst, unm, nmax, None, class_token, distill_token)


def merge(plan: Plan, x, mode: str = "mean") -> np.ndarray:
    xf = _f32(x)
    n, T, C = xf.shape
    assert (n, T) == (plan.n, plan.T)
    out = np.empty((n, T - plan.r, C), np.float32)
    lib().oracle_merge(_p(xf), n, T, C, plan.r, _p(plan.src_idx), _p(plan.dst_idx), _p(plan.unm_idx),
                       int(plan.distill_token), MODES[mode], _p(plan.edge_keep), _p(out))
    return out


def merge_wavg(plan: Plan, x, size=None) -> Tuple[np.ndarray, np.ndarray]:
    xf = _f32(x)
    n, T, C = xf.shape
    assert (n, T) == (plan.n, plan.T)
    sf = None if size is None else _f32(size).reshape(n, T)
    xo = np.empty((n, T - plan.r, C), np.float32)
    so = np.empty((n, T - plan.r, 1), np.float32)
    lib().oracle_merge_wavg(_p(xf), _p(sf), n, T, C, plan.r, _p(plan.src_idx), _p(plan.dst_idx),
                            _p(plan.unm_idx), int(plan.distill_token), _p(plan.edge_keep), _p(xo), _p(so))
    return xo, so


def unmerge(plan: Plan, x) -> np.ndarray:
    xf = _f32(x)
    n, To, C = xf.shape
    assert To == plan.T - plan.r
    out = np.empty((n, plan.T, C), np.float32)
    lib().oracle_unmerge(_p(xf), n, plan.T, C, plan.r, _p(plan.src_idx), _p(plan.dst_idx), _p(plan.unm_idx),
                         _p(out))
    return out


def drop(plan: Plan, x) -> np.ndarray:
    xf = _f32(x)
    n, T, C = xf.shape
    out = np.empty((n, T - plan.r, C), np.float32)
    lib().oracle_drop(_p(xf), n, T, C, plan.r, _p(plan.unm_idx), int(plan.distill_token), _p(out))
    return out
