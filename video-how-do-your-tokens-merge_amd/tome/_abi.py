"""ctypes binding of the C ABI in include/tome_hip.h (lib/libtome_hip.so, gfx950 only).

There is no CPU path and no PyTorch fallback: if the shared library is missing or a tensor is not
on a HIP device, the call raises.  PyTorch is used for device memory and streams only.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("TOME_HIP_LIB", os.path.join(_PKG, "lib", "libtome_hip.so"))

SYMBOLS = (
    "tome_abi_version", "tome_last_error", "tome_effective_r", "tome_match_workspace_bytes", "tome_match",
    "tome_match_keys",
    "tome_match_scores", "tome_edge_keep", "tome_merge_wavg", "tome_merge_wavg_ln", "tome_merge_wavg_regrouped",
    "tome_merge_wavg_regrouped_ln", "tome_add_layernorm", "tome_add_layernorm_skip_first", "tome_add_layernorm_regrouped", "tome_prop_attention", "tome_prop_attention_segments", "tome_trajectory_mix", "tome_short_attention", "tome_merge",
    "tome_drop",
    "tome_drop_regrouped",
    "tome_unmerge", "tome_row_map", "tome_source_init", "tome_gelu_erf", "tome_tubelet_rows",
)

ABI_VERSION = 9
DTYPES = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}
MODES = {"sum": 0, "mean": 1, "amax": 2, "max": 2, "prod": 3, "amin": 4, "min": 4}

_lib = None


class TomeHipError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """Load libtome_hip.so once; fail loudly when it is absent (build it with
    `python video-how-do-your-tokens-merge_amd/csrc/build.py`)."""
    global _lib
    if _lib is not None:
        return _lib
    _lib = bind(LIB_PATH)
    return _lib


def bind(path: str) -> ctypes.CDLL:
    """A library file with the C ABI of include/tome_hip.h, loaded and typed.  `lib()` binds the product library once;
    bench.py's stage-timing leg binds the measurement build beside it (lib/libtome_hip_prof.so)."""
    if not os.path.exists(path):
        raise TomeHipError(
            f"HIP extension not found at {path}: the MI355X merge path has no fallback. "
            "Build it with `python video-how-do-your-tokens-merge_amd/csrc/build.py` (needs hipcc).")
    L = ctypes.CDLL(path)
    i64, i32, vp, sz = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t
    L.tome_abi_version.restype = i32
    L.tome_abi_version.argtypes = []
    L.tome_last_error.restype = ctypes.c_char_p
    L.tome_last_error.argtypes = []
    L.tome_effective_r.restype = i64
    L.tome_effective_r.argtypes = [i64, i64, i32, i32]
    L.tome_match_workspace_bytes.restype = sz
    L.tome_match_workspace_bytes.argtypes = [i64, i64, i64]
    L.tome_match.restype = i32
    L.tome_match.argtypes = [vp, i32, i64, i64, i64, i64, i64, i64, i32, i32, vp, vp, vp, vp, vp, vp, sz, vp]
    L.tome_match_keys.restype = i32
    L.tome_match_keys.argtypes = [vp, i32, i64, i64, i64, i64, i64, i64, i64, i64, i64, i64, i32, i32, vp, vp, vp, vp,
                                  vp, vp, sz, vp]
    L.tome_match_scores.restype = i32
    L.tome_match_scores.argtypes = [vp, i64, i64, i64, i32, i32, vp, vp, vp, vp, vp, vp, sz, vp]
    L.tome_edge_keep.restype = i32
    L.tome_edge_keep.argtypes = [vp, vp, i64, i64, i64, ctypes.c_float, vp, vp]
    L.tome_merge_wavg.restype = i32
    L.tome_merge_wavg.argtypes = [vp, i32, vp, i32, i64, i64, i64, i64, vp, vp, vp, i32, vp, vp, vp, vp, vp]
    L.tome_merge_wavg_ln.restype = i32
    L.tome_merge_wavg_ln.argtypes = [vp, i32, vp, i32, i64, i64, i64, i64, vp, vp, vp, i32, vp, vp, vp, ctypes.c_float,
                                     vp, vp, vp, vp, vp, vp, vp]
    L.tome_merge_wavg_regrouped.restype = i32
    L.tome_merge_wavg_regrouped.argtypes = [vp, i32, vp, i32, i64, i64, i64, i64, i64, i32, vp, vp, vp, vp, vp, vp, vp,
                                            vp]
    L.tome_merge_wavg_regrouped_ln.restype = i32
    L.tome_merge_wavg_regrouped_ln.argtypes = [vp, i32, vp, i32, i64, i64, i64, i64, i64, i32, vp, vp, vp, vp, vp, vp,
                                               ctypes.c_float, vp, i32, vp, vp, vp, vp, vp, vp, vp]
    L.tome_add_layernorm.restype = i32
    L.tome_add_layernorm.argtypes = [vp, vp, i32, i64, i64, vp, vp, ctypes.c_float, vp, vp, vp]
    L.tome_add_layernorm_skip_first.restype = i32
    L.tome_add_layernorm_skip_first.argtypes = [vp, vp, i32, i64, i64, i64, vp, vp, ctypes.c_float, vp, vp, vp]
    L.tome_add_layernorm_regrouped.restype = i32
    L.tome_add_layernorm_regrouped.argtypes = [vp, vp, i32, i64, i64, i64, i64, vp, vp, ctypes.c_float, vp, vp, vp]
    L.tome_merge.restype = i32
    L.tome_merge.argtypes = [vp, i32, i64, i64, i64, i64, vp, vp, vp, i32, i32, vp, vp, vp]
    L.tome_prop_attention.restype = i32
    L.tome_prop_attention.argtypes = [vp, vp, vp, i32, i64, i64, i64, i64, i64, vp, vp, vp, vp, i64, i32, ctypes.c_float,
                                      vp, vp, vp]
    L.tome_prop_attention_segments.restype = i32
    L.tome_prop_attention_segments.argtypes = [vp, vp, vp, i32, i64, i64, i64, i64, i64, vp, vp, vp, vp, i64,
                                               ctypes.c_float, vp, vp, i64, vp, vp]
    L.tome_short_attention.restype = i32
    L.tome_short_attention.argtypes = [vp, vp, vp, i32, i64, i64, i64, i64, vp, vp, vp, ctypes.c_float, vp, vp]
    L.tome_trajectory_mix.restype = i32
    L.tome_trajectory_mix.argtypes = [vp, vp, vp, i32, i64, i64, i64, i64, i64, i64, i64, ctypes.c_float, vp, i64, vp, vp]
    L.tome_drop_regrouped.restype = i32
    L.tome_drop_regrouped.argtypes = [vp, i32, i64, i64, i64, i64, i64, i32, vp, vp, vp]
    L.tome_drop.restype = i32
    L.tome_drop.argtypes = [vp, i32, i64, i64, i64, i64, vp, i32, vp, vp]
    L.tome_unmerge.restype = i32
    L.tome_unmerge.argtypes = [vp, i32, i64, i64, i64, i64, vp, vp, vp, vp, vp]
    L.tome_gelu_erf.restype = i32
    L.tome_gelu_erf.argtypes = [vp, i32, i64, vp, vp]
    L.tome_tubelet_rows.restype = i32
    L.tome_tubelet_rows.argtypes = [vp, i32, i64, i64, i64, i64, i64, vp, i64, i64, i64, vp, vp]
    L.tome_row_map.restype = i32
    L.tome_row_map.argtypes = [i64, i64, i64, i32, vp, vp, vp, vp, vp]
    L.tome_source_init.restype = i32
    L.tome_source_init.argtypes = [i64, i64, i64, i32, i32, vp, vp, vp]
    if L.tome_abi_version() != ABI_VERSION:
        raise TomeHipError(f"{os.path.basename(path)} ABI {L.tome_abi_version()} != expected {ABI_VERSION}")
    return L


def _check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().tome_last_error().decode("utf-8", "replace")
        raise TomeHipError(f"{what} failed (status {rc}): {msg}")


def require_device(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise TomeHipError(
            f"{what}: tensor is on {t.device}; this build runs only on a HIP device (MI355X) and has no CPU path")


def dtype_code(t: torch.Tensor, what: str) -> int:
    try:
        return DTYPES[t.dtype]
    except KeyError:
        raise TomeHipError(f"{what}: dtype {t.dtype} not supported (float32, bfloat16, float16)") from None


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(device) -> int:
    """The caller's current HIP stream on `device` as the raw handle the C ABI takes.  torch's own raw accessor (what
    its compiled-graph runtime calls per kernel) when it exists: `torch.cuda.current_stream(device).cuda_stream` builds
    a Stream object per call, 4-5 us of the ~20 us a wrapper of this module costs at the reference's batch of 8."""
    if _raw_stream is not None:
        idx = device.index
        return _raw_stream(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(device).cuda_stream


class _on_device:
    """`with torch.cuda.device(dev)` without its cost when `dev` already is the current device (the
    common case: one process per GPU)."""

    __slots__ = ("idx", "prev")

    def __init__(self, device):
        self.idx = device.index

    def __enter__(self):
        self.prev = torch.cuda.current_device()
        if self.prev != self.idx:
            torch.cuda.set_device(self.idx)

    def __exit__(self, *exc):
        if self.prev != self.idx:
            torch.cuda.set_device(self.prev)
        return False


def _workspace(device, stream: int, nbytes: int) -> torch.Tensor:
    """Scratch of one matching, taken from PyTorch's caching allocator per call: the allocator hands a block back
    only to later work of the SAME stream (or, under graph capture, keeps it inside that graph's private pool),
    so the kernels still in flight when this tensor is released can never share it with another stream's or
    another graph's matching -- which a process-wide cache keyed by stream handle could not guarantee."""
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def effective_r(T: int, r: int, class_token: bool, distill_token: bool) -> int:
    """merge.py:36-47 (same clamp as tome_effective_r; tests hold the two together)."""
    return max(0, min(int(r), (int(T) - int(bool(class_token)) - int(bool(distill_token))) // 2))


class MatchPlan:
    """Device-resident result of one matching: the reference's closure variables (int64,
    [n,r,1] / [n,T1-r,1]) plus what the fused kernels want (node_max for the hybrid threshold,
    row_map for source tracking)."""

    __slots__ = ("n", "T", "r", "class_token", "distill_token", "src_idx", "dst_idx", "unm_idx", "node_max",
                 "row_map", "edge_keep", "device")

    def __init__(self, n, T, r, class_token, distill_token, src_idx, dst_idx, unm_idx, node_max, row_map, device):
        self.n, self.T, self.r = n, T, r
        self.class_token, self.distill_token = bool(class_token), bool(distill_token)
        self.src_idx, self.dst_idx, self.unm_idx = src_idx, dst_idx, unm_idx
        self.node_max, self.row_map = node_max, row_map
        self.edge_keep = None
        self.device = device


def _alloc_plan(n, T, re, class_token, distill_token, device, want_node_max, want_row_map):
    T1 = (T + 1) // 2
    src = torch.empty((n, re, 1), dtype=torch.int64, device=device)
    dst = torch.empty((n, re, 1), dtype=torch.int64, device=device)
    unm = torch.empty((n, T1 - re, 1), dtype=torch.int64, device=device)
    nmax = torch.empty((n, T1), dtype=torch.float32, device=device) if want_node_max else None
    rmap = torch.empty((n, T1), dtype=torch.int32, device=device) if want_row_map else None
    return MatchPlan(n, T, re, class_token, distill_token, src, dst, unm, nmax, rmap, device)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def match(metric: torch.Tensor, r: int, class_token=False, distill_token=False, want_node_max=False,
          want_row_map=False) -> Optional[MatchPlan]:
    """tome_match on `metric` [n,T,D]; returns None when the clamped r is <= 0."""
    require_device(metric, "bipartite_soft_matching(metric)")
    if metric.dim() != 3:
        raise TomeHipError(f"metric must be [batch, tokens, channels], got {tuple(metric.shape)}")
    code = dtype_code(metric, "metric")
    n, T, D = metric.shape
    re = effective_r(T, r, class_token, distill_token)
    if re <= 0 or n == 0:
        return None
    if metric.stride(2) != 1:
        metric = metric.contiguous()
    L = lib()
    dev = metric.device
    with _on_device(dev):
        st = _stream(dev)
        nbytes = L.tome_match_workspace_bytes(n, T, D)
        ws = _workspace(dev, st, nbytes)
        plan = _alloc_plan(n, T, re, class_token, distill_token, dev, want_node_max, want_row_map)
        rc = L.tome_match(metric.data_ptr(), code, n, T, D, metric.stride(0), metric.stride(1), int(r),
                          int(bool(class_token)), int(bool(distill_token)), plan.src_idx.data_ptr(),
                          plan.dst_idx.data_ptr(), plan.unm_idx.data_ptr(), _ptr(plan.node_max), _ptr(plan.row_map),
                          ws.data_ptr(), ws.numel(), st)
    _check(rc, "tome_match")
    return plan


def keys_fusable(keys: torch.Tensor) -> bool:
    """Can tome_match_keys read these per-head keys in place?  [n, H, T, 64], or [outer, inner, H, T, 64] when the
    groups are interleaved inside a clip (Motionformer's '(s f)' regrouping), any strides with contiguous channels
    and 16-byte aligned rows."""
    if keys.dim() not in (4, 5) or keys.shape[-1] != 64 or keys.stride(-1) != 1 or keys.dtype not in DTYPES \
            or not keys.is_cuda:
        return False
    es = keys.element_size()
    return keys.data_ptr() % 16 == 0 and all((keys.stride(d) * es) % 16 == 0 for d in range(keys.dim() - 1))


def match_keys(keys: torch.Tensor, r: int, class_token=False, distill_token=False, want_node_max=False,
               want_row_map=False, checked: bool = False) -> Optional[MatchPlan]:
    """tome_match_keys on per-head keys [n,H,T,64] or [outer,inner,H,T,64] (group = outer*inner + inner index;
    the metric = keys.mean(heads) is never materialised)."""
    if not checked and not keys_fusable(keys):
        require_device(keys, "match_keys(keys)")
        raise TomeHipError(f"match_keys: keys {tuple(keys.shape)} strides {keys.stride()} are not readable in place")
    if keys.dim() == 5:
        outer, inner, H, T, D = keys.shape
        n = outer * inner
        s_n, s_in, s_h, s_t = keys.stride()[:4]
    else:
        n, H, T, D = keys.shape
        inner, s_in = 1, 0
        s_n, s_h, s_t = keys.stride()[:3]
    re = effective_r(T, r, class_token, distill_token)
    if re <= 0 or n == 0:
        return None
    L = lib()
    dev = keys.device
    with _on_device(dev):
        st = _stream(dev)
        nbytes = L.tome_match_workspace_bytes(n, T, D)
        ws = _workspace(dev, st, nbytes)
        plan = _alloc_plan(n, T, re, class_token, distill_token, dev, want_node_max, want_row_map)
        rc = L.tome_match_keys(keys.data_ptr(), DTYPES[keys.dtype], n, H, T, D, s_n, inner, s_in, s_h, s_t, int(r),
                               int(bool(class_token)), int(bool(distill_token)),
                               plan.src_idx.data_ptr(), plan.dst_idx.data_ptr(), plan.unm_idx.data_ptr(),
                               _ptr(plan.node_max), _ptr(plan.row_map), ws.data_ptr(), ws.numel(), st)
    _check(rc, "tome_match_keys")
    return plan


def match_scores(scores: torch.Tensor, T: int, r: int, class_token=False, distill_token=False,
                 want_node_max=False, want_row_map=False) -> Optional[MatchPlan]:
    require_device(scores, "match_scores(scores)")
    n, T1, T2 = scores.shape
    if T1 != (T + 1) // 2 or T2 != T // 2:
        raise TomeHipError(f"scores shape {tuple(scores.shape)} does not fit T={T}")
    re = effective_r(T, r, class_token, distill_token)
    if re <= 0 or n == 0:
        return None
    scores = scores.float().contiguous()
    L = lib()
    dev = scores.device
    with _on_device(dev):
        st = _stream(dev)
        nbytes = L.tome_match_workspace_bytes(n, T, 1)
        ws = _workspace(dev, st, nbytes)
        plan = _alloc_plan(n, T, re, class_token, distill_token, dev, want_node_max, want_row_map)
        rc = L.tome_match_scores(scores.data_ptr(), n, T, int(r), int(bool(class_token)), int(bool(distill_token)),
                                 plan.src_idx.data_ptr(), plan.dst_idx.data_ptr(), plan.unm_idx.data_ptr(),
                                 _ptr(plan.node_max), _ptr(plan.row_map), ws.data_ptr(), ws.numel(), st)
    _check(rc, "tome_match_scores")
    return plan


def edge_keep(plan: MatchPlan, threshold: float) -> torch.Tensor:
    if plan.node_max is None:
        raise TomeHipError("edge_keep needs a plan made with want_node_max=True")
    keep = torch.empty((plan.n, plan.r), dtype=torch.uint8, device=plan.device)
    with _on_device(plan.device):
        rc = lib().tome_edge_keep(plan.node_max.data_ptr(), plan.src_idx.data_ptr(), plan.n, plan.T, plan.r,
                                  float(threshold), keep.data_ptr(), _stream(plan.device))
    _check(rc, "tome_edge_keep")
    return keep


def _prep_x(plan: MatchPlan, x: torch.Tensor, what: str, tokens: int) -> torch.Tensor:
    require_device(x, what)
    if x.dim() != 3 or x.shape[0] != plan.n or x.shape[1] != tokens:
        raise TomeHipError(f"{what}: expected [{plan.n}, {tokens}, C], got {tuple(x.shape)}")
    if x.device != plan.device:
        raise TomeHipError(f"{what}: tensor on {x.device}, matching was computed on {plan.device}")
    if torch.is_grad_enabled() and x.requires_grad:
        raise TomeHipError(f"{what}: autograd through the HIP merge kernels is not implemented (inference path); "
                           "call under torch.no_grad()")
    return x if x.is_contiguous() else x.contiguous()


def _log_size_like(s_out: torch.Tensor, want: bool) -> Optional[torch.Tensor]:
    """Buffer for log(size') -- the proportional-attention bias of the next block (`size.log()`,
    tome/patch/videomae.py:62-63).  It travels as the attribute `_tome_log` of the size tensor it belongs to, so
    a consumer that finds it (`log_of_size`) can never pair it with another size."""
    if not want:
        return None
    log = torch.empty_like(s_out)
    s_out._tome_log = log
    return log


def log_of_size(size: torch.Tensor) -> torch.Tensor:
    """`size.log()`; free when the merge kernel that produced `size` already emitted it."""
    log = getattr(size, "_tome_log", None)
    return log if log is not None else size.log()


def merge_wavg(plan: MatchPlan, x: torch.Tensor, size: Optional[torch.Tensor], log_size: bool = False):
    x = _prep_x(plan, x, "merge_wavg(x)", plan.T)
    n, T, C = x.shape
    xcode = dtype_code(x, "x")
    if size is not None:
        require_device(size, "merge_wavg(size)")
        if size.shape != (n, T, 1):
            raise TomeHipError(f"size must be [{n}, {T}, 1], got {tuple(size.shape)}")
        if size.dtype not in (x.dtype, torch.float32):
            size = size.to(x.dtype)
        size = size.contiguous()
        sdtype = size.dtype
    else:
        sdtype = x.dtype  # torch.ones_like(x[..., 0, None]) -- merge.py:362-363
    scode = DTYPES[sdtype]
    x_out = torch.empty((n, T - plan.r, C), dtype=x.dtype, device=x.device)
    s_out = torch.empty((n, T - plan.r, 1), dtype=sdtype, device=x.device)
    log = _log_size_like(s_out, log_size)
    with _on_device(x.device):
        rc = lib().tome_merge_wavg(x.data_ptr(), xcode, _ptr(size), scode, n, T, C, plan.r, plan.src_idx.data_ptr(),
                                   plan.dst_idx.data_ptr(), plan.unm_idx.data_ptr(), int(plan.distill_token),
                                   _ptr(plan.edge_keep), x_out.data_ptr(), s_out.data_ptr(), _ptr(log),
                                   _stream(x.device))
    _check(rc, "tome_merge_wavg")
    return x_out, s_out


def ln_fusable(x: torch.Tensor, norm) -> bool:
    """Can tome_merge_wavg_ln produce norm(x') for this LayerNorm module?"""
    C = x.shape[-1]
    return (isinstance(norm, torch.nn.LayerNorm) and norm.elementwise_affine and norm.bias is not None
            and tuple(norm.normalized_shape) == (C,) and x.dtype in (torch.bfloat16, torch.float16)
            and norm.weight.dtype == x.dtype and C % 8 == 0 and C <= 1024 and x.is_cuda
            and not (torch.is_grad_enabled() and (x.requires_grad or norm.weight.requires_grad)))


def _out_bias(out_bias, x, C):
    if out_bias is None:
        return None
    if out_bias.numel() != C or out_bias.dtype != x.dtype or out_bias.device != x.device:
        raise TomeHipError(f"out_bias must hold {C} values of x's dtype on x's device")
    return out_bias.contiguous()


def merge_wavg_ln(plan: MatchPlan, x: torch.Tensor, size: Optional[torch.Tensor], weight: torch.Tensor,
                  bias: torch.Tensor, eps: float, addend: Optional[torch.Tensor] = None, log_size: bool = False,
                  out_bias: Optional[torch.Tensor] = None):
    """merge_wavg + LayerNorm of the merged tokens in one launch: returns (x_out, y_out, size_out).  With
    `addend` the merged tokens are `x + addend` (the residual in front of the merge, added while loading).
    out_bias [C]: x_out is stored as x' + out_bias (y_out stays LayerNorm(x')) -- for a caller whose next GEMM
    accumulates onto x_out in place."""
    x = _prep_x(plan, x, "merge_wavg_ln(x)", plan.T)
    n, T, C = x.shape
    out_bias = _out_bias(out_bias, x, C)
    if addend is not None:
        if addend.shape != x.shape or addend.dtype != x.dtype or addend.device != x.device:
            raise TomeHipError("merge_wavg_ln: addend must match x in shape, dtype and device")
        addend = addend if addend.is_contiguous() else addend.contiguous()
    xcode = dtype_code(x, "x")
    if size is not None:
        if size.shape != (n, T, 1):
            raise TomeHipError(f"size must be [{n}, {T}, 1], got {tuple(size.shape)}")
        if size.dtype not in (x.dtype, torch.float32):
            size = size.to(x.dtype)
        size = size.contiguous()
        sdtype = size.dtype
    else:
        sdtype = x.dtype
    x_out = torch.empty((n, T - plan.r, C), dtype=x.dtype, device=x.device)
    y_out = torch.empty_like(x_out)
    s_out = torch.empty((n, T - plan.r, 1), dtype=sdtype, device=x.device)
    log = _log_size_like(s_out, log_size)
    with _on_device(x.device):
        rc = lib().tome_merge_wavg_ln(x.data_ptr(), xcode, _ptr(size), DTYPES[sdtype], n, T, C, plan.r,
                                      plan.src_idx.data_ptr(), plan.dst_idx.data_ptr(), plan.unm_idx.data_ptr(),
                                      int(plan.distill_token), _ptr(plan.edge_keep), weight.data_ptr(), bias.data_ptr(),
                                      float(eps), _ptr(addend), x_out.data_ptr(), y_out.data_ptr(), s_out.data_ptr(),
                                      _ptr(log), _ptr(out_bias), _stream(x.device))
    _check(rc, "tome_merge_wavg_ln")
    return x_out, y_out, s_out


def add_layernorm(x: torch.Tensor, addend: Optional[torch.Tensor], weight: torch.Tensor, bias: torch.Tensor, eps: float,
                  skip_first: bool = False):
    """(x + addend, LayerNorm(x + addend)) in one launch, for 16-bit [..., C] tensors (C <= 1024, C % 8 == 0).
    skip_first (x [B, N, C]): the LayerNorm output leaves out every clip's first row (the class token) and is
    [B, N-1, C] -- what TimeSformer's temporal_norm1 consumer reads (`xn[:, 1:]`), as a contiguous tensor."""
    require_device(x, "add_layernorm(x)")
    x = x if x.is_contiguous() else x.contiguous()
    C = x.shape[-1]
    if addend is None:
        # LayerNorm only: x holds the finished sum already (returned as it is)
        if skip_first and (x.dim() != 3 or x.shape[1] < 2):
            raise TomeHipError("add_layernorm(skip_first): x must be [B, N >= 2, C]")
        y_out = (torch.empty((x.shape[0], x.shape[1] - 1, C), dtype=x.dtype, device=x.device) if skip_first
                 else torch.empty_like(x))
        with _on_device(x.device):
            if skip_first:
                rc = lib().tome_add_layernorm_skip_first(x.data_ptr(), None, dtype_code(x, "x"), x.shape[0], x.shape[1],
                                                         C, weight.data_ptr(), bias.data_ptr(), float(eps), None,
                                                         y_out.data_ptr(), _stream(x.device))
            else:
                rc = lib().tome_add_layernorm(x.data_ptr(), None, dtype_code(x, "x"), x.numel() // C, C,
                                              weight.data_ptr(), bias.data_ptr(), float(eps), None, y_out.data_ptr(),
                                              _stream(x.device))
        _check(rc, "tome_add_layernorm")
        return x, y_out
    if addend.shape != x.shape or addend.dtype != x.dtype or addend.device != x.device:
        raise TomeHipError("add_layernorm: addend must match x in shape, dtype and device")
    addend = addend if addend.is_contiguous() else addend.contiguous()
    x_out = torch.empty_like(x)
    if skip_first:
        if x.dim() != 3 or x.shape[1] < 2:
            raise TomeHipError("add_layernorm(skip_first): x must be [B, N >= 2, C]")
        y_out = torch.empty((x.shape[0], x.shape[1] - 1, C), dtype=x.dtype, device=x.device)
        with _on_device(x.device):
            rc = lib().tome_add_layernorm_skip_first(x.data_ptr(), addend.data_ptr(), dtype_code(x, "x"), x.shape[0],
                                                     x.shape[1], C, weight.data_ptr(), bias.data_ptr(), float(eps),
                                                     x_out.data_ptr(), y_out.data_ptr(), _stream(x.device))
        _check(rc, "tome_add_layernorm_skip_first")
        return x_out, y_out
    y_out = torch.empty_like(x)
    with _on_device(x.device):
        rc = lib().tome_add_layernorm(x.data_ptr(), addend.data_ptr(), dtype_code(x, "x"), x.numel() // C, C,
                                      weight.data_ptr(), bias.data_ptr(), float(eps), x_out.data_ptr(),
                                      y_out.data_ptr(), _stream(x.device))
    _check(rc, "tome_add_layernorm")
    return x_out, y_out


def add_layernorm_regrouped(x: torch.Tensor, addend: torch.Tensor, frames: int, weight: torch.Tensor,
                            bias: torch.Tensor, eps: float):
    """TimeSformer's mid-block step in one launch: x [B, 1 + P*F, C] (class token first), addend [B, P*F, C] ->
    (x1, y) with x1 = cat(cls, x[:, 1:] + addend) and y = LayerNorm of the tokens regrouped
    'b (p t) m -> (b t) p m' with the class token in front of every frame: [B*F, 1 + P, C]."""
    require_device(x, "add_layernorm_regrouped(x)")
    B, N, C = x.shape
    F = int(frames)
    P = (N - 1) // F
    if N != 1 + P * F or tuple(addend.shape) != (B, P * F, C) or addend.dtype != x.dtype or addend.device != x.device:
        raise TomeHipError(f"add_layernorm_regrouped: x {tuple(x.shape)} / addend {tuple(addend.shape)} do not hold a "
                           f"class token and {F} frames of tokens")
    x = x if x.is_contiguous() else x.contiguous()
    addend = addend if addend.is_contiguous() else addend.contiguous()
    x_out = torch.empty_like(x)
    y_out = torch.empty((B * F, 1 + P, C), dtype=x.dtype, device=x.device)
    with _on_device(x.device):
        rc = lib().tome_add_layernorm_regrouped(x.data_ptr(), addend.data_ptr(), dtype_code(x, "x"), B, F, P, C,
                                                weight.data_ptr(), bias.data_ptr(), float(eps), x_out.data_ptr(),
                                                y_out.data_ptr(), _stream(x.device))
    _check(rc, "tome_add_layernorm_regrouped")
    return x_out, y_out


def merge_wavg_regrouped(plan: MatchPlan, x_full: torch.Tensor, size: Optional[torch.Tensor], frames: int,
                         has_cls: bool = True, ln=None, addend: Optional[torch.Tensor] = None,
                         log_size: bool = False, addend_grouped: Optional[torch.Tensor] = None,
                         cls_addend: Optional[torch.Tensor] = None, out_bias: Optional[torch.Tensor] = None):
    """merge_wavg on the interleaved layout of TimeSformer / Motionformer: x_full [B, has_cls + P*F, C] whose
    token has_cls + p*F + f belongs to group b*F + f; returns x_out [B, has_cls + (P-r)*F, C] and size
    [B*F, P-r, 1].  Replaces rearrange -> merge_wavg -> rearrange -> cat (timesformer.py:89-107).
    With ln=(weight, bias, eps) it also returns y_out = LayerNorm(x_out) (class-token rows included) between the
    two, and `addend` (same shape as x_full) is added to the tokens while they are loaded."""
    require_device(x_full, "merge_wavg_regrouped(x)")
    if x_full.dim() != 3:
        raise TomeHipError(f"merge_wavg_regrouped: x must be [B, tokens, C], got {tuple(x_full.shape)}")
    B, N, C = x_full.shape
    cls = 1 if has_cls else 0
    F, P = int(frames), plan.T
    if N != cls + P * F or plan.n != B * F:
        raise TomeHipError(f"merge_wavg_regrouped: x {tuple(x_full.shape)} does not hold {plan.n} groups of {P} "
                           f"tokens ({F} per clip) plus {cls} class token")
    if x_full.device != plan.device:
        raise TomeHipError("merge_wavg_regrouped: tensor and matching on different devices")
    if torch.is_grad_enabled() and x_full.requires_grad:
        raise TomeHipError("merge_wavg_regrouped: autograd through the HIP merge kernels is not implemented")
    x_full = x_full if x_full.is_contiguous() else x_full.contiguous()
    xcode = dtype_code(x_full, "x")
    if size is not None:
        if size.shape != (plan.n, P, 1):
            raise TomeHipError(f"size must be [{plan.n}, {P}, 1], got {tuple(size.shape)}")
        if size.dtype not in (x_full.dtype, torch.float32):
            size = size.to(x_full.dtype)
        size = size.contiguous()
        sdtype = size.dtype
    else:
        sdtype = x_full.dtype
    x_out = torch.empty((B, cls + (P - plan.r) * F, C), dtype=x_full.dtype, device=x_full.device)
    s_out = torch.empty((plan.n, P - plan.r, 1), dtype=sdtype, device=x_full.device)
    log = _log_size_like(s_out, log_size)
    out_bias = _out_bias(out_bias, x_full, C)
    if ln is None:
        if addend is not None or addend_grouped is not None or out_bias is not None:
            raise TomeHipError("merge_wavg_regrouped: addend / out_bias are only fused together with ln")
        with _on_device(x_full.device):
            rc = lib().tome_merge_wavg_regrouped(x_full.data_ptr(), xcode, _ptr(size), DTYPES[sdtype], B, F, P, C,
                                                 plan.r, cls, plan.src_idx.data_ptr(), plan.dst_idx.data_ptr(),
                                                 plan.unm_idx.data_ptr(), _ptr(plan.edge_keep), x_out.data_ptr(),
                                                 s_out.data_ptr(), _ptr(log), _stream(x_full.device))
        _check(rc, "tome_merge_wavg_regrouped")
        return x_out, s_out
    weight, bias, eps = ln
    grouped = 0
    if addend_grouped is not None:
        # the residual where the spatial attention left it: [B*F, has_cls + P, C] (+ the class tokens' own [B, 1, C])
        if addend is not None:
            raise TomeHipError("merge_wavg_regrouped: pass addend or addend_grouped, not both")
        if tuple(addend_grouped.shape) != (plan.n, cls + P, C) or addend_grouped.dtype != x_full.dtype \
                or addend_grouped.device != x_full.device:
            raise TomeHipError(f"merge_wavg_regrouped: addend_grouped must be {(plan.n, cls + P, C)} of x's dtype")
        addend = addend_grouped if addend_grouped.is_contiguous() else addend_grouped.contiguous()
        grouped = 1
        if cls_addend is not None:
            if cls_addend.numel() != B * C or cls_addend.dtype != x_full.dtype or cls_addend.device != x_full.device:
                raise TomeHipError(f"merge_wavg_regrouped: cls_addend must hold {(B, C)} values of x's dtype")
            cls_addend = cls_addend.contiguous()
    elif addend is not None:
        if addend.shape != x_full.shape or addend.dtype != x_full.dtype or addend.device != x_full.device:
            raise TomeHipError("merge_wavg_regrouped: addend must match x in shape, dtype and device")
        addend = addend if addend.is_contiguous() else addend.contiguous()
    y_out = torch.empty_like(x_out)
    with _on_device(x_full.device):
        rc = lib().tome_merge_wavg_regrouped_ln(x_full.data_ptr(), xcode, _ptr(size), DTYPES[sdtype], B, F, P, C,
                                                plan.r, cls, plan.src_idx.data_ptr(), plan.dst_idx.data_ptr(),
                                                plan.unm_idx.data_ptr(), _ptr(plan.edge_keep), weight.data_ptr(),
                                                bias.data_ptr(), float(eps), _ptr(addend), grouped,
                                                _ptr(cls_addend) if grouped else None, x_out.data_ptr(),
                                                y_out.data_ptr(), s_out.data_ptr(), _ptr(log), _ptr(out_bias),
                                                _stream(x_full.device))
    _check(rc, "tome_merge_wavg_regrouped_ln")
    return x_out, y_out, s_out


def merge(plan: MatchPlan, x: torch.Tensor, mode: str) -> torch.Tensor:
    if mode not in MODES:
        raise TomeHipError(f"merge: unknown reduce mode {mode!r}")
    x = _prep_x(plan, x, "merge(x)", plan.T)
    n, T, C = x.shape
    out = torch.empty((n, T - plan.r, C), dtype=x.dtype, device=x.device)
    with _on_device(x.device):
        rc = lib().tome_merge(x.data_ptr(), dtype_code(x, "x"), n, T, C, plan.r, plan.src_idx.data_ptr(),
                              plan.dst_idx.data_ptr(), plan.unm_idx.data_ptr(), int(plan.distill_token), MODES[mode],
                              _ptr(plan.edge_keep), out.data_ptr(), _stream(x.device))
    _check(rc, "tome_merge")
    return out


def drop(plan: MatchPlan, x: torch.Tensor) -> torch.Tensor:
    x = _prep_x(plan, x, "drop(x)", plan.T)
    n, T, C = x.shape
    out = torch.empty((n, T - plan.r, C), dtype=x.dtype, device=x.device)
    with _on_device(x.device):
        rc = lib().tome_drop(x.data_ptr(), dtype_code(x, "x"), n, T, C, plan.r, plan.unm_idx.data_ptr(),
                             int(plan.distill_token), out.data_ptr(), _stream(x.device))
    _check(rc, "tome_drop")
    return out


def prop_attention_ok(q: torch.Tensor) -> bool:
    """Can tome_prop_attention take these heads?  ([B, H, N, 64] views of 16-bit device tensors, rows 16-byte aligned.)"""
    return (q.is_cuda and q.dim() == 4 and q.shape[-1] == 64 and q.dtype in (torch.bfloat16, torch.float16)
            and q.stride(-1) == 1 and all(s % 8 == 0 for s in q.stride()[:3]) and q.data_ptr() % 16 == 0
            and not (torch.is_grad_enabled() and q.requires_grad))


def prop_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, size: Optional[torch.Tensor], scale: float,
                   bias_skip: bool = False, log_bias: Optional[torch.Tensor] = None,
                   out: Optional[torch.Tensor] = None, checked: bool = False) -> torch.Tensor:
    """softmax(q k^T * scale + log(size) on the keys) v for head views q [B, H, N, 64], k / v [B, H, Nk, 64] (any
    strides with contiguous channels: the slices of a qkv buffer are read in place); returns [B, N, H*64].
    `size` is the token size tensor [B, Nk(-1), 1] (its log comes from the merge kernel when that emitted it),
    or `log_bias` an fp32 [B, Nk(-1)] view that already holds the bias, or both None.  bias_skip: the TimeSformer
    form -- key 0 / query 0 unbiased, the bias describes keys 1..N-1.  `out`: a [B, N, H, 64] view (channels
    contiguous) to write into instead of a fresh tensor."""
    # (checked: the caller has just asked prop_attention_ok about q, k and v -- the patches do, per layer; at the
    # reference's batch of 8 the forward is bound by host time and the three repeated checks are 7 us of it)
    for t, name in (() if checked else ((q, "q"), (k, "k"), (v, "v"))):
        require_device(t, f"prop_attention({name})")
        if not prop_attention_ok(t):
            raise TomeHipError(f"prop_attention: {name} must be a [B, H, N, 64] 16-bit view with 16-byte aligned rows, "
                               f"got {tuple(t.shape)} {t.dtype} strides {t.stride()}")
    if k.dtype != q.dtype or v.dtype != q.dtype or k.device != q.device or v.device != q.device:
        raise TomeHipError("prop_attention: q, k, v must share dtype and device")
    B, H, N, D = q.shape
    if k.shape != v.shape or k.shape[:2] != (B, H):
        raise TomeHipError(f"prop_attention: q {tuple(q.shape)}, k {tuple(k.shape)}, v {tuple(v.shape)} do not match")
    Nk = k.shape[2]
    if bias_skip and Nk != N:
        raise TomeHipError("prop_attention: bias_skip needs as many keys as queries")
    nb = Nk - (1 if bias_skip else 0)
    log = None
    if size is not None:
        if tuple(size.shape) != (B, nb, 1):
            raise TomeHipError(f"prop_attention: size must be {(B, nb, 1)}, got {tuple(size.shape)}")
        log = log_of_size(size).reshape(B, -1).float().contiguous()
    elif log_bias is not None:
        if tuple(log_bias.shape) != (B, nb) or log_bias.dtype != torch.float32 or log_bias.stride(1) != 1 \
                or log_bias.device != q.device:
            raise TomeHipError(f"prop_attention: log_bias must be an fp32 {(B, nb)} view with contiguous rows")
        log = log_bias
    ostr = None
    if out is None:
        result = out = torch.empty((B, N, H * D), dtype=q.dtype, device=q.device)
    else:
        if tuple(out.shape) != (B, N, H, D) or out.dtype != q.dtype or out.device != q.device or out.stride(3) != 1:
            raise TomeHipError(f"prop_attention: out must be a {(B, N, H, D)} view of the q dtype with contiguous channels")
        ostr = (ctypes.c_int64 * 3)(out.stride(0), out.stride(2), out.stride(1))
        result = out
    strides = [(ctypes.c_int64 * 3)(*t.stride()[:3]) for t in (q, k, v)]
    with _on_device(q.device):
        rc = lib().tome_prop_attention(q.data_ptr(), k.data_ptr(), v.data_ptr(), dtype_code(q, "q"), B, H, N, Nk, D,
                                       strides[0], strides[1], strides[2], _ptr(log),
                                       0 if log is None else log.stride(0), 1 if bias_skip else 0, float(scale),
                                       out.data_ptr(), ostr, _stream(q.device))
    _check(rc, "tome_prop_attention")
    return result


def prop_attention_segments(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, nseg: int, scale: float,
                            log_bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Per-segment attention in one launch: queries q [B, H, N, 64]; keys / values [B, H, nseg*P, 64] views whose
    rows [s*P, (s+1)*P) form segment s; every query takes softmax(q k_s^T * scale + log_bias_s) v_s for each
    segment separately (Motionformer: the P keys of one frame, motionformer.py:98-121).  log_bias: fp32
    [B, nseg*P] (contiguous rows) or None.  Returns y [B, N, nseg, H*64]."""
    for t, name in ((q, "q"), (k, "k"), (v, "v")):
        require_device(t, f"prop_attention_segments({name})")
        if not prop_attention_ok(t) or t.dtype != q.dtype or t.device != q.device:
            raise TomeHipError(f"prop_attention_segments: {name} must be a [B, H, N, 64] 16-bit view with 16-byte "
                               f"aligned rows, got {tuple(t.shape)} {t.dtype} strides {t.stride()}")
    B, H, N, D = q.shape
    nseg = int(nseg)
    if k.shape != v.shape or k.shape[:2] != (B, H) or nseg < 1 or k.shape[2] % nseg:
        raise TomeHipError(f"prop_attention_segments: q {tuple(q.shape)}, k {tuple(k.shape)}, v {tuple(v.shape)}, "
                           f"{nseg} segments do not match")
    P = k.shape[2] // nseg
    if log_bias is not None and (tuple(log_bias.shape) != (B, nseg * P) or log_bias.dtype != torch.float32
                                 or log_bias.stride(1) != 1 or log_bias.device != q.device):
        raise TomeHipError(f"prop_attention_segments: log_bias must be an fp32 {(B, nseg * P)} view with contiguous rows")
    y = torch.empty((B, N, nseg, H * D), dtype=q.dtype, device=q.device)
    strides = [(ctypes.c_int64 * 3)(*t.stride()[:3]) for t in (q, k, v)]
    ostr = (ctypes.c_int64 * 3)(y.stride(0), D, y.stride(1))  # {batch, head, token}
    seg = (ctypes.c_int64 * 4)(P * k.stride(2), P * v.stride(2), y.stride(2), P)
    with _on_device(q.device):
        rc = lib().tome_prop_attention_segments(q.data_ptr(), k.data_ptr(), v.data_ptr(), dtype_code(q, "q"), B, H, N, P,
                                                D, strides[0], strides[1], strides[2], _ptr(log_bias),
                                                0 if log_bias is None else log_bias.stride(0), float(scale),
                                                y.data_ptr(), ostr, nseg, seg, _stream(q.device))
    _check(rc, "tome_prop_attention_segments")
    return y


def short_attention_ok(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> bool:
    """Can tome_short_attention take these?  [B, H, N <= 8, 64] views of 16-bit device tensors whose heads lie side
    by side in a token's row (head stride 64), rows 16-byte aligned, no gradient wanted."""
    def ok(t):
        return (t.is_cuda and t.dim() == 4 and t.shape == q.shape and t.dtype == q.dtype and t.device == q.device
                and t.stride(3) == 1 and t.stride(1) == 64 and t.stride(0) % 8 == 0 and t.stride(2) % 8 == 0
                and t.data_ptr() % 16 == 0 and not (torch.is_grad_enabled() and t.requires_grad))
    return (q.dim() == 4 and q.dtype in (torch.bfloat16, torch.float16) and q.shape[-1] == 64 and 1 <= q.shape[2] <= 8
            and ok(q) and ok(k) and ok(v))


def short_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float, checked: bool = False
                    ) -> torch.Tensor:
    """softmax(q k^T * scale) v over sequences of at most 8 tokens (TimeSformer's temporal attention): q, k, v
    [B, H, N, 64] views of one qkv projection, read in place; returns [B, N, H*64].  checked: the caller has just
    asked short_attention_ok."""
    for t, name in (() if checked else ((q, "q"), (k, "k"), (v, "v"))):
        require_device(t, f"short_attention({name})")
    if not checked and not short_attention_ok(q, k, v):
        raise TomeHipError(f"short_attention: q, k, v must be [B, H, N <= 8, 64] 16-bit views with head stride 64 and "
                           f"16-byte aligned rows, got {tuple(q.shape)} {q.dtype} strides {q.stride()}")
    B, H, N, D = q.shape
    out = torch.empty((B, N, H * D), dtype=q.dtype, device=q.device)
    strides = [(ctypes.c_int64 * 3)(*t.stride()[:3]) for t in (q, k, v)]
    with _on_device(q.device):
        rc = lib().tome_short_attention(q.data_ptr(), k.data_ptr(), v.data_ptr(), dtype_code(q, "q"), B, H, N, D,
                                        strides[0], strides[1], strides[2], float(scale), out.data_ptr(),
                                        _stream(q.device))
    _check(rc, "tome_short_attention")
    return out


def trajectory_mix_ok(q2: torch.Tensor, k2: torch.Tensor, val: torch.Tensor, heads: int) -> bool:
    """Can tome_trajectory_mix take these?  q2 [B, S, C], k2 / val [B, S, F, C] views (rows contiguous over C,
    (b, s, f) rows evenly spaced), 16-bit, head dim 64, at most 16 heads and 8 frames."""
    if not (q2.is_cuda and q2.dtype in (torch.bfloat16, torch.float16) and q2.dim() == 3 and k2.dim() == 4):
        return False
    B, S, C = q2.shape
    F = k2.shape[2]
    def rows_ok(t):
        return (t.shape == (B, S, F, C) and t.dtype == q2.dtype and t.stride(3) == 1 and t.stride(2) % 8 == 0
                and t.stride(1) == F * t.stride(2) and t.stride(0) == S * t.stride(1) and t.data_ptr() % 16 == 0)
    return (C == heads * 64 and heads <= 16 and F <= 8 and q2.is_contiguous() and q2.data_ptr() % 16 == 0
            and rows_ok(k2) and rows_ok(val) and not (torch.is_grad_enabled() and (q2.requires_grad or k2.requires_grad)))


def trajectory_mix(q2: torch.Tensor, k2: torch.Tensor, val: torch.Tensor, heads: int, scale: float,
                   want_attn: bool = True, out: Optional[torch.Tensor] = None):
    """softmax over the F frames of (q2*scale . k2[f]) per (batch, token, head), then the weighted sum of val[f]:
    returns (out [B, S, C], attn [B, heads, S, F] fp32 or None).  `out`: a [B, S, C] view with contiguous rows
    (stride(1) == C) to write into -- e.g. rows 1.. of the [B, 1+S, C] buffer whose row 0 takes the class token."""
    if not trajectory_mix_ok(q2, k2, val, heads):
        raise TomeHipError("trajectory_mix: unsupported tensors (16-bit, head dim 64, <= 16 heads, <= 8 frames, "
                           "evenly spaced 16-byte aligned rows)")
    B, S, C = q2.shape
    F = k2.shape[2]
    if out is None:
        out = torch.empty((B, S, C), dtype=q2.dtype, device=q2.device)
    elif (tuple(out.shape) != (B, S, C) or out.dtype != q2.dtype or out.device != q2.device or out.stride(2) != 1
          or out.stride(1) != C or out.stride(0) % 8 or out.stride(0) < S * C or out.data_ptr() % 16):
        raise TomeHipError(f"trajectory_mix: out must be a {(B, S, C)} view of the q2 dtype with contiguous 16-byte "
                           f"aligned rows, got {tuple(out.shape)} {out.dtype} strides {out.stride()}")
    attn = torch.empty((B, heads, S, F), dtype=torch.float32, device=q2.device) if want_attn else None
    with _on_device(q2.device):
        rc = lib().tome_trajectory_mix(q2.data_ptr(), k2.data_ptr(), val.data_ptr(), dtype_code(q2, "q2"), B, S, F, heads,
                                       64, k2.stride(2), val.stride(2), float(scale), out.data_ptr(), out.stride(0),
                                       _ptr(attn), _stream(q2.device))
    _check(rc, "tome_trajectory_mix")
    return out, attn


def drop_regrouped(plan: MatchPlan, x_full: torch.Tensor, frames: int, has_cls: bool = True) -> torch.Tensor:
    """drop on the interleaved layout (see merge_wavg_regrouped): x_full [B, has_cls + P*F, C] ->
    [B, has_cls + (P-r)*F, C], replacing rearrange -> drop -> rearrange -> cat (timesformer.py:111-131)."""
    require_device(x_full, "drop_regrouped(x)")
    if x_full.dim() != 3:
        raise TomeHipError(f"drop_regrouped: x must be [B, tokens, C], got {tuple(x_full.shape)}")
    B, N, C = x_full.shape
    cls = 1 if has_cls else 0
    F, P = int(frames), plan.T
    if N != cls + P * F or plan.n != B * F:
        raise TomeHipError(f"drop_regrouped: x {tuple(x_full.shape)} does not hold {plan.n} groups of {P} tokens "
                           f"({F} per clip) plus {cls} class token")
    if x_full.device != plan.device:
        raise TomeHipError("drop_regrouped: tensor and matching on different devices")
    if torch.is_grad_enabled() and x_full.requires_grad:
        raise TomeHipError("drop_regrouped: autograd through the HIP merge kernels is not implemented")
    x_full = x_full if x_full.is_contiguous() else x_full.contiguous()
    out = torch.empty((B, cls + (P - plan.r) * F, C), dtype=x_full.dtype, device=x_full.device)
    with _on_device(x_full.device):
        rc = lib().tome_drop_regrouped(x_full.data_ptr(), dtype_code(x_full, "x"), B, F, P, C, plan.r, cls,
                                       plan.unm_idx.data_ptr(), out.data_ptr(), _stream(x_full.device))
    _check(rc, "tome_drop_regrouped")
    return out


def unmerge(plan: MatchPlan, x: torch.Tensor) -> torch.Tensor:
    x = _prep_x(plan, x, "unmerge(x)", plan.T - plan.r)
    n, _, C = x.shape
    out = torch.empty((n, plan.T, C), dtype=x.dtype, device=x.device)
    with _on_device(x.device):
        rc = lib().tome_unmerge(x.data_ptr(), dtype_code(x, "x"), n, plan.T, C, plan.r, plan.src_idx.data_ptr(),
                                plan.dst_idx.data_ptr(), plan.unm_idx.data_ptr(), out.data_ptr(), _stream(x.device))
    _check(rc, "tome_unmerge")
    return out


def gelu_ok(x: torch.Tensor) -> bool:
    return (x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and x.is_contiguous() and x.numel() % 8 == 0
            and x.numel() > 0 and x.data_ptr() % 16 == 0 and not (torch.is_grad_enabled() and x.requires_grad))


def gelu_erf(x: torch.Tensor, inplace: bool = False) -> torch.Tensor:
    """nn.GELU() (exact erf form) of a contiguous 16-bit tensor, bit-identical to torch's, as one streaming pass."""
    if not gelu_ok(x):
        raise TomeHipError("gelu_erf: contiguous 16-bit device tensor with a multiple of 8 elements required")
    y = x if inplace else torch.empty_like(x)
    with _on_device(x.device):
        rc = lib().tome_gelu_erf(x.data_ptr(), dtype_code(x, "x"), x.numel(), y.data_ptr(), _stream(x.device))
    _check(rc, "tome_gelu_erf")
    return y


def tubelet_rows_ok(x: torch.Tensor, kt: int, kh: int, kw: int) -> bool:
    """x [B, C, T, H, W] (any view with unit stride along W) can be regrouped by tome_tubelet_rows."""
    if not (x.is_cuda and x.dim() == 5 and x.dtype in DTYPES and x.numel() > 0 and x.stride(4) == 1
            and not (torch.is_grad_enabled() and x.requires_grad)):
        return False
    es = x.element_size()
    _, _, T, H, W = x.shape
    return (T % kt == 0 and H % kh == 0 and W % kw == 0 and (kw * es) % 16 == 0 and x.data_ptr() % 16 == 0
            and all(s >= 0 and (s * es) % 16 == 0 for s in x.stride()[:4]))


def tubelet_rows(x: torch.Tensor, kt: int, kh: int, kw: int) -> torch.Tensor:
    """rows [B, T'*H'*W', C*kt*kh*kw] of a clip x [B, C, T, H, W]: the matrix a stride == kernel convolution's weight
    multiplies (tome_tubelet_rows: a pure 16-byte move, token order = conv(x).flatten(2).transpose(1, 2))."""
    if not tubelet_rows_ok(x, kt, kh, kw):
        require_device(x, "tubelet_rows(x)")
        raise TomeHipError(f"tubelet_rows: x {tuple(x.shape)} strides {x.stride()} with tubelets {(kt, kh, kw)} "
                           "cannot be regrouped in 16-byte chunks")
    B, C, T, H, W = x.shape
    rows = torch.empty((B, (T // kt) * (H // kh) * (W // kw), C * kt * kh * kw), dtype=x.dtype, device=x.device)
    strides = (ctypes.c_int64 * 4)(*x.stride()[:4])
    with _on_device(x.device):
        rc = lib().tome_tubelet_rows(x.data_ptr(), x.element_size(), B, C, T, H, W, strides, kt, kh, kw,
                                     rows.data_ptr(), _stream(x.device))
    _check(rc, "tome_tubelet_rows")
    return rows


def source_init(plan: MatchPlan, drop: bool = False) -> torch.Tensor:
    """The first layer's source matrix [n, T-r, T] fp32 (merge_source with source=None, merge.py:372-384; `drop`:
    what the drop closure makes of the identity) written straight from the matching's row map -- no [n,T,T]
    identity, no reduction over its zeros."""
    with _on_device(plan.device):
        st = _stream(plan.device)
        if plan.row_map is None:
            T1 = (plan.T + 1) // 2
            plan.row_map = torch.empty((plan.n, T1), dtype=torch.int32, device=plan.device)
            _check(lib().tome_row_map(plan.n, plan.T, plan.r, int(plan.distill_token), plan.src_idx.data_ptr(),
                                      plan.dst_idx.data_ptr(), plan.unm_idx.data_ptr(), plan.row_map.data_ptr(), st),
                   "tome_row_map")
        out = torch.empty((plan.n, plan.T - plan.r, plan.T), dtype=torch.float32, device=plan.device)
        _check(lib().tome_source_init(plan.n, plan.T, plan.r, int(plan.distill_token), int(bool(drop)),
                                      plan.row_map.data_ptr(), out.data_ptr(), st), "tome_source_init")
    return out
