"""ToMe token merging for MI355X -- same functions and signatures as the reference's
``tome/merge.py``, with the arithmetic done by the hand-written gfx950 kernels behind the C ABI
of ``include/tome_hip.h`` (see ``_abi.py``).

Reference lines each function stands in for (sjpollard/video-how-do-your-tokens-merge):
  bipartite_soft_matching         tome/merge.py:17-102
  bipartite_soft_matching_drop    tome/merge.py:215-271
  bipartite_soft_matching_hybrid  tome/merge.py:274-352
  merge_wavg                      tome/merge.py:355-369
  merge_source                    tome/merge.py:372-384
  do_nothing                      tome/merge.py:13-14

The returned ``merge`` / ``unmerge`` / ``drop`` are real closures over ``unm_idx``, ``src_idx``,
``dst_idx`` (int64 device tensors shaped [n,r,1] / [n,T1-r,1]) and ``r`` exactly like the reference's,
so code that introspects them keeps working; they also carry ``.plan`` for the fused paths.

Tensors must live on a HIP device: there is no CPU implementation in this package.

Autograd (only the index computation is ``no_grad`` in the reference, merge.py:49; tools/train_net.py:727-741
patches models for training): the kernels are inference code.  When a tensor handed to ``merge`` / ``unmerge`` /
``drop`` / ``merge_wavg`` requires grad (and grad mode is on), the closure applies its index tensors with the
framework's own differentiable gather / scatter_reduce ops on the tensor's device instead (``_merge_with_autograd``
below: the op sequence of merge.py:75-100) -- the matching itself always runs on the HIP kernels.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch

from . import _abi, _overlap


def do_nothing(x, mode=None):
    return x


class HeadMeanKeys:
    """The metric ``k.mean(1)`` kept as a promise: holds the per-head keys [n,H,T,64] (a view of the attention's
    qkv buffer) -- or [outer,inner,H,T,64] when the merge groups are interleaved inside a clip's sequence
    (Motionformer's ``'(b h) (s f) d -> (b f) h s d'``; group = outer*inner + inner index).  The matching functions
    of this module read the keys in place (tome_match_keys: head mean, unit vectors, similarity in one pass over
    them); anything else that wants the tensor calls ``.materialize()``.  Shape queries behave like the metric's."""

    def __init__(self, keys: torch.Tensor):
        if keys.dim() not in (4, 5):
            raise ValueError(f"HeadMeanKeys: [n,H,T,D] or [outer,inner,H,T,D] expected, got {tuple(keys.shape)}")
        self.keys = keys
        self.early = None  # (r, class_token, distill_token, plan) of a matching already issued (tome/_overlap.py)

    @property
    def shape(self):
        *lead, _, t, d = self.keys.shape
        n = lead[0] * (lead[1] if len(lead) == 2 else 1)
        return torch.Size((n, t, d))

    @property
    def device(self):
        return self.keys.device

    @property
    def dtype(self):
        return self.keys.dtype

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def materialize(self) -> torch.Tensor:
        m = self.keys.mean(-3)
        return m if m.dim() == 3 else m.reshape(-1, m.shape[-2], m.shape[-1])


def _scores_for_random(metric: torch.Tensor) -> torch.Tensor:
    """merge.py:54-57 / :239-242 -- the random variants replace the similarity by uniform noise drawn on
    the metric's device (torch's generator, exactly as the reference draws it)."""
    length = metric.size(1)
    len_a, len_b = (length + 1) // 2, length // 2
    return torch.rand(size=(metric.size(0), len_a, len_b), device=metric.device)


def _plan(metric, r, class_token, distill_token, random: bool, **want) -> Optional[_abi.MatchPlan]:
    with torch.no_grad():
        if isinstance(metric, HeadMeanKeys):
            found, plan = _overlap.take(metric, r, class_token, distill_token)
            if found and not random and not any(want.values()):
                return plan  # issued beside the block's attention, on the side stream this stream now waits for
            if not random and _abi.keys_fusable(metric.keys):
                return _abi.match_keys(metric.keys, r, class_token, distill_token, checked=True, **want)
            metric = metric.materialize()
        if random:
            _abi.require_device(metric, "bipartite_soft_matching(metric)")
            t = metric.shape[1]
            if _abi.effective_r(t, r, class_token, distill_token) <= 0:
                return None
            return _abi.match_scores(_scores_for_random(metric), t, r, class_token, distill_token, **want)
        return _abi.match(metric, r, class_token, distill_token, **want)


def bipartite_soft_matching(
    metric: torch.Tensor,
    r: int,
    class_token: bool = False,
    distill_token: bool = False,
    mode: str = "merge",
) -> Tuple[Callable, Callable]:
    """Balanced (even/odd) bipartite matching; input [batch, tokens, channels]; at most 50% of the
    unprotected tokens are merged.  Returns (merge, unmerge)."""
    if mode not in ("merge", "random_merge"):
        raise ValueError(f"bipartite_soft_matching: mode {mode!r} (expected 'merge' or 'random_merge')")
    plan = _plan(metric, r, class_token, distill_token, random=(mode == "random_merge"))
    if plan is None:
        return do_nothing, do_nothing
    return _make_merge_pair(plan)


def _wants_autograd(x: torch.Tensor) -> bool:
    return torch.is_grad_enabled() and x.requires_grad


_SCATTER_MODE = {"sum": "sum", "mean": "mean", "prod": "prod", "max": "amax", "amax": "amax", "min": "amin", "amin": "amin"}


def _interleave_distill(first: torch.Tensor, second: torch.Tensor) -> torch.Tensor:
    # merge.py:82-83: with a distillation token the class token of `first` and the distill token of `second` lead
    return torch.cat([first[:, :1], second[:, :1], first[:, 1:], second[:, 1:]], dim=1)


def _merge_with_autograd(plan, x: torch.Tensor, mode: str, keep_sources: bool = True) -> torch.Tensor:
    """The merge (keep_sources) or drop callback on differentiable framework ops: even tokens gathered by
    ``unm_idx``, sources scattered onto their odd destinations with ``scatter_reduce(include_self=True)``
    (merge.py:75-85, :257-266), a hybrid matching's threshold flags first (merge.py:326)."""
    if mode not in _SCATTER_MODE:
        raise _abi.TomeHipError(f"merge: unknown reduce mode {mode!r}")
    n, t, c = x.shape
    if n != plan.n or t != plan.T:
        raise _abi.TomeHipError(f"merge(x): expected [{plan.n}, {plan.T}, C], got {tuple(x.shape)}")
    a_rows, b_rows = x[:, 0::2], x[:, 1::2]
    r, t1 = plan.r, a_rows.shape[1]
    kept = a_rows.gather(1, plan.unm_idx.expand(n, t1 - r, c))
    if keep_sources:
        where = plan.dst_idx.expand(n, r, c)
        if plan.edge_keep is not None:
            flags = plan.edge_keep.reshape(n, r, 1).to(x.dtype).expand(n, r, c)
            b_rows = b_rows.scatter_reduce(1, where, flags, reduce="prod")
        b_rows = b_rows.scatter_reduce(1, where, a_rows.gather(1, plan.src_idx.expand(n, r, c)),
                                       reduce=_SCATTER_MODE[mode])
    if plan.distill_token:
        return _interleave_distill(kept, b_rows)
    return torch.cat([kept, b_rows], dim=1)


def _unmerge_with_autograd(plan, x: torch.Tensor) -> torch.Tensor:
    """merge.py:87-100 on differentiable ops: odd slots take the destination rows, even slots their unmerged row or a
    copy of the destination they were merged into."""
    n, _, c = x.shape
    r, t1 = plan.r, (plan.T + 1) // 2
    u = t1 - r
    kept, b_rows = x[:, :u], x[:, u:]
    copies = b_rows.gather(1, plan.dst_idx.expand(n, r, c))
    a_rows = x.new_zeros((n, t1, c)).scatter(1, plan.unm_idx.expand(n, u, c), kept)
    a_rows = a_rows.scatter(1, plan.src_idx.expand(n, r, c), copies)
    return torch.stack([a_rows[:, :b_rows.shape[1]], b_rows], dim=2).flatten(1, 2) if t1 == b_rows.shape[1] \
        else torch.cat([torch.stack([a_rows[:, :-1], b_rows], dim=2).flatten(1, 2), a_rows[:, -1:]], dim=1)


def _make_merge_pair(plan: _abi.MatchPlan) -> Tuple[Callable, Callable]:
    unm_idx, src_idx, dst_idx = plan.unm_idx, plan.src_idx, plan.dst_idx
    r, distill_token = plan.r, plan.distill_token

    def merge(x: torch.Tensor, mode="mean") -> torch.Tensor:
        # the index tensors are closure variables on purpose (same names as the reference's closure, so
        # `merge.__closure__` introspection keeps working); the kernels read them through `plan`
        _closure = (unm_idx, src_idx, dst_idx, r, distill_token)  # noqa: F841
        if _wants_autograd(x):
            return _merge_with_autograd(plan, x, mode)
        return _abi.merge(plan, x, mode)

    def unmerge(x: torch.Tensor) -> torch.Tensor:
        _closure = (unm_idx, src_idx, dst_idx, r)  # noqa: F841
        if _wants_autograd(x):
            if distill_token:
                raise _abi.TomeHipError("unmerge: the distillation layout has no differentiable form here")
            return _unmerge_with_autograd(plan, x)
        return _abi.unmerge(plan, x)

    merge.plan = plan
    unmerge.plan = plan
    return merge, unmerge


def bipartite_soft_matching_drop(
    metric: torch.Tensor,
    r: int,
    class_token: bool = False,
    distill_token: bool = False,
    mode: str = "drop",
):
    """Same matching, but the selected tokens are discarded instead of merged.  Returns ``drop``
    (a single callable) -- or the (do_nothing, do_nothing) pair when r <= 0, as the reference does."""
    if mode not in ("drop", "random_drop"):
        raise ValueError(f"bipartite_soft_matching_drop: mode {mode!r}")
    plan = _plan(metric, r, class_token, distill_token, random=(mode == "random_drop"))
    if plan is None:
        return do_nothing, do_nothing
    und_idx, src_idx = plan.unm_idx, plan.src_idx
    r, distill_token = plan.r, plan.distill_token

    def drop(x: torch.Tensor) -> torch.Tensor:
        _closure = (und_idx, src_idx, r, distill_token)  # noqa: F841  (closure variables as in the reference)
        if _wants_autograd(x):
            return _merge_with_autograd(plan, x, "sum", keep_sources=False)
        return _abi.drop(plan, x)

    drop.plan = plan
    return drop


def bipartite_soft_matching_hybrid(
    metric: torch.Tensor,
    r: int,
    class_token: bool = False,
    distill_token: bool = False,
    mode: str = "merge",
    threshold: float = 0.0,
) -> Tuple[Callable, Callable]:
    """Merge, but a destination whose incoming edge scores below ``threshold`` loses its own
    contribution first (merge.py:326)."""
    if mode not in ("merge", "hybrid", "random_merge"):
        raise ValueError(f"bipartite_soft_matching_hybrid: mode {mode!r}")
    plan = _plan(metric, r, class_token, distill_token, random=(mode == "random_merge"), want_node_max=True)
    if plan is None:
        return do_nothing, do_nothing
    with torch.no_grad():
        plan.edge_keep = _abi.edge_keep(plan, threshold)
    merge, unmerge = _make_merge_pair(plan)
    return merge, unmerge


def kth_bipartite_soft_matching(metric: torch.Tensor, k: int):
    """tome/merge.py:105-158 (every k-th token as destination set).  No patch, driver or notebook of the reference
    calls it (outside the hot path); the name exists so that `from tome.merge import ...` of code written
    against the reference fails here, with the reason, instead of at import."""
    raise _abi.TomeHipError("kth_bipartite_soft_matching is not provided by the MI355X path (no caller in the "
                            "reference's patches; see INTEGRATION.md, API table)")


def random_bipartite_soft_matching(metric: torch.Tensor, r: int):
    """tome/merge.py:161-212 (a random source set).  As kth_bipartite_soft_matching: present by name, refused loudly.
    (The random MODES of the patches -- mode='random_merge' / 'random_drop', merge.py:54-57 -- are implemented.)"""
    raise _abi.TomeHipError("random_bipartite_soft_matching is not provided by the MI355X path (no caller in the "
                            "reference's patches; use mode='random_merge' of bipartite_soft_matching)")


def merge_wavg(merge: Callable, x: torch.Tensor, size: Optional[torch.Tensor] = None, log_size: bool = False
               ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Size-weighted average merge; returns the merged tensor and the new token sizes.  With a merge
    made by this package the whole ``x*size -> sum, sum -> x/size`` chain is one kernel launch.
    ``log_size=True`` (not in the reference's signature) makes that launch also emit ``log(size')`` for the
    next block's proportional-attention bias; consumers fetch it with ``_abi.log_of_size(size)``."""
    plan = getattr(merge, "plan", None)
    if plan is not None and not (_wants_autograd(x) or (size is not None and _wants_autograd(size))):
        return _abi.merge_wavg(plan, x, size, log_size=log_size)
    # foreign callables, do_nothing, and tensors that require grad (the closure then runs on the framework's
    # differentiable ops): the reference's op sequence on the tensors' own device
    if size is None:
        size = torch.ones_like(x[..., 0, None])
    x = merge(x * size, mode="sum")
    size = merge(size, mode="sum")
    x = x / size
    return x, size


def merge_source(merge: Callable, x: torch.Tensor, source: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Source tracking: adjacency between the initial tokens and the merged groups."""
    plan = getattr(merge, "plan", None)
    if source is None:
        if plan is not None and plan.edge_keep is None:
            # merging the identity with "max" = the one-hot rows of the matching's row map: written directly
            # (tome_source_init), the [n, T, T] identity is never allocated.  (Not for a hybrid matching: there a
            # destination with an incoming edge below the threshold is zeroed before the amax, merge.py:326-331, so its
            # own column is 0 -- the generic path below keeps that.)
            _abi.require_device(x, "merge_source(x)")
            if x.shape[0] != plan.n or x.shape[1] != plan.T or x.device != plan.device:
                raise _abi.TomeHipError(f"merge_source: x {tuple(x.shape)} on {x.device} does not fit the matching "
                                        f"({plan.n} groups of {plan.T} tokens on {plan.device})")
            return _abi.source_init(plan)
        n, t, _ = x.shape
        source = torch.eye(t, device=x.device)[None, ...].expand(n, t, t)
    source = merge(source, mode="max")
    return source
