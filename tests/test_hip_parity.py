"""GPU parity tests (run with `-m gpu` on an MI355X through gpurun): the HIP path, called through
the drop-in `tome` package (ctypes -> C ABI -> gfx950 kernels), against

  * the CPU oracle (oracle/tome_oracle.c) on the same seeded inputs: BIT-EXACT indices and, for
    fp32, bit-exact values (the oracle restates the kernels' arithmetic contract);
  * the golden vectors produced by the real reference (tests/golden/*.npz): indices bit-exact per the
    certificate stored with each case, fp32 values bit-exact, bf16/fp16 values within the stated
    tolerance.

Nothing here reads /root/reference.
"""
import os

import numpy as np
import pytest
import torch

import golden_io as G
import oracle
import synth

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _tome():
    import tome  # noqa: F401  (video-how-do-your-tokens-merge_amd/ is on sys.path via conftest)
    from tome import merge as tm
    return tm


def closure_vars(fn):
    return dict(zip(fn.__code__.co_freevars, (c.cell_contents for c in fn.__closure__)))


def dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(dtype)


def host(t):
    return t.detach().float().cpu().numpy()


# ------------------------------------------------------------------------------------------------
# matching: indices
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", G.match_cases(), ids=lambda c: c["id"])
def test_match_vs_golden_and_oracle(case):
    tm = _tome()
    metric = G.metric_of(case)
    merge, unmerge = tm.bipartite_soft_matching(dev(metric), case["r"], case["cls"], case["distill"])
    if case["r_eff"] == 0:
        assert merge is tm.do_nothing and unmerge is tm.do_nothing
        return
    cv = closure_vars(merge)  # same introspection the fixtures were read with
    assert cv["r"] == case["r_eff"]
    src, dst, unm = (cv[k].cpu().numpy() for k in ("src_idx", "dst_idx", "unm_idx"))
    assert src.dtype == np.int64 and src.shape == (case["n"], case["r_eff"], 1)
    z = G.arrays("match")
    # vs the reference (certificate-aware)
    G.check_indices(case, src, dst, unm, z[case["id"] + "_src"], z[case["id"] + "_dst"], z[case["id"] + "_unm"])
    # vs the oracle: always bit-exact, ties and all
    plan = oracle.match(metric, case["r"], case["cls"], case["distill"])
    np.testing.assert_array_equal(src, plan.src_idx)
    np.testing.assert_array_equal(dst, plan.dst_idx)
    np.testing.assert_array_equal(unm, plan.unm_idx)


@pytest.mark.parametrize("kind", ["normal", "clustered"])
@pytest.mark.parametrize("shape", [(3, 64, 64), (2, 65, 64), (5, 197, 64), (2, 333, 96), (1, 1568, 64),
                                   (4, 784, 64), (2, 100, 130), (3, 31, 7), (1, 3137, 64), (16, 196, 64)])
def test_match_bit_exact_vs_oracle(kind, shape):
    """More shapes than the fixtures hold, including ragged tiles, D off the 64 grid and tie-heavy
    clustered keys: src/dst/unm AND node_max must equal the oracle bit for bit."""
    from tome import _abi
    n, T, D = shape
    metric = (synth.normal_like if kind == "normal" else synth.clustered)(shape, 4242 + T * 7 + D)
    for r, cls in ((16, False), (T, False), (5, True)):
        plan = oracle.match(metric, r, cls, False)
        got = _abi.match(dev(metric), r, cls, False, want_node_max=True, want_row_map=True)
        np.testing.assert_array_equal(got.src_idx.cpu().numpy(), plan.src_idx)
        np.testing.assert_array_equal(got.dst_idx.cpu().numpy(), plan.dst_idx)
        np.testing.assert_array_equal(got.unm_idx.cpu().numpy(), plan.unm_idx)
        np.testing.assert_array_equal(got.node_max.cpu().numpy().view(np.uint32), plan.node_max.view(np.uint32))
        # row_map: where every even token lands in the merged sequence
        T1 = (T + 1) // 2
        want_map = np.empty((n, T1), np.int32)
        U = T1 - plan.r
        for g in range(n):
            want_map[g, plan.unm_idx[g, :, 0]] = np.arange(U)
            want_map[g, plan.src_idx[g, :, 0]] = U + plan.dst_idx[g, :, 0]
        np.testing.assert_array_equal(got.row_map.cpu().numpy(), want_map)


def _filter_metric(kind, n, T, D, seed):
    """bf16 metrics that stress the candidate filter of csrc/tome_match_filter.h in different ways."""
    g = np.random.default_rng(seed)
    if kind == "normal":
        m = synth.normal_like((n, T, D), seed)
    elif kind == "clustered":          # few directions + small noise: many near-ties inside the window
        m = synth.clustered((n, T, D), seed)
    elif kind == "duplicates":         # exact ties: every token exists 8 times -> lists overflow -> fp32 pass
        base = synth.normal_like((n, max(1, (T + 7) // 8), D), seed)
        m = np.tile(base, (1, 8, 1))[:, :T]
    elif kind == "ramp":               # odd tokens approach every even token's direction along j: each new column is
        d = synth.normal_like((n, 1, D), seed)   # a new running maximum -> long record sequences -> overflow
        noise = synth.normal_like((n, T, D), seed + 1)
        w = np.linspace(3.0, 0.02, T, dtype=np.float32).reshape(1, T, 1)
        m = d + w * noise
    elif kind == "nan":                # zero / NaN / inf tokens among normal ones (merge.py:51 has no epsilon)
        m = synth.normal_like((n, T, D), seed)
        for gi in range(n):
            idx = g.choice(T, size=min(5, T), replace=False)
            m[gi, idx[0]] = 0.0
            if len(idx) > 1:
                m[gi, idx[1], 0] = np.nan
            if len(idx) > 2:
                m[gi, idx[2], 1] = np.inf
    elif kind == "tiny":               # tokens whose squared norm underflows (norm 0, unit channels +-inf): the filter
        m = synth.normal_like((n, T, D), seed)   # must stand aside
        m[:, ::7] *= 1e-33
    elif kind == "huge":               # tokens whose squared norm overflows (norm inf, unit vector 0): likewise
        m = synth.normal_like((n, T, D), seed)
        m[:, ::5] *= 1e25
    elif kind == "scales":             # norms spread over 2^-40 .. 2^40: the reciprocal scaling is exercised
        m = synth.normal_like((n, T, D), seed)
        m *= np.exp2(g.integers(-40, 41, size=(n, T, 1))).astype(np.float32)
    elif kind == "denormal_products":  # norms 2^-70 .. 2^-40 around FILT_NORM_LO = 1e-14: the products v_i[k] v_j[k] of two
        m = synth.normal_like((n, T, D), seed)   # such tokens are denormal (flushed inside the bf16 matrix instruction)
        m *= np.exp2(g.integers(-70, -39, size=(n, T, 1))).astype(np.float32)
    elif kind == "large_norms":        # norms 2^50 .. 2^62 around FILT_NORM_HI = 1e18 (and a squared norm that overflows)
        m = synth.normal_like((n, T, D), seed)
        m *= np.exp2(g.integers(50, 63, size=(n, T, 1))).astype(np.float32)
    else:
        raise ValueError(kind)
    return torch.from_numpy(np.ascontiguousarray(m)).to(torch.bfloat16)


def _same_plan(a, b, what):
    for name in ("src_idx", "dst_idx", "unm_idx"):
        assert torch.equal(getattr(a, name), getattr(b, name)), f"{what}: {name}"
    assert torch.equal(a.node_max.view(torch.int32), b.node_max.view(torch.int32)), f"{what}: node_max bits"


@pytest.mark.parametrize("kind", ["normal", "clustered", "duplicates", "ramp", "nan", "tiny", "huge", "scales",
                                  "denormal_products", "large_norms"])
@pytest.mark.parametrize("n,T,D", [(3, 197, 64), (2, 784, 64), (1, 1568, 64), (4, 65, 16), (2, 3, 8), (2, 64, 64),
                                   (1, 3137, 64), (5, 130, 40)])
def test_match_candidate_filter_is_bit_identical_to_the_fp32_pass(kind, n, T, D, monkeypatch):
    """bf16 metrics through the candidate filter (approximate scores on the bf16 matrix pipe, exact fp32 chain for the
    candidates of every row, fp32 pass for flagged tiles: csrc/tome_match_filter.h; TOME_SCORES_FILTER=2 forces it on
    small launches too) against the same call with the filter off (k_scores_rowmax on every tile) and against the
    oracle on the rounded values: src / dst / unm indices and the BITS of node_max, with and without a class token /
    a protected column, for r in {5, 16, all}.  The kinds cover the filter's every exit: one candidate per row, near-ties
    inside the window, exact ties and monotone columns (list overflow -> fp32 pass of that tile, run by a wave of the
    k_exact_rows launch), NaN / zero / inf tokens, norms outside the trusted range [1e-14, 1e18] (the tiles concerned on
    the fp32 pass: a row's own tile, every tile of a group for a column), norms spread over 80 binades, norms so small
    that products of two tokens' channels are denormal."""
    from tome import _abi
    metric = _filter_metric(kind, n, T, D, 1234 + 13 * T + D).to(DEV)
    host = metric.float().cpu().numpy()
    for r, cls, dist in ((5, False, False), (16, True, False), (T, False, True)):
        monkeypatch.setenv("TOME_SCORES_FILTER", "0")
        want = _abi.match(metric, r, cls, dist, want_node_max=True)
        monkeypatch.setenv("TOME_SCORES_FILTER", "2")
        got = _abi.match(metric, r, cls, dist, want_node_max=True)
        if want is None:
            assert got is None
            continue
        _same_plan(got, want, f"{kind} r={r} cls={cls} distill={dist}")
        # (oracle comparison: not for NaN tokens -- that rule is pinned by test_nan_rule... -- and not for "tiny": a
        # token whose squared norm underflows has +-inf unit channels, its scores are NaN without its unit vector
        # being NaN, a case outside the NaN-flag contract of both paths alike)
        if kind not in ("nan", "tiny", "large_norms"):  # (large_norms: some squared norms overflow -> norm inf, as "huge")
            plan = oracle.match(host, r, cls, dist)
            np.testing.assert_array_equal(got.src_idx.cpu().numpy(), plan.src_idx)
            np.testing.assert_array_equal(got.dst_idx.cpu().numpy(), plan.dst_idx)
            np.testing.assert_array_equal(got.unm_idx.cpu().numpy(), plan.unm_idx)
            np.testing.assert_array_equal(got.node_max.cpu().numpy().view(np.uint32), plan.node_max.view(np.uint32))


@pytest.mark.parametrize("kind", ["normal", "clustered", "duplicates", "nan"])
@pytest.mark.parametrize("n,H,T", [(3, 12, 197), (2, 12, 1568), (4, 2, 66), (16, 12, 196)])
def test_match_keys_candidate_filter_is_bit_identical(kind, n, H, T, monkeypatch):
    """The production entry (tome_match_keys: head mean inside the kernel, keys read in place from a qkv buffer)
    with the candidate filter forced on vs off: identical indices and node_max bits."""
    from tome import _abi
    qkv = torch.zeros(n, T, 3, H, 64, dtype=torch.bfloat16)
    for h in range(H):
        qkv[:, :, 1, h] = _filter_metric(kind, n, T, 64, 777 + 31 * h + T)
    keys = qkv.to(DEV).permute(2, 0, 3, 1, 4)[1]
    for r, cls in ((16, False), (7, True)):
        monkeypatch.setenv("TOME_SCORES_FILTER", "0")
        want = _abi.match_keys(keys, r, cls, False, want_node_max=True)
        monkeypatch.setenv("TOME_SCORES_FILTER", "2")
        got = _abi.match_keys(keys, r, cls, False, want_node_max=True)
        _same_plan(got, want, f"{kind} r={r} cls={cls}")


def test_match_candidate_filter_runs_by_itself_on_large_launches(monkeypatch):
    """Without the switch the filter serves launches of >= 1024 A tiles (the benchmark's: 384 x 25): same answer as
    the fp32 pass at that size."""
    from tome import _abi
    monkeypatch.delenv("TOME_SCORES_FILTER", raising=False)
    g = torch.Generator(device=DEV).manual_seed(11)
    qkv = torch.randn(48, 1568, 3, 12, 64, device=DEV, generator=g).bfloat16()
    keys = qkv.permute(2, 0, 3, 1, 4)[1]
    got = _abi.match_keys(keys, 16, want_node_max=True)
    monkeypatch.setenv("TOME_SCORES_FILTER", "0")
    want = _abi.match_keys(keys, 16, want_node_max=True)
    _same_plan(got, want, "48 x 1568")


def test_match_candidate_filter_random_campaign():
    """tools/filter_fuzz.py: 150 random shapes / key statistics (scales over 40 binades, low-rank keys with cosines
    packed within 1e-4, duplicated tokens, a nearly constant token) / r / class token / protected column, filter on vs
    off: indices and node_max bits identical.  (600 cases of two other seeds ran clean when the path was built.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "filter_fuzz.py"), "150", "20261005"], capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-2000:])
    assert "150 cases, 0 mismatches" in p.stdout


def test_match_exact_ties_are_stable():
    """Duplicate tokens give exactly equal scores: first maximal column wins, equal node_max keep row
    order (the contract's tie rule; the reference's argsort leaves this undefined)."""
    from tome import _abi
    base = synth.normal_like((2, 8, 64), 99)
    metric = np.tile(base, (1, 12, 1))  # 96 tokens, every token repeated 12 times
    plan = oracle.match(metric, 20, False, False)
    got = _abi.match(dev(metric), 20, False, False, want_node_max=True)
    np.testing.assert_array_equal(got.src_idx.cpu().numpy(), plan.src_idx)
    np.testing.assert_array_equal(got.dst_idx.cpu().numpy(), plan.dst_idx)
    np.testing.assert_array_equal(got.unm_idx.cpu().numpy(), plan.unm_idx)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_match_low_precision_metric_is_upcast(dtype):
    """bf16/fp16 keys are converted to fp32 and matched there: same answer as the fp32 run on the
    rounded values (and as the oracle)."""
    from tome import _abi
    metric = synth.normal_like((3, 196, 64), 31337)
    m16 = dev(metric, dtype)
    rounded = host(m16)
    plan = oracle.match(rounded, 16, False, False)
    got = _abi.match(m16, 16, False, False)
    np.testing.assert_array_equal(got.src_idx.cpu().numpy(), plan.src_idx)
    np.testing.assert_array_equal(got.dst_idx.cpu().numpy(), plan.dst_idx)
    np.testing.assert_array_equal(got.unm_idx.cpu().numpy(), plan.unm_idx)


def test_match_strided_metric_view():
    """timesformer.py:83 hands over `k.mean(1)[:, 1:, :]` -- a view with a token offset."""
    from tome import _abi
    full = synth.normal_like((4, 197, 64), 555)
    view = dev(full)[:, 1:, :]
    assert not view.is_contiguous()
    plan = oracle.match(full[:, 1:, :], 16, False, False)
    got = _abi.match(view, 16, False, False)
    np.testing.assert_array_equal(got.src_idx.cpu().numpy(), plan.src_idx)
    np.testing.assert_array_equal(got.unm_idx.cpu().numpy(), plan.unm_idx)


# ------------------------------------------------------------------------------------------------
# merge values
# ------------------------------------------------------------------------------------------------
def _merge_for(case, fn="bipartite_soft_matching", **kw):
    tm = _tome()
    res = getattr(tm, fn)(dev(G.metric_of(case)), case["r"], case["cls"], case["distill"], **kw)
    return tm, res


@pytest.mark.parametrize("case", G.value_cases("wavg"), ids=lambda c: c["id"])
def test_merge_wavg_fp32_bit_exact(case):
    tm, (merge, _) = _merge_for(case)
    size = G.size_of(case)
    xo, so = tm.merge_wavg(merge, dev(G.x_of(case)), None if size is None else dev(size))
    z = G.arrays("values")
    np.testing.assert_array_equal(host(so), z[case["id"] + "_size"])
    np.testing.assert_array_equal(host(xo), z[case["id"] + "_x"])


@pytest.mark.parametrize("case", G.value_cases("wavg")[:8], ids=lambda c: c["id"])
@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 2 ** -7), (torch.float16, 2 ** -10)])
def test_merge_wavg_low_precision(case, dtype, tol):
    """bf16/fp16 tokens: the kernel accumulates in fp32 and rounds once, the reference rounds after every
    op; stated tolerance: |diff| <= tol * max(1, |ref|) with tol = one unit of the format's epsilon."""
    tm, (merge, _) = _merge_for(case)
    x16 = dev(G.x_of(case), dtype)
    size = G.size_of(case)
    s16 = None if size is None else dev(size, dtype)
    xo, so = tm.merge_wavg(merge, x16, s16)
    assert xo.dtype == dtype and so.dtype == dtype
    plan = oracle.match(G.metric_of(case), case["r"], case["cls"], case["distill"])
    want_x, want_s = oracle.merge_wavg(plan, host(x16), None if size is None else size)
    np.testing.assert_array_equal(host(so), want_s)  # small integers: exact in both formats
    # exactly the oracle's fp32 result rounded once
    np.testing.assert_array_equal(host(xo), host(torch.from_numpy(want_x).to(dtype)))
    ref = G.arrays("values")[case["id"] + "_x"]
    assert np.all(np.abs(host(xo) - ref) <= 2 * tol * np.maximum(1.0, np.abs(ref)))


@pytest.mark.parametrize("case", G.value_cases("merge"), ids=lambda c: c["id"] + c["mode"])
def test_merge_modes(case):
    tm, (merge, _) = _merge_for(case)
    out = merge(dev(G.x_of(case)), mode=case["mode"])
    np.testing.assert_array_equal(host(out), G.arrays("values")[case["id"] + "_x"])


@pytest.mark.parametrize("case", G.value_cases("unmerge"), ids=lambda c: c["id"])
def test_unmerge(case):
    tm, (merge, unmerge) = _merge_for(case)
    merged = merge(dev(G.x_of(case)), mode="mean")
    z = G.arrays("values")
    np.testing.assert_array_equal(host(merged), z[case["id"] + "_merged"])
    np.testing.assert_array_equal(host(unmerge(merged)), z[case["id"] + "_x"])


@pytest.mark.parametrize("case", G.value_cases("drop"), ids=lambda c: c["id"])
def test_drop(case):
    tm, drop = _merge_for(case, "bipartite_soft_matching_drop")
    np.testing.assert_array_equal(host(drop(dev(G.x_of(case)))), G.arrays("values")[case["id"] + "_x"])
    cv = closure_vars(drop)
    np.testing.assert_array_equal(cv["und_idx"].cpu().numpy()[..., 0], G.arrays("values")[case["id"] + "_unm"])


@pytest.mark.parametrize("case", G.value_cases("hybrid") + G.value_cases("hybrid_merge"), ids=lambda c: c["id"])
def test_hybrid(case):
    tm, (merge, _) = _merge_for(case, "bipartite_soft_matching_hybrid", mode="hybrid", threshold=case["threshold"])
    z = G.arrays("values")
    if case["op"] == "hybrid":
        size = G.size_of(case)
        xo, so = tm.merge_wavg(merge, dev(G.x_of(case)), None if size is None else dev(size))
        np.testing.assert_array_equal(host(so), z[case["id"] + "_size"])
        np.testing.assert_array_equal(host(xo), z[case["id"] + "_x"])
    else:
        np.testing.assert_array_equal(host(merge(dev(G.x_of(case)), mode=case["mode"])), z[case["id"] + "_x"])


@pytest.mark.parametrize("case", G.value_cases("source"), ids=lambda c: c["id"])
def test_merge_source_two_layers(case):
    tm, (merge, _) = _merge_for(case)
    n, T = case["n"], case["T"]
    z = G.arrays("values")
    x = torch.zeros(n, T, 4, device=DEV)
    s1 = tm.merge_source(merge, x, None)
    np.testing.assert_array_equal(host(s1).astype(np.uint8), z[case["id"] + "_s1"])
    m2 = G.metric2_of(case, case["r_eff"])
    merge2, _ = tm.bipartite_soft_matching(dev(m2), case["r"], case["cls"], case["distill"])
    s2 = tm.merge_source(merge2, torch.zeros(n, T - case["r_eff"], 4, device=DEV), s1)
    np.testing.assert_array_equal(host(s2).astype(np.uint8), z[case["id"] + "_s2"])


@pytest.mark.parametrize("n,T,r,cls,distill", [(2, 40, 8, False, False), (3, 197, 16, True, False), (1, 1568, 150, False, False),
                                               (2, 65, 40, False, False), (2, 50, 7, True, True), (1, 3, 1, False, False)])
def test_merge_source_first_layer_without_identity(n, T, r, cls, distill):
    """merge_source(source=None) writes the one-hot rows from the matching's row map (tome_source_init); it must
    equal what the reference does -- the identity merged with mode "max" (merge.py:372-384) -- element for element,
    whether the row map came from the matching kernel or is rebuilt from the index tensors (tome_row_map); same for
    the drop closure applied to the identity (tome/patch/videomae.py:112-117)."""
    from tome import _abi
    tm = _tome()
    metric = dev(synth.normal_like((n, T, 16), 9000 + T + r))
    x = torch.zeros(n, T, 1, device=DEV)
    eye = torch.eye(T, device=DEV)[None].expand(n, T, T).contiguous()
    merge, _ = tm.bipartite_soft_matching(metric, r, cls, distill)
    want = merge(eye, mode="max")
    got = tm.merge_source(merge, x, None)
    assert got.dtype == torch.float32 and torch.equal(got, want)
    assert torch.equal(got.sum(1), torch.ones(n, T, device=DEV))  # every token is in exactly one merged row
    plan2 = _abi.match(metric, r, cls, distill, want_row_map=True)
    assert torch.equal(plan2.row_map, merge.plan.row_map)  # tome_row_map == what k_rank_select writes
    assert torch.equal(_abi.source_init(plan2), want)
    drop = tm.bipartite_soft_matching_drop(metric, r, cls, distill)
    assert torch.equal(_abi.source_init(drop.plan, drop=True), drop(eye))
    before = torch.cuda.max_memory_allocated()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    tm.merge_source(merge, x, None)
    peak = torch.cuda.max_memory_allocated() - base
    assert peak < 1.5 * want.numel() * 4 + (1 << 20), (peak, want.numel() * 4, before)  # no [n,T,T] identity beside the result


def test_merge_source_first_layer_hybrid_keeps_killed_destinations():
    """A hybrid matching (merge.py:274-352) zeroes a destination whose incoming edge scores below the threshold
    before the amax (merge.py:326-331): in the first layer's source matrix that destination's OWN column is 0.
    merge_source(source=None) must reproduce that (== merge(eye, "max") of the same closure) -- the one-hot shortcut
    of the plain matching would write a 1 there.  The threshold sits inside the selected edges' scores, so some
    destinations are killed and some are not (asserted)."""
    tm = _tome()
    n, T, r = 3, 96, 20
    metric = dev(synth.normal_like((n, T, 16), 4242))
    plain, _ = tm.bipartite_soft_matching(metric, r)
    nm = plain.plan.node_max if getattr(plain.plan, "node_max", None) is not None else None
    if nm is None:
        from tome import _abi
        nm = _abi.match(metric, r, False, False, want_node_max=True).node_max
    sel = torch.sort(nm, dim=1, descending=True).values[:, :r]
    thr = float(sel[:, r // 2].mean())  # about half of the selected edges lie below it
    merge, _ = tm.bipartite_soft_matching_hybrid(metric, r, threshold=thr)
    assert merge.plan.edge_keep is not None
    keep = merge.plan.edge_keep.bool()
    assert bool(keep.any()) and bool((~keep).any()), "threshold does not split the selected edges"
    eye = torch.eye(T, device=DEV)[None].expand(n, T, T).contiguous()
    want = merge(eye, mode="max")
    got = tm.merge_source(merge, torch.zeros(n, T, 1, device=DEV), None)
    assert torch.equal(got, want)
    # and it differs from the plain matching's source exactly in the killed destinations' own columns
    assert float(want.sum()) < float(tm.merge_source(plain, torch.zeros(n, T, 1, device=DEV), None).sum())


def test_random_merge_uses_given_scores():
    """random_merge / random_drop draw torch.rand scores (merge.py:54-57); the selection from a given
    score matrix must equal the oracle's."""
    from tome import _abi
    n, T = 3, 197
    scores = synth.uniform01((n, (T + 1) // 2, T // 2), 77)
    for cls in (False, True):
        plan = oracle.match_scores(scores, T, 24, cls, False)
        got = _abi.match_scores(dev(scores), T, 24, cls, False, want_node_max=True)
        np.testing.assert_array_equal(got.src_idx.cpu().numpy(), plan.src_idx)
        np.testing.assert_array_equal(got.dst_idx.cpu().numpy(), plan.dst_idx)
        np.testing.assert_array_equal(got.unm_idx.cpu().numpy(), plan.unm_idx)
        np.testing.assert_array_equal(got.node_max.cpu().numpy(), plan.node_max)
    tm = _tome()
    torch.manual_seed(0)
    merge, _ = tm.bipartite_soft_matching(torch.zeros(n, T, 8, device=DEV), 24, mode="random_merge")
    out = merge(torch.ones(n, T, 8, device=DEV), mode="sum")
    assert out.shape == (n, T - 24, 8) and float(out.sum()) == n * T * 8


# ------------------------------------------------------------------------------------------------
# full-size properties (BASELINE.json sizes; no oracle needed)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,T,C,r,cls,dtype", [(8, 1568, 768, 16, False, torch.bfloat16),
                                                (8, 1568, 768, 16, False, torch.float32),
                                                (64, 196, 768, 16, False, torch.bfloat16),
                                                (2, 3137, 768, 64, True, torch.float32),
                                                (8, 784, 768, 392, False, torch.float32)])
def test_full_size_properties(n, T, C, r, cls, dtype):
    """Token-count conservation (sum of sizes == T0), index permutation property, unmerge o merge
    reproduces untouched tokens, and a 12-layer chain keeps conserving."""
    tm = _tome()
    g = torch.Generator(device=DEV).manual_seed(1234)
    x = torch.randn(n, T, C, device=DEV, generator=g).to(dtype)
    size = None
    T0 = T
    for layer in range(12):
        Tcur = x.shape[1]
        metric = torch.randn(n, Tcur, 64, device=DEV, generator=g).to(dtype)
        merge, unmerge = tm.bipartite_soft_matching(metric, r, cls)
        if merge is tm.do_nothing:
            break
        cv = closure_vars(merge)
        re = cv["r"]
        T1 = (Tcur + 1) // 2
        both = torch.cat([cv["src_idx"], cv["unm_idx"]], 1)[..., 0].sort(1).values
        assert torch.equal(both, torch.arange(T1, device=DEV).expand(n, T1))
        assert int(cv["dst_idx"].min()) >= 0 and int(cv["dst_idx"].max()) < Tcur // 2
        if cls:
            assert bool((cv["unm_idx"][:, 0, 0] == 0).all())  # class token stays first
            assert bool((cv["unm_idx"][:, 1:, 0] > cv["unm_idx"][:, :-1, 0]).all())
        x_new, size = tm.merge_wavg(merge, x, size)
        assert x_new.shape == (n, Tcur - re, C)
        assert torch.equal(size.float().sum(1), torch.full((n, 1), float(T0), device=DEV))
        if layer == 0:
            # rows that were not touched come back bit-identical through unmerge(merge_sum(x))
            back = unmerge(merge(x, mode="sum"))
            keep = torch.ones(n, Tcur, dtype=torch.bool, device=DEV)
            keep.scatter_(1, 2 * cv["src_idx"][..., 0], False)
            keep.scatter_(1, 2 * cv["dst_idx"][..., 0] + 1, False)
            assert torch.equal(back[keep], x[keep])
        x = x_new
    assert torch.isfinite(x.float()).all()


def test_errors_are_loud():
    tm = _tome()
    from tome._abi import TomeHipError
    with pytest.raises(TomeHipError):
        tm.bipartite_soft_matching(torch.randn(2, 16, 8), 4)  # CPU tensor: no CPU path
    merge, _ = tm.bipartite_soft_matching(torch.randn(2, 16, 8, device=DEV), 4)
    with pytest.raises(TomeHipError):
        merge(torch.randn(2, 15, 8, device=DEV))  # wrong token count
    with pytest.raises(TomeHipError):
        merge(torch.randn(2, 16, 8, device=DEV), mode="median")
    # more misuse that must be refused on the host, never reach a kernel with wrong shapes
    from tome import _abi
    with pytest.raises(TomeHipError):
        merge(torch.randn(3, 16, 8, device=DEV))  # wrong number of groups
    with pytest.raises(TomeHipError):
        merge(torch.randn(2, 16, device=DEV))  # not [n, T, C]
    with pytest.raises(TomeHipError):
        merge(torch.randint(0, 5, (2, 16, 8), device=DEV))  # integer tokens
    with pytest.raises(TomeHipError):
        tm.merge_wavg(merge, torch.randn(2, 16, 8, device=DEV), torch.ones(2, 15, 1, device=DEV))  # size shape
    # tokens that require grad: the closures take the framework's differentiable ops (tome/merge.py, round 4); the
    # kernels themselves, called directly, still refuse them
    xg = torch.randn(2, 16, 8, device=DEV, requires_grad=True)
    assert tm.merge_wavg(merge, xg)[0].requires_grad
    with pytest.raises(TomeHipError):
        _abi.merge_wavg(merge.plan, xg, None)
    with pytest.raises(TomeHipError):
        tm.bipartite_soft_matching(torch.randn(2, 16, device=DEV), 4)  # metric not [n, T, D]
    with pytest.raises(TomeHipError):
        tm.bipartite_soft_matching(torch.randn(2, 16, 8, device=DEV).double(), 4)  # fp64 keys
    w = torch.ones(8, device=DEV, dtype=torch.bfloat16)
    xb = torch.randn(2, 16, 8, device=DEV).bfloat16()
    with pytest.raises(TomeHipError):
        _abi.merge_wavg_ln(merge.plan, xb, None, w, w, 1e-6, addend=xb[:, :8])  # addend shape
    with pytest.raises(TomeHipError):
        _abi.merge_wavg_regrouped(merge.plan, torch.randn(1, 1 + 16 * 3, 8, device=DEV), None, 3)  # 2 groups != 1*3
    with pytest.raises(TomeHipError):
        _abi.drop_regrouped(merge.plan, torch.randn(1, 16 * 2, 8, device=DEV), 2)  # class token missing
    with pytest.raises(TomeHipError):
        _abi.add_layernorm(xb, xb[:, :, :4], w, w, 1e-6)
    # non-contiguous inputs are accepted (made contiguous), with the same result
    xt = torch.randn(2, 8, 16, device=DEV).transpose(1, 2)
    assert torch.equal(merge(xt), merge(xt.contiguous()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,F,P,C,r", [(2, 4, 36, 32, 6), (3, 8, 196, 768, 16), (1, 8, 49, 64, 24), (2, 2, 9, 8, 4)])
def test_merge_wavg_regrouped_equals_rearranged(B, F, P, C, r, dtype):
    """tome_merge_wavg_regrouped == rearrange -> merge_wavg -> rearrange -> cat (timesformer.py:89-107,
    motionformer.py:150-168), bit for bit, including chained sizes."""
    from tome import _abi
    tm = _tome()
    seed = 900 + B * F + P
    x_full = dev(synth.normal_like((B, 1 + P * F, C), seed), dtype)
    size = None
    Pc = P
    for layer in range(2):
        metric = dev(synth.normal_like((B * F, Pc, 16), seed + 17 * layer))
        merge, _ = tm.bipartite_soft_matching(metric, r)
        plan = merge.plan
        grouped = x_full[:, 1:, :].reshape(B, Pc, F, C).transpose(1, 2).reshape(B * F, Pc, C)
        want_x, want_s = tm.merge_wavg(merge, grouped, size)
        P2 = Pc - plan.r
        want_full = torch.cat((x_full[:, :1, :], want_x.reshape(B, F, P2, C).transpose(1, 2).reshape(B, P2 * F, C)), 1)
        got_full, got_s = _abi.merge_wavg_regrouped(plan, x_full, size, F, has_cls=True)
        assert torch.equal(got_full, want_full)
        assert torch.equal(got_s, want_s)
        x_full, size, Pc = got_full, got_s, P2


@pytest.mark.parametrize("cls", [False, True])
@pytest.mark.parametrize("where", ["a_row", "b_row", "both", "b_first_and_later", "inf_value"])
def test_nan_semantics_zero_tokens(where, cls):
    """A zero token has a NaN unit vector (merge.py:51 divides by the norm without epsilon): every score it
    takes part in is NaN, torch.max returns NaN with the FIRST NaN column, argsort puts NaN rows first.
    The kernels ignore NaN in the MFMA pass and restore these semantics from per-token flags; the oracle
    implements torch's rule directly."""
    from tome import _abi
    n, T, D = 3, 70, 64
    metric = synth.normal_like((n, T, D), 2718).copy()
    if where in ("a_row", "both"):
        metric[0, 2 * 7] = 0.0          # A row 7 of group 0
        metric[2, 2 * 30] = 0.0
    if where in ("b_row", "both"):
        metric[1, 2 * 5 + 1] = 0.0      # B row 5 of group 1
    if where == "b_first_and_later":
        metric[0, 2 * 20 + 1] = 0.0
        metric[0, 2 * 3 + 1] = 0.0      # first bad column is 3
    if where == "inf_value":
        metric[1, 2 * 9, 5] = np.inf    # inf / inf = NaN in one channel
    for r in (4, 35):
        plan = oracle.match(metric, r, cls, False)
        got = _abi.match(dev(metric), r, cls, False, want_node_max=True)
        np.testing.assert_array_equal(got.src_idx.cpu().numpy(), plan.src_idx)
        np.testing.assert_array_equal(got.dst_idx.cpu().numpy(), plan.dst_idx)
        np.testing.assert_array_equal(got.unm_idx.cpu().numpy(), plan.unm_idx)
        gm, om = got.node_max.cpu().numpy(), plan.node_max
        np.testing.assert_array_equal(np.isnan(gm), np.isnan(om))
        np.testing.assert_array_equal(gm[~np.isnan(gm)], om[~np.isnan(om)])
    # and torch agrees with the oracle on which rows are NaN / where they point (reference semantics)
    m = torch.from_numpy(metric)
    u = m / m.norm(dim=-1, keepdim=True)
    s = u[:, ::2] @ u[:, 1::2].transpose(-1, -2)
    if cls:
        s[:, 0, :] = -float("inf")
    tv, ti = s.max(-1)
    plan = oracle.match(metric, 4, cls, False)
    np.testing.assert_array_equal(torch.isnan(tv).numpy(), np.isnan(plan.node_max))
    nanrows = np.isnan(plan.node_max)
    np.testing.assert_array_equal(ti.numpy()[nanrows], plan.node_idx[nanrows])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cls", [False, True])
def test_match_keys_fused_head_mean(dtype, cls):
    """tome_match_keys reads the per-head keys in place (a strided view of a qkv buffer, as the patches hand
    them over) and must equal: head mean by the oracle's rule (fp32 sum in head order / H, one rounding to the
    keys' dtype -- bit-identical to torch's CPU `k.mean(1)`), then the ordinary matching."""
    from tome import _abi
    from tome.merge import HeadMeanKeys
    tm = _tome()
    B, H, N, hd = 3, 12, 197, 64
    qkv = dev(synth.normal_like((B, N, 3, H, hd), 606), dtype)
    k = qkv.permute(2, 0, 3, 1, 4)[1]  # [B,H,N,hd], strides of the patches' key view
    assert not k.is_contiguous() and _abi.keys_fusable(k)
    k_host = host(k)
    mean = oracle.head_mean(k_host)
    if dtype != torch.float32:
        mean = host(torch.from_numpy(mean).to(dtype))
    assert np.array_equal(mean, host(k.cpu().mean(1)))  # the rule IS torch's CPU mean
    for r in (16, 200):
        plan = oracle.match(mean, r, cls, False)
        got = _abi.match_keys(k, r, cls, False, want_node_max=True)
        np.testing.assert_array_equal(got.src_idx.cpu().numpy(), plan.src_idx)
        np.testing.assert_array_equal(got.dst_idx.cpu().numpy(), plan.dst_idx)
        np.testing.assert_array_equal(got.unm_idx.cpu().numpy(), plan.unm_idx)
        np.testing.assert_array_equal(got.node_max.cpu().numpy().view(np.uint32), plan.node_max.view(np.uint32))
        # through the public API, and with the class-token slice TimeSformer takes
        merge, _ = tm.bipartite_soft_matching(HeadMeanKeys(k), r, cls)
        np.testing.assert_array_equal(merge.plan.src_idx.cpu().numpy(), plan.src_idx)
    sl = k[:, :, 1:, :]
    assert _abi.keys_fusable(sl)
    plan = oracle.match(mean[:, 1:, :], 16, False, False)
    got = _abi.match_keys(sl, 16)
    np.testing.assert_array_equal(got.unm_idx.cpu().numpy(), plan.unm_idx)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_match_keys_interleaved_groups(dtype):
    """Motionformer's metric (tome/patch/motionformer.py:143-144): keys regrouped '(b h) (s f) d -> (b f) h s d' and
    averaged over the heads.  tome_match_keys takes the regrouped keys as a strided [b, f, h, s, d] VIEW of the qkv
    buffer (inner groups) and must give what the materialised metric gives: the head mean by torch's CPU rule,
    then the ordinary matching, bit for bit against the oracle."""
    from einops import rearrange
    from tome import _abi
    from tome.merge import HeadMeanKeys
    tm = _tome()
    B, H, P, F, hd = 2, 12, 49, 8, 64
    N = 1 + P * F
    qkv = dev(synth.normal_like((B, N, 3, H, hd), 717), dtype)
    k = qkv.permute(2, 0, 3, 1, 4)[1]  # [B, H, N, hd]
    view = k[:, :, 1:, :].unflatten(2, (P, F)).permute(0, 3, 1, 2, 4)  # [b, f, h, s, d]
    assert view.shape == (B, F, H, P, hd) and _abi.keys_fusable(view)
    # the reference's own expression, on CPU
    k_ = rearrange(k.cpu()[:, :, 1:, :], "b h n d -> (b h) n d")
    metric = rearrange(k_, "(b h) (s f) d -> (b f) h s d", f=F, h=H).mean(1)
    hm = HeadMeanKeys(view)
    # (.materialize() is the framework's GPU mean: same values up to its summation order)
    assert hm.shape == (B * F, P, hd) and torch.allclose(hm.materialize().cpu().float(), metric.float(), atol=1e-2)
    for r in (5, 16, 40):
        plan = oracle.match(host(metric), r, False, False)
        got = _abi.match_keys(view, r, want_node_max=True)
        np.testing.assert_array_equal(got.src_idx.cpu().numpy(), plan.src_idx)
        np.testing.assert_array_equal(got.dst_idx.cpu().numpy(), plan.dst_idx)
        np.testing.assert_array_equal(got.unm_idx.cpu().numpy(), plan.unm_idx)
        np.testing.assert_array_equal(got.node_max.cpu().numpy().view(np.uint32), plan.node_max.view(np.uint32))
        merge, _ = tm.bipartite_soft_matching(hm, r)
        np.testing.assert_array_equal(merge.plan.src_idx.cpu().numpy(), plan.src_idx)


def test_empty_and_degenerate_inputs():
    """Empty batch, single token, r larger than anything, r = 0: the do_nothing pair, as merge.py:46-47."""
    tm = _tome()
    for shape, r in [((0, 16, 8), 4), ((2, 1, 8), 4), ((2, 16, 8), 0), ((2, 16, 8), -3), ((2, 1, 8), 0)]:
        merge, unmerge = tm.bipartite_soft_matching(torch.zeros(shape, device=DEV), r)
        assert merge is tm.do_nothing and unmerge is tm.do_nothing
        x = torch.randn(shape[0], shape[1], 4, device=DEV)
        xo, so = tm.merge_wavg(merge, x)
        assert xo.shape == x.shape and so.shape == (shape[0], shape[1], 1)
    # class token alone protects the only pair: T=2 with cls -> (2-1)//2 = 0
    merge, _ = tm.bipartite_soft_matching(torch.randn(2, 2, 8, device=DEV), 5, class_token=True)
    assert merge is tm.do_nothing
    # T=3 with cls: one merge possible, class token (A row 0) must survive
    merge, _ = tm.bipartite_soft_matching(torch.randn(2, 3, 8, device=DEV), 5, class_token=True)
    p = merge.plan
    assert p.r == 1 and bool((p.src_idx == 1).all()) and bool((p.unm_idx == 0).all())


def _fuzz_cases(count, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        n = int(rng.choice([1, 2, 3, 5, 9, 17]))
        T = int(rng.choice([2, 3, 7, 31, 32, 33, 63, 64, 65, 66, 129, 130, 195, 196, 197, 392, 500, 784]))
        D = int(rng.choice([8, 16, 24, 64, 64, 64, 72, 128, 192]))
        C = int(rng.choice([1, 3, 8, 12, 64, 96, 768, 772]))
        r = int(rng.choice([1, 2, 5, 16, 17, 63, 64, 65, 100, 10 ** 6]))
        cls = bool(rng.integers(0, 2))
        dtype = [torch.float32, torch.bfloat16, torch.float16][int(rng.integers(0, 3))]
        sized = bool(rng.integers(0, 2))
        kind = "clustered" if rng.integers(0, 3) == 0 else "normal"
        out.append((n, T, D, C, r, cls, dtype, sized, kind, int(rng.integers(1, 1 << 30))))
    return out


@pytest.mark.parametrize("case", _fuzz_cases(48, 20260104), ids=lambda c: f"n{c[0]}T{c[1]}D{c[2]}C{c[3]}r{c[4]}")
def test_fuzz_against_oracle(case):
    """Seeded random shapes (ragged tiles, D and C off every grid, r from 1 to 'everything', class token,
    three dtypes, tie-heavy keys): matching bit-exact vs the oracle, merge_wavg / sum / amax / unmerge equal
    to the oracle's fp32 result rounded once to the tensor dtype."""
    from tome import _abi
    n, T, D, C, r, cls, dtype, sized, kind, seed = case
    tm = _tome()
    metric = (synth.clustered if kind == "clustered" else synth.normal_like)((n, T, D), seed)
    m_dev = dev(metric, dtype)
    m_host = host(m_dev)
    plan = oracle.match(m_host, r, cls, False)
    merge, unmerge = tm.bipartite_soft_matching(m_dev, r, cls)
    if plan is None:
        assert merge is tm.do_nothing
        return
    p = merge.plan
    np.testing.assert_array_equal(p.src_idx.cpu().numpy(), plan.src_idx)
    np.testing.assert_array_equal(p.dst_idx.cpu().numpy(), plan.dst_idx)
    np.testing.assert_array_equal(p.unm_idx.cpu().numpy(), plan.unm_idx)
    x = dev(synth.normal_like((n, T, C), seed ^ 0x77), dtype)
    size = dev(synth.small_ints((n, T, 1), seed ^ 0x99, 1, 6), dtype) if sized else None
    xo, so = tm.merge_wavg(merge, x, size)
    want_x, want_s = oracle.merge_wavg(plan, host(x), None if size is None else host(size))
    assert torch.equal(xo.cpu(), torch.from_numpy(want_x).to(dtype)), "merge_wavg x"
    assert torch.equal(so.cpu(), torch.from_numpy(want_s).to(dtype)), "merge_wavg size"
    for mode in ("sum", "amax"):
        got = merge(x, mode=mode)
        assert torch.equal(got.cpu(), torch.from_numpy(oracle.merge(plan, host(x), mode)).to(dtype)), mode
    back = unmerge(xo)
    assert torch.equal(back.cpu(), torch.from_numpy(oracle.unmerge(plan, host(xo))).to(dtype)), "unmerge"


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 2 ** -7), (torch.float16, 2 ** -10)])
@pytest.mark.parametrize("n,T,C,r,cls", [(3, 197, 768, 16, True), (2, 64, 64, 30, False), (4, 392, 1024, 150, False)])
def test_merge_wavg_ln_fused(n, T, C, r, cls, dtype, tol):
    """tome_merge_wavg_ln: x_out bit-identical to tome_merge_wavg, y_out = LayerNorm(x_out) within one unit of
    the format's epsilon of an fp32 LayerNorm of the same stored tokens (the block's norm2)."""
    from tome import _abi
    tm = _tome()
    seed = 31 * n + T + C
    metric = dev(synth.normal_like((n, T, 64), seed))
    x = dev(synth.normal_like((n, T, C), seed + 1), dtype)
    size = dev(synth.small_ints((n, T, 1), seed + 2, 1, 4), dtype)
    w = dev(1.0 + 0.1 * synth.normal_like((C,), seed + 3), dtype)
    b = dev(0.1 * synth.normal_like((C,), seed + 4), dtype)
    merge, _ = tm.bipartite_soft_matching(metric, r, cls)
    want_x, want_s = tm.merge_wavg(merge, x, size)
    got_x, got_y, got_s = _abi.merge_wavg_ln(merge.plan, x, size, w, b, 1e-6)
    assert torch.equal(got_x, want_x) and torch.equal(got_s, want_s)
    ref = torch.nn.functional.layer_norm(want_x.float(), (C,), w.float(), b.float(), 1e-6)
    err = (got_y.float() - ref).abs()
    assert float((err / ref.abs().clamp(min=1.0)).max()) <= tol, float(err.max())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n,T,C,r,cls", [(3, 197, 768, 16, True), (2, 64, 64, 30, False), (2, 392, 768, 150, False)])
def test_merge_wavg_ln_fused_residual(n, T, C, r, cls, dtype):
    """tome_merge_wavg_ln with an addend == the same call on the pre-added tokens (torch's `x + a`, rounded to
    the dtype), bit for bit, for x', y and sizes."""
    from tome import _abi
    tm = _tome()
    seed = 77 * n + T + C
    metric = dev(synth.normal_like((n, T, 64), seed))
    x = dev(synth.normal_like((n, T, C), seed + 1), dtype)
    a = dev(0.5 * synth.normal_like((n, T, C), seed + 5), dtype)
    size = dev(synth.small_ints((n, T, 1), seed + 2, 1, 4), dtype)
    w = dev(1.0 + 0.1 * synth.normal_like((C,), seed + 3), dtype)
    b = dev(0.1 * synth.normal_like((C,), seed + 4), dtype)
    merge, _ = tm.bipartite_soft_matching(metric, r, cls)
    want = _abi.merge_wavg_ln(merge.plan, x + a, size, w, b, 1e-6)
    got = _abi.merge_wavg_ln(merge.plan, x, size, w, b, 1e-6, addend=a)
    for g_, w_ in zip(got, want):
        assert torch.equal(g_, w_)
    # x_out_bias: x' leaves the kernel as torch's `x' + bias` (one rounding), y and the sizes are those of x' itself
    ob = dev(0.3 * synth.normal_like((C,), seed + 6), dtype)
    bx, by, bs = _abi.merge_wavg_ln(merge.plan, x, size, w, b, 1e-6, addend=a, out_bias=ob)
    assert torch.equal(bx, want[0] + ob) and torch.equal(by, want[1]) and torch.equal(bs, want[2])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,F,P,C,r", [(2, 4, 36, 64, 6), (2, 8, 196, 768, 16), (1, 8, 49, 1024, 24)])
def test_merge_wavg_regrouped_ln(B, F, P, C, r, dtype):
    """tome_merge_wavg_regrouped_ln (residual + merge on the interleaved layout + LayerNorm, class-token rows
    included) == add, tome_merge_wavg_regrouped, then an fp32 LayerNorm of the stored tokens."""
    from tome import _abi
    tm = _tome()
    seed = 1300 + B * F + P
    x_full = dev(synth.normal_like((B, 1 + P * F, C), seed), dtype)
    res = dev(0.5 * synth.normal_like((B, 1 + P * F, C), seed + 9), dtype)
    size = dev(synth.small_ints((B * F, P, 1), seed + 2, 1, 4), dtype)
    w = dev(1.0 + 0.1 * synth.normal_like((C,), seed + 3), dtype)
    b = dev(0.1 * synth.normal_like((C,), seed + 4), dtype)
    metric = dev(synth.normal_like((B * F, P, 16), seed + 5))
    merge, _ = tm.bipartite_soft_matching(metric, r)
    plan = merge.plan
    want_x, want_s = _abi.merge_wavg_regrouped(plan, x_full + res, size, F, has_cls=True)
    got_x, got_y, got_s = _abi.merge_wavg_regrouped(plan, x_full, size, F, has_cls=True, ln=(w, b, 1e-6), addend=res)
    assert torch.equal(got_x, want_x) and torch.equal(got_s, want_s)
    ref = torch.nn.functional.layer_norm(want_x.float(), (C,), w.float(), b.float(), 1e-6)
    tol = 2 ** -7 if dtype == torch.bfloat16 else 2 ** -10
    assert float(((got_y.float() - ref).abs() / ref.abs().clamp(min=1.0)).max()) <= tol
    # the same residual handed over where TimeSformer's spatial attention leaves it -- [(b f), 1 + p, C] with a class
    # row per frame (ignored) plus the class tokens' own addend [B, 1, C] -- must give the same bits as the
    # '(b f) p -> b (p f)' rearranged + concatenated tensor (tome/patch/timesformer.py:40-52)
    from einops import rearrange
    grouped = torch.cat((dev(synth.normal_like((B * F, 1, C), seed + 11), dtype),
                         rearrange(res[:, 1:], "b (p f) c -> (b f) p c", f=F)), 1).contiguous()
    alt = _abi.merge_wavg_regrouped(plan, x_full, size, F, has_cls=True, ln=(w, b, 1e-6), addend_grouped=grouped,
                                    cls_addend=res[:, :1].contiguous())
    for g_, w_ in zip(alt, (got_x, got_y, got_s)):
        assert torch.equal(g_, w_)
    # x_out_bias on the interleaved layout, class-token rows included
    ob = dev(0.3 * synth.normal_like((C,), seed + 6), dtype)
    for kw in (dict(addend=res), dict(addend_grouped=grouped, cls_addend=res[:, :1].contiguous())):
        bx, by, bs = _abi.merge_wavg_regrouped(plan, x_full, size, F, has_cls=True, ln=(w, b, 1e-6), out_bias=ob, **kw)
        assert torch.equal(bx, got_x + ob) and torch.equal(by, got_y) and torch.equal(bs, got_s)


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 2 ** -7), (torch.float16, 2 ** -10)])
@pytest.mark.parametrize("shape", [(3, 197, 768), (5, 64), (2, 1001, 1024), (7, 8)])
def test_add_layernorm(shape, dtype, tol):
    """tome_add_layernorm: x_out == torch's x + a bit for bit, y_out == LayerNorm(x_out) within one epsilon."""
    from tome import _abi
    C = shape[-1]
    x = dev(synth.normal_like(shape, 11 + C), dtype)
    a = dev(0.5 * synth.normal_like(shape, 12 + C), dtype)
    w = dev(1.0 + 0.1 * synth.normal_like((C,), 13), dtype)
    b = dev(0.1 * synth.normal_like((C,), 14), dtype)
    xo, yo = _abi.add_layernorm(x, a, w, b, 1e-6)
    assert torch.equal(xo, x + a)
    ref = torch.nn.functional.layer_norm((x + a).float(), (C,), w.float(), b.float(), 1e-6)
    assert float(((yo.float() - ref).abs() / ref.abs().clamp(min=1.0)).max()) <= tol
    if len(shape) == 3:
        # skip_first: the same sums, the LayerNorm output without every clip's first (class) row, compacted -- the
        # bits of yo[:, 1:] (TimeSformer's temporal_norm1 hand-over, tome/patch/timesformer.py:24-26)
        xs, ys = _abi.add_layernorm(x, a, w, b, 1e-6, skip_first=True)
        assert torch.equal(xs, xo) and ys.is_contiguous() and torch.equal(ys, yo[:, 1:])
        _, yn = _abi.add_layernorm(xo, None, w, b, 1e-6, skip_first=True)
        assert torch.equal(yn, ys)
    # without an addend: the LayerNorm of x as it is (same bits as the fused form produces for that sum), x untouched
    keep = xo.clone()
    xn, yn = _abi.add_layernorm(xo, None, w, b, 1e-6)
    assert xn is xo and torch.equal(xo, keep) and torch.equal(yn, yo)


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 2 ** -7), (torch.float16, 2 ** -10)])
@pytest.mark.parametrize("B,F,P,C", [(2, 8, 196, 768), (3, 4, 36, 128), (1, 2, 1, 64), (2, 3, 50, 1024), (5, 1, 7, 8)])
def test_add_layernorm_regrouped(B, F, P, C, dtype, tol):
    """tome_add_layernorm_regrouped == TimeSformer's mid-block steps (tome/patch/timesformer.py:24-38): the temporal
    residual (class token untouched), 'b (p t) m -> (b t) p m' with the class token in front of every frame, norm1.
    x1 bit-identical to torch's add; y within one epsilon of an fp32 LayerNorm of the regrouped stored tokens."""
    from einops import rearrange
    from tome import _abi
    seed = 4400 + B * F + P + C
    x = dev(synth.normal_like((B, 1 + P * F, C), seed), dtype)
    a = dev(0.5 * synth.normal_like((B, P * F, C), seed + 1), dtype)
    w = dev(1.0 + 0.1 * synth.normal_like((C,), seed + 2), dtype)
    b = dev(0.1 * synth.normal_like((C,), seed + 3), dtype)
    x1, y = _abi.add_layernorm_regrouped(x, a, F, w, b, 1e-6)
    cls0, xt = x[:, :1], x[:, 1:] + a
    assert torch.equal(x1, torch.cat((cls0, xt), 1))
    xs = torch.cat((cls0.expand(B, F, C).reshape(B * F, 1, C), rearrange(xt, "b (p t) m -> (b t) p m", t=F)), 1)
    assert y.shape == xs.shape
    ref = torch.nn.functional.layer_norm(xs.float(), (C,), w.float(), b.float(), 1e-6)
    assert float(((y.float() - ref).abs() / ref.abs().clamp(min=1.0)).max()) <= tol


@pytest.mark.parametrize("dtype,offset,step,tol", [(torch.bfloat16, 300.0, 2.0, 2 ** -7), (torch.float16, 1000.0, 0.5, 2 ** -10),
                                                   (torch.float16, 0.0, 0.5, 2 ** -10)])
def test_layernorm_of_rows_far_from_zero(dtype, offset, step, tol):
    """Every fused LayerNorm (streaming rows, rows built by edge waves, the class-token rows of the regrouped form,
    tome_add_layernorm) takes the variance centred on the mean (two passes over the registers): rows whose mean is
    hundreds of standard deviations away from zero normalise like torch's LayerNorm.  E[x^2] - mean^2 in fp32
    loses the variance of the fp16 case (std 0.5 on an offset of 1000: the squares carry 6e-8 * 1e6 = 0.06 of
    absolute error against a variance of 0.25)."""
    from tome import _abi
    tm = _tome()
    C, B, F, P, r = 768, 2, 4, 36, 6
    # rows = offset + small multiples of the format's spacing there, so that they survive storage
    noise = step * np.round(synth.normal_like((B, 1 + P * F, C), 501))
    x_full = dev(offset + noise, dtype)
    res = torch.zeros_like(x_full)
    w = dev(1.0 + 0.1 * synth.normal_like((C,), 3), dtype)
    b = dev(0.1 * synth.normal_like((C,), 4), dtype)
    merge, _ = tm.bipartite_soft_matching(dev(synth.normal_like((B * F, P, 16), 505)), r)
    got_x, got_y, _ = _abi.merge_wavg_regrouped(merge.plan, x_full, None, F, has_cls=True, ln=(w, b, 1e-6), addend=res)
    ref = torch.nn.functional.layer_norm(got_x.float(), (C,), w.float(), b.float(), 1e-6)
    err = ((got_y.float() - ref).abs() / ref.abs().clamp(min=1.0))
    assert float(err.max()) <= tol, (float(err.max()), float(err[:, 0].max()))
    assert float(err[:, 0].max()) <= tol  # the class-token rows on their own
    flat = x_full.reshape(-1, C)
    xo, yo = _abi.add_layernorm(flat, torch.zeros_like(flat), w, b, 1e-6)
    ref2 = torch.nn.functional.layer_norm(xo.float(), (C,), w.float(), b.float(), 1e-6)
    assert float(((yo.float() - ref2).abs() / ref2.abs().clamp(min=1.0)).max()) <= tol
    # contiguous form (VideoMAE / ViViT): streaming rows and rows built by edge waves
    xc = x_full[:, 1:, :].contiguous()
    m2, _ = tm.bipartite_soft_matching(dev(synth.normal_like((B, P * F, 16), 506)), 20)
    gx, gy, _ = _abi.merge_wavg_ln(m2.plan, xc, None, w, b, 1e-6)
    ref3 = torch.nn.functional.layer_norm(gx.float(), (C,), w.float(), b.float(), 1e-6)
    assert float(((gy.float() - ref3).abs() / ref3.abs().clamp(min=1.0)).max()) <= tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n,T,C,r,cls", [(3, 197, 768, 16, True), (2, 64, 24, 30, False), (2, 392, 768, 150, False),
                                         (2, 50, 7, 9, False)])
def test_log_size_emitted_by_merge(n, T, C, r, cls, dtype):
    """Every tome_merge_wavg* entry can emit log(size') for the next block's proportional-attention bias
    (`size.log()`, videomae.py:62-63): the emitted tensor equals torch's `.log()` of the emitted size bit for
    bit (fp32 log rounded to the size dtype), and equals the CPU oracle's size logged in fp32 within 1 ulp of
    fp32 / exactly after rounding to a 16-bit format.  The other outputs do not change."""
    from tome import _abi
    tm = _tome()
    seed = 13 * n + T + C
    metric = dev(synth.normal_like((n, T, 64), seed))
    x = dev(synth.normal_like((n, T, C), seed + 1), dtype)
    size = dev(synth.small_ints((n, T, 1), seed + 2, 1, 40), dtype)
    merge, _ = tm.bipartite_soft_matching(metric, r, cls)
    for s_in in (None, size):
        want_x, want_s = tm.merge_wavg(merge, x, s_in)
        assert getattr(want_s, "_tome_log", None) is None
        got_x, got_s = tm.merge_wavg(merge, x, s_in, log_size=True)
        assert torch.equal(got_x, want_x) and torch.equal(got_s, want_s)
        log = got_s._tome_log
        assert log.shape == got_s.shape and log.dtype == got_s.dtype
        assert torch.equal(log, got_s.log())
        assert _abi.log_of_size(got_s) is log and torch.equal(_abi.log_of_size(want_s), log)
        o_plan = oracle.match(metric, r, cls)
        _, o_s = oracle.merge_wavg(o_plan, x, None if s_in is None else s_in)
        o_st = torch.from_numpy(o_s).to(dtype)  # the stored size
        assert torch.equal(o_st, got_s.cpu())
        if dtype == torch.float32:
            np.testing.assert_allclose(log.cpu().numpy().astype(np.float64), np.log(o_s.astype(np.float64)),
                                       rtol=2.5e-7, atol=1e-7)
        else:
            assert torch.equal(log.cpu(), torch.from_numpy(np.log(o_st.float().numpy())).to(dtype))
    if dtype != torch.float32 and C % 8 == 0:
        w = dev(1.0 + 0.1 * synth.normal_like((C,), seed + 3), dtype)
        b = dev(0.1 * synth.normal_like((C,), seed + 4), dtype)
        a = dev(0.5 * synth.normal_like((n, T, C), seed + 5), dtype)
        base = _abi.merge_wavg_ln(merge.plan, x, size, w, b, 1e-6, addend=a)
        got = _abi.merge_wavg_ln(merge.plan, x, size, w, b, 1e-6, addend=a, log_size=True)
        for g_, w_ in zip(got, base):
            assert torch.equal(g_, w_)
        assert torch.equal(got[2]._tome_log, got[2].log())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_log_size_regrouped_and_all_integer_sizes(dtype):
    """(a) the regrouped entries emit the same log; (b) the kernel's logf equals torch's for every integer token
    size a 16-frame clip can reach (1..1568) -- sizes are sums of ones in every model of the reference."""
    from tome import _abi
    tm = _tome()
    B, F, P, C, r = 2, 8, 196, 768, 16
    metric = dev(synth.normal_like((B * F, P, 64), 5))
    x_full = dev(synth.normal_like((B, 1 + P * F, C), 6), dtype)
    size = dev(synth.small_ints((B * F, P, 1), 7, 1, 9), dtype)
    merge, _ = tm.bipartite_soft_matching(metric, r)
    x0, s0 = _abi.merge_wavg_regrouped(merge.plan, x_full, size, F)
    x1, s1 = _abi.merge_wavg_regrouped(merge.plan, x_full, size, F, log_size=True)
    assert torch.equal(x0, x1) and torch.equal(s0, s1) and torch.equal(s1._tome_log, s1.log())
    if dtype != torch.float32:
        w = dev(1.0 + 0.1 * synth.normal_like((C,), 8), dtype)
        b = dev(0.1 * synth.normal_like((C,), 9), dtype)
        got = _abi.merge_wavg_regrouped(merge.plan, x_full, size, F, ln=(w, b, 1e-6), log_size=True)
        assert torch.equal(got[0], x0) and torch.equal(got[2]._tome_log, s0.log())
    # (b): one group whose token t has size t+1, nothing but the smallest r=1 merge
    T = 1568
    sizes = torch.arange(1, T + 1, dtype=torch.float32).reshape(1, T, 1)
    m2 = dev(synth.normal_like((1, T, 8), 11))
    mg, _ = tm.bipartite_soft_matching(m2, 1)
    xs = dev(synth.normal_like((1, T, 8), 12))
    _, s_out = tm.merge_wavg(mg, xs, sizes.to(DEV), log_size=True)
    assert torch.equal(s_out._tome_log, s_out.log())
    assert float(s_out.sum()) == float(sizes.sum())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,F,P,C,r", [(2, 4, 36, 32, 6), (3, 8, 196, 768, 16), (1, 8, 49, 64, 24), (2, 2, 9, 8, 4)])
def test_drop_regrouped_equals_rearranged(B, F, P, C, r, dtype):
    """tome_drop_regrouped == rearrange -> drop -> rearrange -> cat (timesformer.py:111-131): bit-identical rows,
    class token copied through, and equal to the CPU oracle's drop on the regrouped tokens."""
    from tome import _abi
    tm = _tome()
    metric = dev(synth.normal_like((B * F, P, 16), 400 + P))
    x_full = dev(synth.normal_like((B, 1 + P * F, C), 401 + P), dtype)
    drop = tm.bipartite_soft_matching_drop(metric, r)
    plan = drop.plan
    got = _abi.drop_regrouped(plan, x_full, F)
    cls, body = x_full[:, :1], x_full[:, 1:]
    grouped = body.reshape(B, P, F, C).permute(0, 2, 1, 3).reshape(B * F, P, C).contiguous()
    d = drop(grouped)
    want = torch.cat((cls, d.reshape(B, F, P - plan.r, C).permute(0, 2, 1, 3).reshape(B, (P - plan.r) * F, C)), dim=1)
    assert torch.equal(got, want)
    o_plan = oracle.match(metric, r)
    o = oracle.drop(o_plan, grouped)
    assert np.array_equal(d.float().cpu().numpy(), o)


def test_more_than_2_31_elements():
    """Maximum sizes: a token tensor with more than 2^31 elements (1800 x 1568 x 768 bf16, 4.3 GB) goes through
    match + merge_wavg (+ fused LayerNorm / residual, + add_layernorm) with 64-bit addressing: the result equals
    the same work done in three chunks of 600 groups, each below 2^31 elements."""
    from tome import _abi
    tm = _tome()
    n, T, C, r = 1800, 1568, 768, 16
    assert n * T * C > 2 ** 31
    g = torch.Generator(device=DEV).manual_seed(5)
    metric = torch.randn(n, T, 64, device=DEV, generator=g)
    x = torch.randn(n, T, C, device=DEV, generator=g).bfloat16()
    a = (0.1 * torch.randn(n, T, C, device=DEV, generator=g)).bfloat16()
    w = torch.ones(C, device=DEV, dtype=torch.bfloat16)
    b = torch.zeros(C, device=DEV, dtype=torch.bfloat16)
    merge, _ = tm.bipartite_soft_matching(metric, r)
    xo, so = tm.merge_wavg(merge, x)
    fx, fy, fs = _abi.merge_wavg_ln(merge.plan, x, None, w, b, 1e-6, addend=a)
    ax, ay = _abi.add_layernorm(x, a, w, b, 1e-6)
    assert float(so.float().sum()) == n * T
    for lo in range(0, n, 600):
        sl = slice(lo, lo + 600)
        m2, _ = tm.bipartite_soft_matching(metric[sl].contiguous(), r)
        for name in ("src_idx", "dst_idx", "unm_idx"):
            assert torch.equal(getattr(m2.plan, name), getattr(merge.plan, name)[sl]), name
        xo2, so2 = tm.merge_wavg(m2, x[sl].contiguous())
        assert torch.equal(xo2, xo[sl]) and torch.equal(so2, so[sl])
        gx, gy, gs = _abi.merge_wavg_ln(m2.plan, x[sl].contiguous(), None, w, b, 1e-6, addend=a[sl].contiguous())
        assert torch.equal(gx, fx[sl]) and torch.equal(gy, fy[sl]) and torch.equal(gs, fs[sl])
        bx, by = _abi.add_layernorm(x[sl].contiguous(), a[sl].contiguous(), w, b, 1e-6)
        assert torch.equal(bx, ax[sl]) and torch.equal(by, ay[sl])


def _attn_reference(q, k, v, log_bias, scale, skip):
    """fp32 softmax(q k^T * scale + bias) v -> [B, N, H*D]; the bias as the reference's patches build it."""
    s = (q.float() @ k.float().transpose(-1, -2)) * scale
    if log_bias is not None:
        B, H, N, _ = q.shape
        bias = torch.zeros(B, 1, N, N, device=q.device)
        if skip:
            bias[:, :, 1:, 1:] = log_bias[:, None, None, :]  # timesformer.py:73-74
        else:
            bias[:, :, :, :] = log_bias[:, None, None, :]  # videomae.py:62-63
        s = s + bias
    return (s.softmax(-1) @ v.float()).transpose(1, 2).reshape(q.shape[0], q.shape[2], -1)


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1e-2), (torch.float16, 2e-3)])
@pytest.mark.parametrize("B,H,N", [(1, 1, 64), (2, 3, 197), (1, 2, 100), (2, 12, 333), (1, 4, 1), (1, 2, 65),
                                   (2, 5, 1568), (1, 2, 129)])
def test_prop_attention_against_fp32_reference(B, H, N, dtype, tol):
    """tome_prop_attention (ToMeAttention.forward, tome/patch/videomae.py:55-66 / timesformer.py:66-78 /
    vivit.py:95-113) against an fp32 softmax(q k^T * scale + log(size)) v: plain, with the size bias on every key,
    and in the TimeSformer form (class key / class query unbiased); q, k, v are strided views of one qkv buffer.
    Tolerance: absolute, values are convex combinations of v ~ N(0,1); P is rounded to the 16-bit format before
    the second product, as every fused attention does."""
    from tome import _abi
    g = torch.Generator(device=DEV).manual_seed(N * 7 + H)
    qkv = torch.randn(B, N, 3, H, 64, device=DEV, generator=g).to(dtype)
    q, k, v = qkv.permute(2, 0, 3, 1, 4)
    for mode in ("none", "bias", "skip"):
        n = N - (1 if mode == "skip" else 0)
        size = log_b = None
        if mode != "none":
            if n <= 0:
                continue
            size = torch.randint(1, 30, (B, n, 1), device=DEV, generator=g).float()
            log_b = size.log()[:, :, 0]
        out = _abi.prop_attention(q, k, v, size, 0.125, bias_skip=(mode == "skip"))
        want = _attn_reference(q, k, v, log_b, 0.125, mode == "skip")
        assert out.shape == (B, N, H * 64) and out.dtype == dtype
        assert float((out.float() - want).abs().max()) <= tol, mode


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1.5e-2), (torch.float16, 2e-3)])
@pytest.mark.parametrize("B,H,P,F", [(2, 3, 196, 8), (1, 2, 36, 4), (1, 12, 64, 8), (2, 1, 100, 3), (1, 2, 5, 2)])
def test_prop_attention_segments_against_fp32_reference(B, H, P, F, dtype, tol):
    """tome_prop_attention_segments: every query against F key segments of P keys with a softmax per segment
    (the per-frame stage of ToMeTrajectoryAttention.forward, tome/patch/motionformer.py:98-121) == the reference's own
    expression in fp32: q_dot_k regrouped 'b q (f n) -> b q f n', + the flat per-key bias, softmax over n, times v
    regrouped 'b (f n) d -> b f n d'; q / k / v are views of one qkv buffer behind a class token, as in the patch.
    Tolerance (bf16): the weights go to the second product rounded to the 16-bit format (relative 2^-9, as the
    reference's own bf16 `softmax` output is).  The resident-K/V kernel (<= 224 keys, tome_attn_resident.h) takes its
    reference point from the first 32 keys, so a DOMINANT key in a later block carries a weight 2^x that is rounded
    (the streaming kernel's first 64-key tile gives such a key the exactly representable weight 1): on a row where one
    key holds nearly all the mass the output moves by up to 2^-9 |v| ~ 0.006 more.  Measured on random N(0, 1) q / k / v
    (tools/probes/attn_segment_error.py): mean error identical (4.0e-4), maximum 0.0062-0.0080 streaming, 0.0085-0.0129
    resident."""
    from tome import _abi
    g = torch.Generator(device=DEV).manual_seed(P * 31 + F)
    N = 1 + P * F
    qkv = torch.randn(B, N, 3, H, 64, device=DEV, generator=g).to(dtype)
    q, k, v = qkv.permute(2, 0, 3, 1, 4)
    qs, ks, vs = q[:, :, 1:], k[:, :, 1:], v[:, :, 1:]
    for with_bias in (False, True):
        lb = None
        if with_bias:
            lb = torch.randint(1, 30, (B, P * F), device=DEV, generator=g).float().log()
        y = _abi.prop_attention_segments(qs, ks, vs, F, 0.125, log_bias=lb)
        assert y.shape == (B, N - 1, F, H * 64) and y.dtype == dtype
        logits = (qs.float() @ ks.float().transpose(-1, -2)) * 0.125          # [B, H, S, F*P]
        if lb is not None:
            logits = logits + lb[:, None, None, :]
        w = logits.reshape(B, H, N - 1, F, P).softmax(-1)
        want = torch.einsum("b h q f n, b h f n d -> b q f h d", w, vs.float().reshape(B, H, F, P, 64))
        assert float((y.float() - want.reshape(B, N - 1, F, H * 64)).abs().max()) <= tol, with_bias


def test_prop_attention_properties():
    """Size-independent properties at full size (ViViT: 3137 tokens): (a) the weights of a query sum to one
    (v = ones -> out = ones); (b) a key of size s counts like s copies of that key (what proportional attention
    means); (c) a common factor on all sizes changes nothing; (d) separate q / k / v tensors (ViViT's three
    projections) give the same result as views of one buffer."""
    from tome import _abi
    g = torch.Generator(device=DEV).manual_seed(3)
    B, H, N = 2, 12, 3137
    q = torch.randn(B, H, N, 64, device=DEV, generator=g).bfloat16()
    k = torch.randn(B, H, N, 64, device=DEV, generator=g).bfloat16()
    ones = torch.ones(B, H, N, 64, device=DEV).bfloat16()
    size = torch.randint(1, 9, (B, N, 1), device=DEV, generator=g).float()
    out = _abi.prop_attention(q, k, ones, size, 0.125)
    assert float((out.float() - 1.0).abs().max()) <= 2 ** -7
    # (b) small case: sizes as integer multiplicities
    Bs, Hs, Ns = 1, 2, 40
    qs = torch.randn(Bs, Hs, Ns, 64, device=DEV, generator=g).bfloat16()
    ks = torch.randn(Bs, Hs, Ns, 64, device=DEV, generator=g).bfloat16()
    vs = torch.randn(Bs, Hs, Ns, 64, device=DEV, generator=g).bfloat16()
    mult = torch.randint(1, 4, (Ns,), device=DEV, generator=g)
    rep = torch.repeat_interleave(torch.arange(Ns, device=DEV), mult)
    a = _abi.prop_attention(qs, ks, vs, mult.float().reshape(1, Ns, 1), 0.125)
    qpad = torch.zeros(Bs, Hs, rep.numel(), 64, device=DEV, dtype=torch.bfloat16)
    qpad[:, :, :Ns] = qs
    b_ = _abi.prop_attention(qpad, ks[:, :, rep].contiguous(), vs[:, :, rep].contiguous(), None, 0.125)[:, :Ns]
    assert float((a.float() - b_.float()).abs().max()) <= 1e-2
    # (c) scale invariance in the sizes
    c1 = _abi.prop_attention(qs, ks, vs, mult.float().reshape(1, Ns, 1), 0.125)
    c2 = _abi.prop_attention(qs, ks, vs, (mult.float() * 8.0).reshape(1, Ns, 1), 0.125)
    assert float((c1.float() - c2.float()).abs().max()) <= 1e-2
    # (d) views of a packed buffer vs separate tensors
    qkv = torch.stack((qs, ks, vs)).permute(1, 3, 0, 2, 4).contiguous()  # [B, N, 3, H, 64]
    qv, kv, vv = qkv.permute(2, 0, 3, 1, 4)
    assert torch.equal(_abi.prop_attention(qv, kv, vv, None, 0.125), _abi.prop_attention(qs, ks, vs, None, 0.125))


def test_prop_attention_refuses_what_it_cannot_do():
    from tome import _abi
    from tome._abi import TomeHipError
    q = torch.randn(1, 2, 16, 64, device=DEV).bfloat16()
    with pytest.raises(TomeHipError):
        _abi.prop_attention(q.float(), q.float(), q.float(), None, 0.125)  # 16-bit only
    with pytest.raises(TomeHipError):
        _abi.prop_attention(q[..., :32], q[..., :32], q[..., :32], None, 0.125)  # head dim 64 only
    with pytest.raises(TomeHipError):
        _abi.prop_attention(q, q[:, :, :8], q, None, 0.125)  # shapes differ
    with pytest.raises(TomeHipError):
        _abi.prop_attention(q, q, q, torch.ones(1, 15, 1, device=DEV), 0.125)  # size length
    with pytest.raises(TomeHipError):
        _abi.prop_attention(q.cpu(), q.cpu(), q.cpu(), None, 0.125)  # no CPU path
    assert not _abi.prop_attention_ok(q.float()) and _abi.prop_attention_ok(q)


def _attn_fuzz_cases(count, seed):
    import random
    rng = random.Random(seed)
    cases = []
    for _ in range(count):
        B, H = rng.randint(1, 3), rng.randint(1, 5)
        N = rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257]) if rng.random() < 0.6 else rng.randint(1, 700)
        layout = rng.choice(["qkv", "separate", "padded"])
        mode = rng.choice(["none", "bias", "skip"])
        dtype = rng.choice(["bf16", "f16"])
        cases.append((B, H, N, layout, mode, dtype))
    return cases


@pytest.mark.parametrize("case", _attn_fuzz_cases(60, 20261004), ids=lambda c: "B%dH%dN%d-%s-%s-%s" % c)
def test_prop_attention_fuzz(case):
    """Random shapes around the tile edges (32-query waves, 64-key tiles, 128/256-query workgroups), three memory
    layouts of the heads (slices of one qkv buffer, three separate projections, rows with padding between the
    heads) and the three bias forms, against the fp32 reference."""
    from tome import _abi
    B, H, N, layout, mode, dt = case
    dtype = torch.bfloat16 if dt == "bf16" else torch.float16
    g = torch.Generator(device=DEV).manual_seed(B * 1000003 + H * 1009 + N)
    if layout == "qkv":
        qkv = torch.randn(B, N, 3, H, 64, device=DEV, generator=g).to(dtype)
        q, k, v = qkv.permute(2, 0, 3, 1, 4)
    elif layout == "separate":
        q, k, v = (torch.randn(B, N, H, 64, device=DEV, generator=g).to(dtype).permute(0, 2, 1, 3) for _ in range(3))
    else:  # every token row carries 16 unused elements behind the heads
        bufs = [torch.randn(B, N, H * 64 + 16, device=DEV, generator=g).to(dtype) for _ in range(3)]
        q, k, v = (t[:, :, :H * 64].view(B, N, H, 64).permute(0, 2, 1, 3) for t in bufs)
    n = N - (1 if mode == "skip" else 0)
    size = log_b = None
    if mode != "none" and n > 0:
        size = torch.randint(1, 50, (B, n, 1), device=DEV, generator=g).float()
        log_b = size.log()[:, :, 0]
    elif mode == "skip":
        mode = "none"
    out = _abi.prop_attention(q, k, v, size, 0.125, bias_skip=(mode == "skip"))
    want = _attn_reference(q, k, v, log_b, 0.125, mode == "skip")
    tol = 1e-2 if dtype == torch.bfloat16 else 2e-3
    assert float((out.float() - want).abs().max()) <= tol


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 2e-2), (torch.float16, 3e-3)])
@pytest.mark.parametrize("B,S,F,H", [(2, 37, 8, 12), (1, 5, 4, 2), (3, 196, 8, 16), (1, 1, 1, 1), (2, 64, 3, 5)])
def test_trajectory_mix_against_fp32_reference(B, S, F, H, dtype, tol):
    """tome_trajectory_mix (temporal stage of ToMeTrajectoryAttention.forward, tome/patch/motionformer.py:122-139)
    against the fp32 einsum / softmax / einsum of the reference, with k2 and val as strided halves of one
    proj_kv buffer (as in the model) and as separate tensors."""
    from tome import _abi
    g = torch.Generator(device=DEV).manual_seed(B * 100 + S + F + H)
    C = H * 64
    q2 = torch.randn(B, S, C, device=DEV, generator=g).to(dtype)
    kv = torch.randn(B, S, F, 2 * C, device=DEV, generator=g).to(dtype)
    y = torch.randn(B, S, F, C, device=DEV, generator=g).to(dtype)
    scale = 0.125
    for k2, val in ((kv[..., :C], kv[..., C:]), (kv[..., :C], y)):
        out, attn = _abi.trajectory_mix(q2, k2, val, H, scale)
        qh = q2.float().view(B, S, H, 64).permute(0, 2, 1, 3) * scale
        kh = k2.float().reshape(B, S, F, H, 64).permute(0, 3, 1, 2, 4)
        vh = val.float().reshape(B, S, F, H, 64).permute(0, 3, 1, 2, 4)
        want_attn = torch.einsum("b h s d, b h s f d -> b h s f", qh, kh).softmax(dim=-1)
        want = torch.einsum("b h s f, b h s f d -> b h s d", want_attn, vh).permute(0, 2, 1, 3).reshape(B, S, C)
        assert out.shape == (B, S, C) and out.dtype == dtype and attn.shape == (B, H, S, F)
        assert float((attn - want_attn).abs().max()) <= 1e-4
        assert float((out.float() - want).abs().max()) <= tol
        # written straight behind a class row of a [B, 1+S, C] buffer (the patched Motionformer attention: no cat),
        # without the attention map: the same bits, the neighbouring row untouched
        joined = torch.full((B, 1 + S, C), 7.0, device=DEV, dtype=dtype)
        out2, none = _abi.trajectory_mix(q2, k2, val, H, scale, want_attn=False, out=joined[:, 1:])
        assert none is None and out2.data_ptr() == joined[:, 1:].data_ptr()
        assert torch.equal(joined[:, 1:], out) and bool((joined[:, 0] == 7.0).all())
    assert not _abi.trajectory_mix_ok(q2.float(), kv[..., :C].float(), y.float(), H)
    with pytest.raises(_abi.TomeHipError):
        _abi.trajectory_mix(q2, kv[..., :C], y, H, scale, out=torch.empty(B, S, C + 8, device=DEV, dtype=dtype)[..., :C])


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 2.0 ** -8), (torch.float16, 2.0 ** -11)])
@pytest.mark.parametrize("B,H,N", [(37, 12, 8), (3, 2, 5), (1, 1, 1), (700, 16, 8), (4097, 12, 8), (9, 5, 3)])
def test_short_attention_against_fp32_reference(B, H, N, dtype, tol):
    """tome_short_attention (TimeSformer's temporal attention, tome/patch/timesformer.py:25-27: the host model's
    softmax(q k^T * scale) v over the T <= 8 copies of a spatial token) against the fp32 expression, q / k / v read in
    place from one qkv projection; the framework's fused attention on the same views gives the second opinion."""
    from tome import _abi
    g = torch.Generator(device=DEV).manual_seed(B * 31 + H * 7 + N)
    qkv = torch.randn(B, N, 3, H, 64, device=DEV, generator=g).to(dtype)
    qkv[:, :, 0] *= 3.0  # (peaky softmax rows as well as flat ones)
    q, k, v = qkv.permute(2, 0, 3, 1, 4)
    assert _abi.short_attention_ok(q, k, v)
    out = _abi.short_attention(q, k, v, 0.125)
    assert out.shape == (B, N, H * 64) and out.dtype == dtype and out.is_contiguous()
    att = (q.float() @ k.float().transpose(-2, -1) * 0.125).softmax(dim=-1)
    want = (att @ v.float()).transpose(1, 2).reshape(B, N, H * 64)
    # the result is rounded once to the 16-bit format (tol = one ulp, relative) on top of fp32 arithmetic
    excess = (out.float() - want).abs() - (tol * want.abs() + 1e-3)
    assert float(excess.max()) <= 0.0, float(excess.max())
    sdpa = torch.nn.functional.scaled_dot_product_attention(q, k, v, scale=0.125).transpose(1, 2).reshape(B, N, H * 64)
    excess = (out.float() - sdpa.float()).abs() - (3 * tol * want.abs() + 4e-3)  # (the framework rounds P to 16 bits)
    assert float(excess.max()) <= 0.0, float(excess.max())


def test_short_attention_refuses_what_it_cannot_do():
    from tome import _abi
    qkv = torch.randn(4, 9, 3, 2, 64, device=DEV).bfloat16()
    q, k, v = qkv.permute(2, 0, 3, 1, 4)
    assert not _abi.short_attention_ok(q, k, v)  # nine tokens
    with pytest.raises(_abi.TomeHipError):
        _abi.short_attention(q, k, v, 0.125)
    heads_first = torch.randn(4, 2, 8, 64, device=DEV).bfloat16()  # [B, H, N, 64] contiguous: head stride N * 64
    assert not _abi.short_attention_ok(heads_first, heads_first, heads_first)
    ok = torch.randn(4, 8, 3, 2, 64, device=DEV).bfloat16().permute(2, 0, 3, 1, 4)
    assert not _abi.short_attention_ok(ok[0].float(), ok[1].float(), ok[2].float())
    assert not _abi.short_attention_ok(ok[0].cpu(), ok[1].cpu(), ok[2].cpu())


@pytest.mark.parametrize("growth", [0.05, 1.0, 40.0])
@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 2e-2), (torch.float16, 4e-3)])
def test_prop_attention_running_maximum_moves(growth, dtype, tol):
    """Keys ordered so that every query's logits GROW along the sequence (by `growth` per key on average): the
    online softmax has to move its reference point tile after tile -- slowly (deferred rescale never triggers),
    steadily (triggers regularly) and violently (an un-rescaled exponent would overflow).  Plain attention and
    the size-biased form, against the fp32 reference."""
    from tome import _abi
    g = torch.Generator(device=DEV).manual_seed(17)
    B, H, N = 2, 3, 700
    direction = torch.randn(B, H, 1, 64, device=DEV, generator=g)
    direction = direction / direction.norm(dim=-1, keepdim=True)
    ramp = torch.arange(N, device=DEV).float().view(1, 1, N, 1) * growth
    q = (8.0 * direction + 0.1 * torch.randn(B, H, N, 64, device=DEV, generator=g)).to(dtype)
    k = (ramp * direction + 0.1 * torch.randn(B, H, N, 64, device=DEV, generator=g)).to(dtype)
    v = torch.randn(B, H, N, 64, device=DEV, generator=g).to(dtype)
    size = torch.randint(1, 9, (B, N, 1), device=DEV, generator=g).float()
    for sz in (None, size):
        out = _abi.prop_attention(q, k, v, sz, 0.125)
        want = _attn_reference(q, k, v, None if sz is None else sz.log()[:, :, 0], 0.125, False)
        assert torch.isfinite(out.float()).all()
        assert float((out.float() - want).abs().max()) <= tol


@pytest.mark.parametrize("waves", ["4", "8"])
@pytest.mark.parametrize("case", _attn_fuzz_cases(24, 20261005), ids=lambda c: "B%dH%dN%d-%s-%s-%s" % c)
def test_prop_attention_fuzz_each_workgroup_shape(case, waves, monkeypatch):
    """The fuzz shapes again with the workgroup shape forced (TOME_ATTN_WAVES is read per call): 8-wave workgroups are
    what the large launches of the benchmark run -- waves 4-7 half a barrier interval behind waves 0-3 -- and small
    inputs only reach them this way."""
    monkeypatch.setenv("TOME_ATTN_WAVES", waves)
    test_prop_attention_fuzz(case)


@pytest.mark.parametrize("waves", ["4", "8"])
@pytest.mark.parametrize("N", [65, 96, 97, 128, 160, 161, 449, 480, 481, 1568])
def test_prop_attention_tile_edges_each_workgroup_shape(N, waves, monkeypatch):
    """Key counts around the 64-key tile with at most / more than 32 keys in the last tile (the half-tile shortcut),
    one to many tiles, for both workgroup shapes, all three bias forms and the segmented form."""
    from tome import _abi
    monkeypatch.setenv("TOME_ATTN_WAVES", waves)
    g = torch.Generator(device=DEV).manual_seed(N)
    B, H = 2, 3
    for dtype, tol in ((torch.bfloat16, 1e-2), (torch.float16, 2e-3)):
        qkv = torch.randn(B, N, 3, H, 64, device=DEV, generator=g).to(dtype)
        q, k, v = qkv.permute(2, 0, 3, 1, 4)
        for mode in ("none", "bias", "skip"):
            n = N - (1 if mode == "skip" else 0)
            size = log_b = None
            if mode != "none":
                size = torch.randint(1, 30, (B, n, 1), device=DEV, generator=g).float()
                log_b = size.log()[:, :, 0]
            out = _abi.prop_attention(q, k, v, size, 0.125, bias_skip=(mode == "skip"))
            want = _attn_reference(q, k, v, log_b, 0.125, mode == "skip")
            assert float((out.float() - want).abs().max()) <= tol, (mode, dtype)
    if N <= 161:  # segments of N keys each
        F = 3
        qkv = torch.randn(B, N * F, 3, H, 64, device=DEV, generator=g).bfloat16()
        q, k, v = qkv.permute(2, 0, 3, 1, 4)
        lb = torch.randint(1, 30, (B, N * F), device=DEV, generator=g).float().log()
        y = _abi.prop_attention_segments(q, k, v, F, 0.125, log_bias=lb)
        logits = (q.float() @ k.float().transpose(-1, -2)) * 0.125 + lb[:, None, None, :]
        w = logits.reshape(B, H, N * F, F, N).softmax(-1)
        want = torch.einsum("b h q f n, b h f n d -> b q f h d", w, v.float().reshape(B, H, F, N, 64))
        assert float((y.float() - want.reshape(B, N * F, F, H * 64)).abs().max()) <= 1e-2


@pytest.mark.parametrize("waves", ["4", "8"])
@pytest.mark.parametrize("dtype,tol,jump", [(torch.bfloat16, 2e-2, 70.0), (torch.float16, 4e-3, 18.0),
                                            (torch.float16, 4e-3, 70.0)])
@pytest.mark.parametrize("N", [200, 224, 700])
def test_prop_attention_rerun_path(N, dtype, tol, jump, waves, monkeypatch):
    """The first pass keeps the reference point of tile 0 and is void when a row sum leaves the 16-bit format's
    range; the workgroup then repeats the block on the general path.  Forced here: the logits of every query sit
    near 0 in the first tile and `jump` (log2 units: > 60 for bf16, > 15 for fp16) higher from some later tile on --
    the un-rescaled weights overflow, only the rerun gives finite, correct rows.  Plain, size-biased, TimeSformer's
    bias_skip form and key segments; last tiles with <= 32 and > 32 keys; both workgroup shapes."""
    from tome import _abi
    monkeypatch.setenv("TOME_ATTN_WAVES", waves)
    g = torch.Generator(device=DEV).manual_seed(N + int(jump))
    B, H = 2, 3
    direction = torch.randn(B, H, 1, 64, device=DEV, generator=g)
    direction = direction / direction.norm(dim=-1, keepdim=True)
    # q . k * scale * log2(e) = jump for the late keys: q = 8 d, k = a d -> 8 a / 8 * 1.4427 = jump
    amp = jump / 1.4426950408889634
    step = torch.zeros(1, 1, N, 1, device=DEV)
    step[:, :, 100:] = amp  # from the second tile on
    q = (8.0 * direction + 0.05 * torch.randn(B, H, N, 64, device=DEV, generator=g)).to(dtype)
    k = (step * direction + 0.05 * torch.randn(B, H, N, 64, device=DEV, generator=g)).to(dtype)
    v = torch.randn(B, H, N, 64, device=DEV, generator=g).to(dtype)
    size = torch.randint(1, 9, (B, N, 1), device=DEV, generator=g).float()
    for mode in ("none", "bias", "skip"):
        sz = None if mode == "none" else (size if mode == "bias" else size[:, 1:])
        out = _abi.prop_attention(q, k, v, sz, 0.125, bias_skip=(mode == "skip"))
        want = _attn_reference(q, k, v, None if sz is None else sz.log()[:, :, 0], 0.125, mode == "skip")
        assert torch.isfinite(out.float()).all(), mode
        assert float((out.float() - want).abs().max()) <= tol, mode
    # the naive single-reference computation does overflow on these inputs (the case is not vacuous)
    s2 = (q.float() @ k.float().transpose(-1, -2)) * (0.125 * 1.4426950408889634)
    first_max = s2[..., :64].amax(-1, keepdim=True)
    lim = 60.0 if dtype == torch.bfloat16 else 15.0
    assert float((s2 - first_max).amax()) > lim
    # key segments (Motionformer): the jump inside every segment
    F = 2
    P = N // F
    stepf = torch.zeros(1, 1, F, P, 1, device=DEV)
    stepf[:, :, :, 70:] = amp
    ks = (stepf.reshape(1, 1, F * P, 1) * direction + 0.05 * torch.randn(B, H, F * P, 64, device=DEV, generator=g)).to(dtype)
    qs, vs = q[:, :, :F * P].contiguous(), v[:, :, :F * P].contiguous()
    lb = size[:, :F * P, 0].log().contiguous()
    y = _abi.prop_attention_segments(qs, ks, vs, F, 0.125, log_bias=lb)
    logits = (qs.float() @ ks.float().transpose(-1, -2)) * 0.125 + lb[:, None, None, :]
    w = logits.reshape(B, H, F * P, F, P).softmax(-1)
    wantf = torch.einsum("b h q f n, b h f n d -> b q f h d", w, vs.float().reshape(B, H, F, P, 64))
    assert torch.isfinite(y.float()).all()
    assert float((y.float() - wantf.reshape(B, F * P, F, H * 64)).abs().max()) <= tol


def test_prop_attention_large_launch_takes_eight_wave_workgroups():
    """A launch of the benchmark's size class (B*H*blocks >= 1024: the dispatcher picks 8-wave workgroups by itself)
    against the fp32 reference, per batch slice to bound the reference's memory."""
    from tome import _abi
    g = torch.Generator(device=DEV).manual_seed(5)
    B, H, N = 16, 12, 1568
    qkv = torch.randn(B, N, 3, H, 64, device=DEV, generator=g).bfloat16()
    q, k, v = qkv.permute(2, 0, 3, 1, 4)
    size = torch.randint(1, 9, (B, N, 1), device=DEV, generator=g).float()
    for sz in (None, size):
        out = _abi.prop_attention(q, k, v, sz, 0.125)
        for b in range(0, B, 4):
            want = _attn_reference(q[b:b + 4], k[b:b + 4], v[b:b + 4], None if sz is None else sz[b:b + 4].log()[:, :, 0],
                                   0.125, False)
            assert float((out[b:b + 4].float() - want).abs().max()) <= 1e-2


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gelu_erf_is_the_frameworks_gelu(dtype):
    """tome_gelu_erf (the MLP's activation inside the patched block) == torch's nn.GELU() bit for bit: same fp32
    expression, one rounding; every finite 16-bit value is tried, plus a large tensor and the in-place form."""
    from tome import _abi
    bits = torch.arange(-32768, 32768, dtype=torch.int32).to(torch.int16).to(DEV)
    x = bits.view(dtype)
    x = x[torch.isfinite(x.float())]
    x = x[: x.numel() // 8 * 8].contiguous()
    want = torch.nn.functional.gelu(x)
    assert torch.equal(_abi.gelu_erf(x), want)
    g = torch.Generator(device=DEV).manual_seed(1)
    big = (3.0 * torch.randn(64, 197, 3072, device=DEV, generator=g)).to(dtype)
    want = torch.nn.functional.gelu(big)
    assert torch.equal(_abi.gelu_erf(big), want)
    assert torch.equal(_abi.gelu_erf(big.clone(), inplace=True), want)
    with pytest.raises(_abi.TomeHipError):
        _abi.gelu_erf(big.float())


def test_kernels_are_deterministic():
    """No atomics, no data-dependent reduction order anywhere on the path: the same inputs give the same bits,
    call after call (matching, fused merge + LayerNorm, add + LayerNorm, attention with and without bias)."""
    from tome import _abi
    tm = _tome()
    g = torch.Generator(device=DEV).manual_seed(99)
    n, T, C = 6, 1568, 768
    metric = torch.randn(n, T, 64, device=DEV, generator=g)
    x = torch.randn(n, T, C, device=DEV, generator=g).bfloat16()
    a = torch.randn(n, T, C, device=DEV, generator=g).bfloat16()
    w = torch.randn(C, device=DEV, generator=g).bfloat16()
    qkv = torch.randn(2, 700, 3, 12, 64, device=DEV, generator=g).bfloat16()
    q, k, v = qkv.permute(2, 0, 3, 1, 4)
    size = torch.randint(1, 5, (2, 700, 1), device=DEV, generator=g).float()

    def run():
        merge, _ = tm.bipartite_soft_matching(metric, 16)
        p = merge.plan
        out = [p.src_idx, p.dst_idx, p.unm_idx]
        out += list(_abi.merge_wavg_ln(p, x, None, w, w, 1e-6, addend=a))
        out += list(_abi.add_layernorm(x, a, w, w, 1e-6))
        out += [_abi.prop_attention(q, k, v, None, 0.125), _abi.prop_attention(q, k, v, size, 0.125)]
        return out

    first = run()
    for _ in range(3):
        for t0, t1 in zip(first, run()):
            assert torch.equal(t0, t1)


def test_merge_wavg_ln_many_groups():
    """More merge groups than one grid dimension holds (n > 65535: the (group) index of k_merge_rows_fast spills into
    grid.z): x' == tome_merge_wavg bit for bit, y within one epsilon, and the regrouped form with as many class rows."""
    from tome import _abi
    tm = _tome()
    n, T, C, r = 70001, 6, 16, 2
    metric = dev(synth.normal_like((n, T, 8), 901))
    x = dev(synth.normal_like((n, T, C), 902), torch.bfloat16)
    a = dev(0.5 * synth.normal_like((n, T, C), 903), torch.bfloat16)
    w = dev(1.0 + 0.1 * synth.normal_like((C,), 904), torch.bfloat16)
    b = dev(0.1 * synth.normal_like((C,), 905), torch.bfloat16)
    merge, _ = tm.bipartite_soft_matching(metric, r)
    want_x, want_s = tm.merge_wavg(merge, x + a)
    got_x, got_y, got_s = _abi.merge_wavg_ln(merge.plan, x, None, w, b, 1e-6, addend=a)
    assert torch.equal(got_x, want_x) and torch.equal(got_s, want_s)
    ref = torch.nn.functional.layer_norm(want_x.float(), (C,), w.float(), b.float(), 1e-6)
    assert float(((got_y.float() - ref).abs() / ref.abs().clamp(min=1.0)).max()) <= 2 ** -7
    # interleaved layout: B clips x F frames = n groups, one class row per clip
    B, F = n // 7, 7
    xf = dev(synth.normal_like((B, 1 + T * F, C), 906), torch.bfloat16)
    plan = tm.bipartite_soft_matching(metric[:B * F], r)[0].plan
    wx, ws = _abi.merge_wavg_regrouped(plan, xf, None, F, has_cls=True)
    gx, gy, gs = _abi.merge_wavg_regrouped(plan, xf, None, F, has_cls=True, ln=(w, b, 1e-6))
    assert torch.equal(gx, wx) and torch.equal(gs, ws)
    ref = torch.nn.functional.layer_norm(wx.float(), (C,), w.float(), b.float(), 1e-6)
    assert float(((gy.float() - ref).abs() / ref.abs().clamp(min=1.0)).max()) <= 2 ** -7


def test_merge_wavg_ln_fused_random_shapes():
    """Seeded sweep over the fused kernel's shape space (1-4 rows per wave, 3 and 6 chunks per lane, groups of 2 tokens,
    every even token merged away, class token on / off, sizes on / off, residual on / off, folded bias on / off, both
    16-bit formats): x' and the sizes equal the unfused composition bit for bit, y is the LayerNorm of x' within one
    epsilon of the format."""
    from tome import _abi
    tm = _tome()
    rng = np.random.default_rng(20261004)
    widths = [8, 16, 64, 96, 128, 256, 384, 512, 768, 1024]
    for case in range(48):
        n = int(rng.integers(1, 5))
        T = int(rng.choice([2, 3, 5, 8, 17, 50, 197, 300]))
        C = int(rng.choice(widths))
        cls = bool(rng.integers(0, 2)) and T >= 3
        rmax = (T - (1 if cls else 0)) // 2
        if rmax < 1:
            continue
        r = int(rng.choice([1, rmax, int(rng.integers(1, rmax + 1))]))
        dtype = torch.bfloat16 if rng.integers(0, 2) else torch.float16
        tol = 2 ** -7 if dtype == torch.bfloat16 else 2 ** -10
        seed = 7000 + 11 * case
        metric = dev(synth.normal_like((n, T, 16), seed))
        x = dev(synth.normal_like((n, T, C), seed + 1), dtype)
        a = dev(0.5 * synth.normal_like((n, T, C), seed + 2), dtype) if rng.integers(0, 2) else None
        size = dev(synth.small_ints((n, T, 1), seed + 3, 1, 5), dtype) if rng.integers(0, 2) else None
        ob = dev(0.3 * synth.normal_like((C,), seed + 4), dtype) if rng.integers(0, 2) else None
        w = dev(1.0 + 0.1 * synth.normal_like((C,), seed + 5), dtype)
        b = dev(0.1 * synth.normal_like((C,), seed + 6), dtype)
        merge, _ = tm.bipartite_soft_matching(metric, r, cls)
        want_x, want_s = tm.merge_wavg(merge, x if a is None else x + a, size)
        got_x, got_y, got_s = _abi.merge_wavg_ln(merge.plan, x, size, w, b, 1e-6, addend=a, out_bias=ob)
        tag = f"case {case}: n={n} T={T} C={C} r={r} cls={cls} {dtype} addend={a is not None} size={size is not None} bias={ob is not None}"
        assert torch.equal(got_x, want_x if ob is None else want_x + ob), tag
        assert torch.equal(got_s, want_s), tag
        ref = torch.nn.functional.layer_norm(want_x.float(), (C,), w.float(), b.float(), 1e-6)
        assert float(((got_y.float() - ref).abs() / ref.abs().clamp(min=1.0)).max()) <= tol, tag


def test_regrouped_and_layernorm_random_shapes():
    """Seeded sweep over the interleaved layout (B clips x F frames, class row per clip) and the LayerNorm entry points:
    fused regrouped merge == add, tome_merge_wavg_regrouped, LayerNorm (bit-exact x' and sizes, y within one epsilon);
    tome_add_layernorm with / without addend and with the class row left out agree with each other bit for bit."""
    from tome import _abi
    tm = _tome()
    rng = np.random.default_rng(4102026)
    for case in range(32):
        B, F = int(rng.integers(1, 4)), int(rng.choice([1, 2, 4, 8]))
        P = int(rng.choice([2, 3, 9, 36, 100, 196]))
        C = int(rng.choice([8, 64, 128, 384, 768, 1024]))
        r = int(rng.integers(1, P // 2 + 1))
        dtype = torch.bfloat16 if rng.integers(0, 2) else torch.float16
        tol = 2 ** -7 if dtype == torch.bfloat16 else 2 ** -10
        seed = 9100 + 13 * case
        xf = dev(synth.normal_like((B, 1 + P * F, C), seed), dtype)
        res = dev(0.5 * synth.normal_like((B, 1 + P * F, C), seed + 1), dtype)
        size = dev(synth.small_ints((B * F, P, 1), seed + 2, 1, 4), dtype) if rng.integers(0, 2) else None
        ob = dev(0.3 * synth.normal_like((C,), seed + 3), dtype) if rng.integers(0, 2) else None
        w = dev(1.0 + 0.1 * synth.normal_like((C,), seed + 4), dtype)
        b = dev(0.1 * synth.normal_like((C,), seed + 5), dtype)
        plan = tm.bipartite_soft_matching(dev(synth.normal_like((B * F, P, 16), seed + 6)), r)[0].plan
        tag = f"case {case}: B={B} F={F} P={P} C={C} r={r} {dtype} size={size is not None} bias={ob is not None}"
        wx, ws = _abi.merge_wavg_regrouped(plan, xf + res, size, F, has_cls=True)
        gx, gy, gs = _abi.merge_wavg_regrouped(plan, xf, size, F, has_cls=True, ln=(w, b, 1e-6), addend=res, out_bias=ob)
        assert torch.equal(gx, wx if ob is None else wx + ob) and torch.equal(gs, ws), tag
        ref = torch.nn.functional.layer_norm(wx.float(), (C,), w.float(), b.float(), 1e-6)
        assert float(((gy.float() - ref).abs() / ref.abs().clamp(min=1.0)).max()) <= tol, tag
        # LayerNorm entry points on the merged tokens
        xo, yo = _abi.add_layernorm(wx, gy, w, b, 1e-6)
        assert torch.equal(xo, wx + gy), tag
        assert torch.equal(_abi.add_layernorm(xo, None, w, b, 1e-6)[1], yo), tag
        if wx.shape[1] >= 2:
            xs, ys = _abi.add_layernorm(wx, gy, w, b, 1e-6, skip_first=True)
            assert torch.equal(xs, xo) and torch.equal(ys, yo[:, 1:]), tag
            assert torch.equal(_abi.add_layernorm(xo, None, w, b, 1e-6, skip_first=True)[1], ys), tag
        refy = torch.nn.functional.layer_norm(xo.float(), (C,), w.float(), b.float(), 1e-6)
        assert float(((yo.float() - refy).abs() / refy.abs().clamp(min=1.0)).max()) <= tol, tag
