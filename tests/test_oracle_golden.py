"""Pins the CPU oracle (oracle/tome_oracle.c) against golden vectors produced by the real
reference (tests/golden/generate.py imported /root/reference/tome/merge.py in the build
container).  CPU only."""
import numpy as np
import pytest

import golden_io as G
import oracle


@pytest.mark.parametrize("case", G.match_cases(), ids=lambda c: c["id"])
def test_match_indices(case):
    plan = oracle.match(G.metric_of(case), case["r"], case["cls"], case["distill"])
    if case["r_eff"] == 0:  # merge.py:46-47 do_nothing
        assert plan is None
        return
    assert plan.r == case["r_eff"]
    z = G.arrays("match")
    G.check_indices(case, plan.src_idx, plan.dst_idx, plan.unm_idx, z[case["id"] + "_src"],
                    z[case["id"] + "_dst"], z[case["id"] + "_unm"])
    # structural invariants: src ∪ unm is a permutation of the even-token rows
    T1 = (case["T"] + 1) // 2
    both = np.concatenate([plan.src_idx[..., 0], plan.unm_idx[..., 0]], axis=1)
    assert np.array_equal(np.sort(both, axis=1), np.broadcast_to(np.arange(T1), both.shape))
    assert plan.dst_idx.min() >= 0 and plan.dst_idx.max() < case["T"] // 2


def _plan_for(case):
    plan = oracle.match(G.metric_of(case), case["r"], case["cls"], case["distill"])
    z = G.arrays("values")
    k = case["id"]
    np.testing.assert_array_equal(plan.src_idx[..., 0], z[k + "_src"])
    np.testing.assert_array_equal(plan.unm_idx[..., 0], z[k + "_unm"])
    if case["op"] != "drop":
        np.testing.assert_array_equal(plan.dst_idx[..., 0], z[k + "_dst"])
    return plan, z


@pytest.mark.parametrize("case", G.value_cases("wavg"), ids=lambda c: c["id"])
def test_merge_wavg_values(case):
    plan, z = _plan_for(case)
    xo, so = oracle.merge_wavg(plan, G.x_of(case), G.size_of(case))
    # same fp32 op sequence as the reference => bit-exact
    np.testing.assert_array_equal(so, z[case["id"] + "_size"])
    np.testing.assert_array_equal(xo, z[case["id"] + "_x"])
    T0 = case["T"] if G.size_of(case) is None else G.size_of(case).sum(1)
    np.testing.assert_array_equal(so.sum(1), np.broadcast_to(T0, so.sum(1).shape))


@pytest.mark.parametrize("case", G.value_cases("merge"), ids=lambda c: c["id"] + c["mode"])
def test_merge_modes(case):
    plan, z = _plan_for(case)
    out = oracle.merge(plan, G.x_of(case), case["mode"])
    np.testing.assert_array_equal(out, z[case["id"] + "_x"])


@pytest.mark.parametrize("case", G.value_cases("unmerge"), ids=lambda c: c["id"])
def test_unmerge(case):
    plan, z = _plan_for(case)
    merged = oracle.merge(plan, G.x_of(case), "mean")
    np.testing.assert_array_equal(merged, z[case["id"] + "_merged"])
    np.testing.assert_array_equal(oracle.unmerge(plan, merged), z[case["id"] + "_x"])


@pytest.mark.parametrize("case", G.value_cases("drop"), ids=lambda c: c["id"])
def test_drop(case):
    plan, z = _plan_for(case)
    np.testing.assert_array_equal(oracle.drop(plan, G.x_of(case)), z[case["id"] + "_x"])


@pytest.mark.parametrize("case", G.value_cases("hybrid") + G.value_cases("hybrid_merge"),
                         ids=lambda c: c["id"])
def test_hybrid(case):
    plan, z = _plan_for(case)
    hp = plan.with_threshold(case["threshold"])
    if case["op"] == "hybrid":
        xo, so = oracle.merge_wavg(hp, G.x_of(case), G.size_of(case))
        np.testing.assert_array_equal(so, z[case["id"] + "_size"])
        want = z[case["id"] + "_x"]
        # a destination whose own term was dropped and that has size 0 cannot occur (sources add >=1)
        np.testing.assert_array_equal(xo, want)
    else:
        np.testing.assert_array_equal(oracle.merge(hp, G.x_of(case), case["mode"]), z[case["id"] + "_x"])


@pytest.mark.parametrize("case", G.value_cases("source"), ids=lambda c: c["id"])
def test_merge_source(case):
    """merge.py:372-384: source = merge(eye, mode='max'), chained over two layers."""
    plan, z = _plan_for(case)
    n, T = case["n"], case["T"]
    eye = np.broadcast_to(np.eye(T, dtype=np.float32), (n, T, T))
    s1 = oracle.merge(plan, eye, "amax")
    np.testing.assert_array_equal(s1.astype(np.uint8), z[case["id"] + "_s1"])
    plan2 = oracle.match(G.metric2_of(case, plan.r), case["r"], case["cls"], case["distill"])
    np.testing.assert_array_equal(plan2.src_idx[..., 0], z[case["id"] + "_src2"])
    np.testing.assert_array_equal(plan2.dst_idx[..., 0], z[case["id"] + "_dst2"])
    np.testing.assert_array_equal(plan2.unm_idx[..., 0], z[case["id"] + "_unm2"])
    s2 = oracle.merge(plan2, s1, "amax")
    np.testing.assert_array_equal(s2.astype(np.uint8), z[case["id"] + "_s2"])
    assert np.array_equal(s2.sum(1), np.ones((n, T)))  # every original token is in exactly one group


def test_effective_r_matches_python_floor_division():
    for T in range(0, 12):
        for r in (-3, 0, 1, 5, 100):
            for cls in (0, 1):
                for dist in (0, 1):
                    want = max(0, min(r, (T - cls - dist) // 2))
                    assert oracle.effective_r(T, r, cls, dist) == want
