"""hosts/evalloop.py against the REFERENCE's own eval-loop functions (CPU): `topk_counts` == what
slowfast/utils/metrics.py:9-41 `topks_correct` answered, `ClipEnsembleMeter` == what slowfast/utils/meters.py:324-359,
395-436 `TestMeter` accumulated and reported, for the seeded inputs of tests/golden/generate_eval.py (which imported
both reference files in the build container and stored their answers in tests/golden/eval.npz)."""
import os

import numpy as np
import pytest
import torch

import golden_io as G

from eval_cases import meter_inputs, topk_inputs

EVAL = G.manifest().get("eval", {"topk": [], "meter": []})


def _z():
    return np.load(os.path.join(G.GOLDEN, "eval.npz"))


def test_the_eval_fixtures_exist():
    assert len(EVAL["topk"]) >= 5 and len(EVAL["meter"]) >= 5, "run tests/golden/generate_eval.py"


@pytest.mark.parametrize("case", EVAL["topk"], ids=lambda c: c["name"])
def test_topk_counts_equal_the_references_topks_correct(case):
    from hosts.evalloop import topk_counts
    logits, labels = topk_inputs(case)
    got = topk_counts(logits, labels, tuple(case["ks"]))
    want = _z()[case["name"]]
    assert got.dtype == torch.int64 and got.shape == (len(case["ks"]) + 1,)
    assert [int(v) for v in got[:-1]] == [int(v) for v in want], (got, want)
    assert int(got[-1]) == case["n"]
    # 16-bit logits are counted on their fp32 values, like the reference's `preds` after `.float()`-free topk on the
    # same numbers: the counts of the bf16-rounded logits equal the counts the function gives for their fp32 copies
    lb = logits.bfloat16()
    assert torch.equal(topk_counts(lb, labels, tuple(case["ks"])), topk_counts(lb.float(), labels, tuple(case["ks"])))


@pytest.mark.parametrize("case", EVAL["meter"], ids=lambda c: c["name"])
def test_clip_ensemble_meter_equals_the_references_test_meter(case):
    from hosts.evalloop import ClipEnsembleMeter
    preds, labels, clip_ids, batches = meter_inputs(case)
    z = _z()
    meter = ClipEnsembleMeter(case["videos"], case["clips"], case["classes"], ensemble_method=case["method"])
    for idx in batches:
        meter.update(preds[idx], labels[idx], clip_ids[idx])
    want = z[case["name"] + "_video_preds"]
    if case["method"] == "max":
        assert np.array_equal(meter.video_preds.numpy(), want)          # a maximum has no rounding
    else:
        # the reference adds a video's clips one by one in arrival order; index_add_ adds the same fp32 numbers in an
        # order of its own: within a few ulp of the partial sums
        np.testing.assert_allclose(meter.video_preds.numpy(), want, rtol=0, atol=4e-6 * case["clips"])
    assert np.array_equal(meter.video_labels.numpy(), z[case["name"] + "_video_labels"])
    assert np.array_equal(meter.clip_count.numpy(), z[case["name"] + "_clip_count"])
    stats = meter.finalize(ks=(1, 5))
    assert stats["videos"] == case["videos"] and stats["all_clips_seen"]
    # the reference reports "{:.2f}" strings of 100 * correct / videos (meters.py:421-433)
    assert "{:.2f}".format(stats["top1_acc"]) == case["top1_acc"]
    assert "{:.2f}".format(stats["top5_acc"]) == case["top5_acc"]
    assert round(stats["top1_acc"] * case["videos"] / 100.0) == int(case["topk_counts"][0])
    assert round(stats["top5_acc"] * case["videos"] / 100.0) == int(case["topk_counts"][1])


def test_meter_refuses_an_unknown_ensemble_method_like_the_reference():
    from hosts.evalloop import ClipEnsembleMeter
    with pytest.raises(NotImplementedError, match="not supported"):   # meters.py:353-357
        ClipEnsembleMeter(4, 2, 3, ensemble_method="mean")
