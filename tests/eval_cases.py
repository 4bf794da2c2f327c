"""Inputs of the eval-loop golden vectors (tests/golden/generate_eval.py makes the answers with the reference's own
functions, tests/test_eval_golden.py holds hosts/evalloop.py to them): case lists and the seeded input builders."""
import numpy as np
import torch

import synth


def topk_inputs(case):
    """Seeded logits [N, C] and labels [N]; `ties` quantises the logits so that exact ties occur inside the top 5 and
    across its boundary."""
    n, c, seed = case["n"], case["classes"], case["seed"]
    logits = synth.normal_like((n, c), seed)
    if case.get("ties"):
        logits = np.round(logits * np.float32(case["ties"])) / np.float32(case["ties"])
    labels = (synth.small_ints((n,), seed ^ 0x77, 0, c - 1)).astype(np.int64)
    return torch.from_numpy(logits.astype(np.float32)), torch.from_numpy(labels)


def meter_inputs(case):
    """Multi-view batches: V videos x K clips, clip id = video * K + view, one label per video; the clips arrive in a
    seeded shuffled order, in batches of `batch` (the last one ragged)."""
    v, k, c, seed = case["videos"], case["clips"], case["classes"], case["seed"]
    preds = torch.from_numpy(synth.normal_like((v * k, c), seed))
    if case.get("ties"):
        preds = torch.round(preds * case["ties"]) / case["ties"]
    labels = torch.from_numpy(synth.small_ints((v,), seed ^ 0x99, 0, c - 1).astype(np.int64)).repeat_interleave(k)
    clip_ids = torch.arange(v * k)
    order = np.argsort(synth.uniform01((v * k,), seed ^ 0x1234), kind="stable")
    batches = [torch.from_numpy(order[a:a + case["batch"]].copy()) for a in range(0, v * k, case["batch"])]
    return preds, labels, clip_ids, batches


TOPK_CASES = [
    dict(name="topk_plain", n=64, classes=400, seed=101),
    dict(name="topk_small", n=7, classes=11, seed=102),
    dict(name="topk_ties", n=96, classes=12, seed=103, ties=2.0),
    dict(name="topk_all_equal", n=5, classes=9, seed=104, ties=0.01),   # every logit rounds to 0: all tied
    dict(name="topk_k_equals_classes", n=9, classes=5, seed=105),
]
METER_CASES = [
    dict(name="meter_sum", videos=23, clips=3, classes=11, seed=201, batch=7, method="sum"),
    dict(name="meter_max", videos=23, clips=3, classes=11, seed=201, batch=7, method="max"),
    dict(name="meter_sum_10x3", videos=16, clips=30, classes=24, seed=202, batch=64, method="sum"),
    dict(name="meter_max_ties", videos=12, clips=4, classes=6, seed=203, batch=5, method="max", ties=2.0),
    dict(name="meter_single_view", videos=31, clips=1, classes=13, seed=204, batch=8, method="sum"),
]
