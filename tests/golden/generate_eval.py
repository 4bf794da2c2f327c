#!/usr/bin/env python3
"""Golden vectors of the reference's EVAL-LOOP functions.  Build container only (reads /root/reference); the GPU box
never runs this.

  * `slowfast/utils/metrics.py` is imported by path as it is (torch + numpy only): `topks_correct` (:9-41) on seeded
    logits, with exact ties inside and across the top-k boundary;
  * `slowfast/utils/meters.py` is imported with name-only stand-ins for what it pulls in besides torch / numpy / pandas
    (fvcore's Timer, wandb, the slowfast.* siblings it only uses for logging and AVA) -- the arithmetic that makes the
    fixture is `TestMeter.update_stats` / `finalize_metrics` themselves (:324-359, :395-436), fed shuffled multi-view
    batches, "sum" and "max" ensembling.

Inputs come from tests/synth.py seeds (regenerated bit-exactly by the tests); stored are only the reference's answers.
    python tests/golden/generate_eval.py        # writes tests/golden/eval.npz + manifest["eval"]
"""
from __future__ import annotations

import importlib.util
import json
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import synth  # noqa: E402,F401
from eval_cases import METER_CASES, TOPK_CASES, meter_inputs, topk_inputs  # noqa: E402  (tests/eval_cases.py)

REF = os.environ.get("TOME_REFERENCE", "/root/reference")


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def install_stand_ins():
    """Names only -- nothing here computes anything the fixture stores."""
    def pkg(name):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
        return m

    for name in ("fvcore", "fvcore.common", "slowfast", "slowfast.utils", "slowfast.datasets"):
        pkg(name)
    timer = types.ModuleType("fvcore.common.timer")

    class Timer:
        def reset(self): pass
        def pause(self): pass
        def seconds(self): return 0.0
    timer.Timer = Timer
    sys.modules["fvcore.common.timer"] = timer
    sys.modules["wandb"] = types.ModuleType("wandb")
    du = types.ModuleType("slowfast.utils.distributed")
    du.is_master_proc = lambda *a, **k: True
    sys.modules["slowfast.utils.distributed"] = du
    for name in ("slowfast.datasets.ava_helper", "slowfast.utils.misc"):
        sys.modules[name] = types.ModuleType(name)
    log = types.ModuleType("slowfast.utils.logging")
    log.get_logger = lambda name: SimpleNamespace(warning=lambda *a, **k: None, info=lambda *a, **k: None)
    log.log_json_stats = lambda *a, **k: None
    sys.modules["slowfast.utils.logging"] = log
    dsu = types.ModuleType("slowfast.datasets.dataset_utils")
    dsu.load_lengths = None
    sys.modules["slowfast.datasets.dataset_utils"] = dsu
    ava = types.ModuleType("slowfast.utils.ava_eval_helper")
    ava.evaluate_ava = ava.read_csv = ava.read_exclusions = ava.read_labelmap = None
    sys.modules["slowfast.utils.ava_eval_helper"] = ava
    metrics = _load("slowfast.utils.metrics", "slowfast/utils/metrics.py")   # the real thing
    sys.modules["slowfast.utils"].metrics = metrics
    meters = _load("slowfast.utils.meters", "slowfast/utils/meters.py")      # the real thing
    return metrics, meters


def main():
    metrics, meters = install_stand_ins()
    arrays = {}
    for case in TOPK_CASES:
        logits, labels = topk_inputs(case)
        ks = (1, min(5, case["classes"]))
        got = metrics.topks_correct(logits, labels, ks)
        arrays[case["name"]] = np.array([float(g) for g in got], dtype=np.float64)
        case["ks"] = list(ks)
    cfg = SimpleNamespace(TEST=SimpleNamespace(CLIP_LENGTH_HISTOGRAM=False))
    for case in METER_CASES:
        preds, labels, clip_ids, batches = meter_inputs(case)
        meter = meters.TestMeter(case["videos"], cfg, case["clips"], case["classes"], len(batches),
                                 ensemble_method=case["method"])
        for idx in batches:
            meter.update_stats(preds[idx], labels[idx], clip_ids[idx])
        meter.finalize_metrics(ks=(1, 5))
        arrays[case["name"] + "_video_preds"] = meter.video_preds.numpy().copy()
        arrays[case["name"] + "_video_labels"] = meter.video_labels.numpy().copy()
        arrays[case["name"] + "_clip_count"] = meter.clip_count.numpy().copy()
        case["top1_acc"], case["top5_acc"] = meter.stats["top1_acc"], meter.stats["top5_acc"]  # the reference's strings
        counts = metrics.topks_correct(meter.video_preds, meter.video_labels, (1, 5))
        case["topk_counts"] = [float(c) for c in counts]
    np.savez_compressed(os.path.join(HERE, "eval.npz"), **arrays)
    man_path = os.path.join(HERE, "manifest.json")
    manifest = json.load(open(man_path))
    manifest["eval"] = {"topk": TOPK_CASES, "meter": METER_CASES,
                        "what": "answers of the reference's slowfast/utils/metrics.py:9-41 (topks_correct) and "
                                "slowfast/utils/meters.py:324-359,395-436 (TestMeter) for tests/synth.py inputs"}
    with open(man_path, "w") as f:
        json.dump(manifest, f, indent=1)
    print({c["name"]: arrays[c["name"]].tolist() for c in TOPK_CASES})
    print({c["name"]: (c["top1_acc"], c["top5_acc"], c["topk_counts"]) for c in METER_CASES})


if __name__ == "__main__":
    main()
