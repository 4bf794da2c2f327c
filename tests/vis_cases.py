"""Deterministic inputs of the visualisation fixtures (shared by tests/golden/generate_vis.py and
tests/test_vis_cpu.py): pictures / clips from the splitmix64 streams of synth.py and a random assignment of
the original tokens to merged tokens written as the 0/1 `source` matrix a trace_source forward produces."""
import numpy as np
import torch

import synth

CASES = [
    {"id": "img0", "kind": "image", "shape": (48, 64), "patch": (8,), "groups": 20, "cls": True, "seed": 11},
    {"id": "img1", "kind": "image", "shape": (37, 50), "patch": (12,), "groups": 5, "cls": False, "seed": 12},
    {"id": "sp0", "kind": "spatial", "shape": (4, 32, 48), "patch": (8, 8, 2), "groups": 9, "cls": True, "avg": True,
     "seed": 13},
    {"id": "sp1", "kind": "spatial", "shape": (4, 32, 48), "patch": (8, 16, 1), "groups": 4, "cls": False, "avg": False,
     "seed": 14},
    {"id": "st0", "kind": "spatiotemporal", "shape": (4, 32, 48), "patch": (8, 8, 2), "groups": 14, "cls": True,
     "avg": True, "uncovered": 3, "seed": 15},
    {"id": "st1", "kind": "spatiotemporal", "shape": (6, 24, 24), "patch": (8, 8, 2), "groups": 6, "cls": False,
     "avg": False, "uncovered": 0, "seed": 16},
]


def _source(n, tokens, groups, cls, seed, uncovered=0):
    u = synth.uniform01((n, tokens), seed)
    assign = np.minimum((u * groups).astype(np.int64), groups - 1)
    src = np.zeros((n, groups + (1 if cls else 0), tokens + (1 if cls else 0)), np.float32)
    off = 1 if cls else 0
    if cls:
        src[:, 0, 0] = 1.0
    for b in range(n):
        src[b, assign[b] + off, np.arange(tokens) + off] = 1.0
        if uncovered:  # tokens no merged token accounts for (spatiotemporal: drawn black)
            dead = np.argsort(synth.uniform01((tokens,), seed + 99))[:uncovered]
            src[b, :, dead + off] = 0.0
    return torch.from_numpy(src)


def vis_inputs(c):
    if c["kind"] == "image":
        h, w = c["shape"]
        pixels = (synth.uniform01((h, w, 3), c["seed"]) * 256).astype(np.uint8)
        p = c["patch"][0]
        return pixels, _source(1, (h // p) * (w // p), c["groups"], c["cls"], c["seed"] + 1)
    t, h, w = c["shape"]
    pixels = synth.uniform01((t, 3, h, w), c["seed"])
    ph, pw, pt = h // c["patch"][0], w // c["patch"][1], t // c["patch"][2]
    if c["kind"] == "spatial":
        return pixels, _source(pt, ph * pw, c["groups"], c["cls"], c["seed"] + 1)
    return pixels, _source(1, pt * ph * pw, c["groups"], c["cls"], c["seed"] + 1, c.get("uncovered", 0))
