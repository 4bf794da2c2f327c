#!/usr/bin/env python3
"""Golden-vector generator.  Runs ONLY in the build container, where the reference checkout is
mounted read-only at /root/reference; the GPU box never sees the reference and never runs this.

It imports the reference's own ``tome/merge.py`` and ``tome/utils.py`` by file path, feeds them the
deterministic inputs of ``tests/synth.py`` and stores what the reference answered:

  match.npz    index tensors (src_idx, dst_idx, unm_idx) of bipartite_soft_matching, plus a margin
               certificate computed in fp64 (which parts of the answer are independent of fp32
               summation order and of argsort's undefined tie order -- SURVEY 7.1)
  values.npz   merge_wavg / merge(mode) / merge_source / unmerge / drop / hybrid outputs
  parse_r.json r-schedule tables of tome.utils.parse_r
  manifest.json  the case list (shape, seed, flags) the tests iterate over

Only inputs' seeds and the reference's outputs are stored -- no reference source text.
"""
from __future__ import annotations

import importlib.util
import json
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import synth  # noqa: E402

REF = os.environ.get("TOME_REFERENCE", "/root/reference")
TAU = 1e-6  # decision margin (fp64) above which fp32 implementations must agree


def load_ref(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


ref_merge = load_ref("_ref_tome_merge", "tome/merge.py")
ref_utils = load_ref("_ref_tome_utils", "tome/utils.py")


def closure_vars(fn):
    return dict(zip(fn.__code__.co_freevars, (c.cell_contents for c in fn.__closure__)))


def make_metric(kind, n, T, D, seed):
    if kind == "normal":
        return synth.normal_like((n, T, D), seed)
    if kind == "clustered":
        return synth.clustered((n, T, D), seed)
    raise ValueError(kind)


def certificate(metric, r, cls, distill, src, dst, unm, check_dst=True):
    """fp64 re-evaluation of merge.py:51-73 on the same inputs.  Returns which parts of the
    reference's answer have decision margins > TAU, and checks the reference agrees with fp64
    there."""
    m = torch.from_numpy(metric).double()
    m = m / m.norm(dim=-1, keepdim=True)
    a, b = m[:, ::2], m[:, 1::2]
    s = a @ b.transpose(-1, -2)
    if cls:
        s[:, 0, :] = -math.inf
    if distill:
        s[:, :, 0] = -math.inf
    nmax, nidx = s.max(-1)
    order = nmax.argsort(dim=-1, descending=True, stable=True)
    snm = nmax.gather(-1, order)
    gaps = (snm[:, :-1] - snm[:, 1:])  # [n, T1-1], >= 0
    gaps = torch.nan_to_num(gaps, nan=math.inf)  # (-inf) - (-inf)
    T1 = s.shape[1]
    # src order + boundary: gaps 0..r-1 (gap k is between rank k and k+1)
    g_src = gaps[:, :r].min().item() if r > 0 and T1 > 1 else math.inf
    g_unm = gaps[:, r:].min().item() if T1 - r > 1 else math.inf
    # dst choice for the selected rows: top-2 gap
    rows = order[:, :r]
    if s.shape[2] > 1:
        top2 = s.gather(1, rows[..., None].expand(-1, -1, s.shape[2])).topk(2, dim=-1).values
        g_dst = torch.nan_to_num(top2[..., 0] - top2[..., 1], nan=math.inf).min().item()
    else:
        g_dst = math.inf
    cert = {
        "src": bool(g_src > TAU),
        "dst": bool(g_dst > TAU),
        "unm": bool((g_unm > TAU) if not cls else (g_src > TAU)),
        "g_src": g_src, "g_dst": g_dst, "g_unm": g_unm,
    }
    # sanity: where certified, the fp32 reference must equal the fp64 evaluation
    src64 = order[:, :r].numpy()
    if cert["src"]:
        assert np.array_equal(src64, src), "reference fp32 src_idx differs from fp64 despite margin"
        if cert["dst"] and check_dst:
            assert np.array_equal(nidx.gather(-1, order[:, :r]).numpy(), dst)
    if cert["unm"] and cert["src"]:
        u64 = order[:, r:].numpy()
        if cls:
            u64 = np.sort(u64, axis=1)
        assert np.array_equal(u64, unm)
    return cert


def run_match(case, max_tries):
    n, T, D, r, cls, kind = case["n"], case["T"], case["D"], case["r"], case["cls"], case["kind"]
    distill = case.get("distill", False)
    best = None
    for attempt in range(max_tries):
        seed = case["seed0"] + 1000003 * attempt
        metric = make_metric(kind, n, T, D, seed)
        merge, unmerge = ref_merge.bipartite_soft_matching(torch.from_numpy(metric), r, cls, distill)
        if merge is ref_merge.do_nothing:
            return {"seed": seed, "r_eff": 0, "cert": {"src": True, "dst": True, "unm": True}}, None
        cv = closure_vars(merge)
        src = cv["src_idx"][..., 0].numpy()
        dst = cv["dst_idx"][..., 0].numpy()
        unm = cv["unm_idx"][..., 0].numpy()
        cert = certificate(metric, cv["r"], cls, distill, src, dst, unm)
        score = (cert["src"] and cert["dst"], cert["unm"])
        cand = ({"seed": seed, "r_eff": int(cv["r"]), "cert": cert}, (src, dst, unm))
        if best is None or score > best[0]:
            best = (score, cand)
        if all(score):
            break
        if score[0] and not case.get("want_unm", True):
            break
    return best[1]


def match_cases():
    cases = []
    cid = 0

    def add(kind, n, T, D, r, cls, distill=False, want_unm=True):
        nonlocal cid
        cases.append(dict(id=f"m{cid:03d}", kind=kind, n=n, T=T, D=D, r=r, cls=cls, distill=distill,
                          seed0=7919 * (cid + 1), want_unm=want_unm))
        cid += 1

    # tiny / odd / clamp edge cases (merge.py:43-47), with and without class token
    for T in (1, 2, 3, 5, 9, 20):
        for r in (0, 1, 8, 10**6):
            for cls in (False, True):
                add("normal", 3, T, 8, r, cls)
    # TimeSformer / Motionformer frame groups (196 tokens, n = frames*batch), schedule tail sizes
    for T in (196, 164, 36, 18):
        for r in (8, 16, 32):
            add("normal", 8, T, 64, r, False)
    add("clustered", 8, 196, 64, 16, False)
    add("normal", 4, 197, 64, 16, True)      # odd T with class token
    add("normal", 4, 197, 64, 98, True)      # r == max
    add("normal", 2, 197, 64, 16, True, distill=True)
    add("normal", 2, 198, 64, 16, True, distill=True)
    # head_aggregation='concat' metric width (videomae.py:74-75)
    add("normal", 2, 196, 768, 16, False)
    add("normal", 2, 60, 40, 7, False)       # D not a multiple of 64, odd-ish sizes
    add("normal", 2, 61, 2, 7, True)
    # VideoMAE 8-frame and 16-frame sequences (784 / 1568 tokens) and schedule points
    for r in (8, 16, 64):
        add("normal", 2, 784, 64, r, False)
    add("normal", 1, 784, 64, 392, False, want_unm=False)
    add("clustered", 2, 784, 64, 16, False, want_unm=False)
    for r in (8, 16):
        add("normal", 2, 1568, 64, r, False)
    add("normal", 1, 1568, 64, 150, False, want_unm=False)   # experiments.sh "roughly optimal" r
    add("normal", 1, 1392, 64, 16, False)                     # last VideoMAE layer at r=16
    add("normal", 1, 1568, 64, 16, False)                     # single group: unm order certifiable
    add("clustered", 2, 1568, 64, 16, False, want_unm=False)
    # ViViT: 3137 tokens with class token (unm sorted => order fragility gone)
    add("normal", 2, 3137, 64, 64, True)
    add("normal", 1, 3137, 64, 300, True)
    add("clustered", 1, 3137, 64, 64, True)
    add("normal", 1, 1569, 64, 64, True)
    return cases


def gen_match(out_dir):
    arrays, manifest = {}, []
    for case in match_cases():
        big = case["T"] >= 700
        info, idx = run_match(case, max_tries=3000 if big else 60)
        entry = {k: case[k] for k in ("id", "kind", "n", "T", "D", "r", "cls", "distill")}
        entry.update(seed=info["seed"], r_eff=info["r_eff"],
                     cert={k: info["cert"][k] for k in ("src", "dst", "unm")})
        manifest.append(entry)
        if idx is not None:
            src, dst, unm = idx
            arrays[case["id"] + "_src"] = src.astype(np.int16)
            arrays[case["id"] + "_dst"] = dst.astype(np.int16)
            arrays[case["id"] + "_unm"] = unm.astype(np.int16)
        print(entry["id"], entry["kind"], entry["n"], entry["T"], entry["D"], "r", entry["r"], "->",
              entry["r_eff"], "cls" if entry["cls"] else "", entry["cert"], flush=True)
    np.savez_compressed(os.path.join(out_dir, "match.npz"), **arrays)
    return manifest


def value_cases():
    cases = []
    vid = 0

    def add(**kw):
        nonlocal vid
        kw["id"] = f"v{vid:03d}"
        kw.setdefault("seed", 104729 * (vid + 1))
        kw.setdefault("cls", False)
        kw.setdefault("distill", False)
        kw.setdefault("kind", "normal")
        cases.append(kw)
        vid += 1

    for (n, T, D, C, r, cls) in [(2, 20, 8, 8, 3, False), (2, 21, 8, 8, 4, True), (3, 196, 64, 8, 16, False),
                                 (2, 197, 64, 8, 32, True), (1, 784, 64, 4, 16, False),
                                 (2, 9, 8, 1, 4, False), (2, 196, 64, 8, 98, False),
                                 (1, 1568, 64, 2, 16, False)]:
        add(op="wavg", n=n, T=T, D=D, C=C, r=r, cls=cls, sizes="none")
        add(op="wavg", n=n, T=T, D=D, C=C, r=r, cls=cls, sizes="ints")
    add(op="wavg", n=2, T=198, D=64, C=8, r=16, cls=True, distill=True, sizes="ints")
    for mode in ("sum", "mean", "max", "amax", "prod", "amin"):
        add(op="merge", n=2, T=61, D=16, C=8, r=20, mode=mode)
        add(op="merge", n=2, T=196, D=64, C=4, r=60, mode=mode, kind="clustered")
    add(op="source", n=2, T=40, D=16, r=8)            # merge_source twice (two layers)
    add(op="source", n=1, T=41, D=16, r=9, cls=True)
    add(op="unmerge", n=2, T=61, D=16, C=8, r=20)
    add(op="unmerge", n=2, T=196, D=64, C=8, r=16)
    add(op="unmerge", n=2, T=197, D=64, C=8, r=16, cls=True)
    add(op="drop", n=2, T=61, D=16, C=8, r=20)
    add(op="drop", n=2, T=197, D=64, C=8, r=16, cls=True)
    for thr in (-1.0, 0.05, 0.3, 0.5, 2.0):
        add(op="hybrid", n=2, T=196, D=64, C=8, r=40, threshold=thr, sizes="ints")
    add(op="hybrid", n=2, T=197, D=64, C=8, r=40, threshold=0.3, sizes="none", cls=True)
    add(op="hybrid_merge", n=2, T=196, D=64, C=8, r=40, threshold=0.3, mode="mean")
    return cases


def gen_values(out_dir):
    arrays, manifest = {}, []
    for c in value_cases():
        n, T, D, r = c["n"], c["T"], c["D"], c["r"]
        C = c.get("C", 0)
        cls, distill = c["cls"], c["distill"]
        # find a seed whose matching is fully certified, so the values do not depend on tie order
        seed = c["seed"]
        for attempt in range(400):
            metric = make_metric(c["kind"], n, T, D, seed)
            mt = torch.from_numpy(metric)
            if c["op"] in ("hybrid", "hybrid_merge"):
                merge, unmerge = ref_merge.bipartite_soft_matching_hybrid(mt, r, cls, distill, "hybrid",
                                                                          c["threshold"])
            elif c["op"] == "drop":
                merge = ref_merge.bipartite_soft_matching_drop(mt, r, cls, distill)
                unmerge = None
            else:
                merge, unmerge = ref_merge.bipartite_soft_matching(mt, r, cls, distill)
            cv = closure_vars(merge)
            src = cv["src_idx"][..., 0].numpy()
            unm = cv["unm_idx" if "unm_idx" in cv else "und_idx"][..., 0].numpy()
            dst = cv["dst_idx"][..., 0].numpy() if "dst_idx" in cv else np.zeros_like(src)
            cert = certificate(metric, cv["r"], cls, distill, src, dst, unm, check_dst=c["op"] != "drop")
            if c["op"] == "drop":
                ok = cert["src"] and cert["unm"]
            else:
                ok = cert["src"] and cert["dst"] and cert["unm"]
            if ok and c["op"] in ("hybrid", "hybrid_merge"):
                # the threshold decision must be robust too
                nm = cv["node_max"].double()
                ok = bool(((nm - c["threshold"]).abs() > 1e-5).all())
            if ok:
                break
            seed += 1000003
        assert ok, f"could not certify {c}"
        c["seed"] = seed
        c["r_eff"] = int(cv["r"])
        k = c["id"]
        arrays[k + "_src"] = src.astype(np.int16)
        arrays[k + "_dst"] = dst.astype(np.int16)
        arrays[k + "_unm"] = unm.astype(np.int16)
        if C:
            x = synth.normal_like((n, T, C), seed ^ 0xABCDEF)
            xt = torch.from_numpy(x)
        if c["op"] in ("wavg", "hybrid"):
            size = None
            if c["sizes"] == "ints":
                size = torch.from_numpy(synth.small_ints((n, T, 1), seed ^ 0x51235))
            xo, so = ref_merge.merge_wavg(merge, xt, size)
            arrays[k + "_x"] = xo.numpy()
            arrays[k + "_size"] = so.numpy()
        elif c["op"] in ("merge", "hybrid_merge"):
            arrays[k + "_x"] = merge(xt, mode=c["mode"]).numpy()
        elif c["op"] == "source":
            x = torch.zeros(n, T, 4)
            s1 = ref_merge.merge_source(merge, x, None)
            # second layer on the merged sequence
            metric2 = make_metric(c["kind"], n, T - cv["r"], D, seed ^ 0x2222)
            merge2, _ = ref_merge.bipartite_soft_matching(torch.from_numpy(metric2), r, cls, distill)
            cv2 = closure_vars(merge2)
            s2 = ref_merge.merge_source(merge2, torch.zeros(n, T - cv["r"], 4), s1)
            arrays[k + "_s1"] = s1.numpy().astype(np.uint8)
            arrays[k + "_s2"] = s2.numpy().astype(np.uint8)
            arrays[k + "_src2"] = cv2["src_idx"][..., 0].numpy().astype(np.int16)
            arrays[k + "_dst2"] = cv2["dst_idx"][..., 0].numpy().astype(np.int16)
            arrays[k + "_unm2"] = cv2["unm_idx"][..., 0].numpy().astype(np.int16)
        elif c["op"] == "unmerge":
            merged = merge(xt, mode="mean")
            arrays[k + "_x"] = unmerge(merged).numpy()
            arrays[k + "_merged"] = merged.numpy()
        elif c["op"] == "drop":
            arrays[k + "_x"] = merge(xt).numpy()
        manifest.append(c)
        print(c, flush=True)
    np.savez_compressed(os.path.join(out_dir, "values.npz"), **arrays)
    return manifest


def gen_parse_r(out_dir):
    table = []
    args = [(12, 16), (12, 0), (12, 8), (12, (16, 0)), (12, (16, -1)), (12, (16, 1)), (12, (16, 0.5)),
            (12, (16, -0.5)), (12, (18, -1)), (12, (150, 0)), (12, (300, 1)), (12, [8, 8]),
            (12, [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12]), (13, (16, -1)), (24, (8, -1)), (2, (5, 1)),
            (12, [4] * 14)]
    for L, r in args:
        rr = list(r) if isinstance(r, list) else r
        table.append({"num_layers": L, "r": list(r) if isinstance(r, tuple) else rr,
                      "r_type": type(r).__name__, "out": ref_utils.parse_r(L, r)})
    with open(os.path.join(out_dir, "parse_r.json"), "w") as f:
        json.dump(table, f, indent=1)
    return table


def main():
    out_dir = HERE
    which = sys.argv[1:] or ["match", "values", "parse_r"]
    man_path = os.path.join(out_dir, "manifest.json")
    manifest = json.load(open(man_path)) if os.path.exists(man_path) else {}
    manifest["tau"] = TAU
    manifest["torch"] = torch.__version__
    if "match" in which:
        manifest["match"] = gen_match(out_dir)
    if "values" in which:
        manifest["values"] = gen_values(out_dir)
    if "parse_r" in which:
        gen_parse_r(out_dir)
    with open(man_path, "w") as f:
        json.dump(manifest, f, indent=1)


if __name__ == "__main__":
    main()
