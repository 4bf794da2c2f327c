#!/usr/bin/env python3
"""Golden vectors of the reference's PATCHED MODELS at reduced width.  Build container only (reads
/root/reference); the GPU box never runs this.

The reference's `tome.patch.{videomae,timesformer,motionformer}` and the slowfast model files they patch
are imported from /root/reference with the missing third-party modules (timm, fvcore, torchvision, ...)
replaced by minimal in-memory stand-ins that provide only names (no arithmetic): the arithmetic that
produces the fixtures is the reference's own.  ViViT cannot be imported (installed transformers lacks
VivitSelfAttention) and has no block-level fixture.

Per model the fixture (models_<name>.npz) holds: the logits of the patched forward, the final token sizes and
per layer the matching's indices; the manifest holds the clip seed, the weight seed (weights are filled by
tests/synth.fill_parameters, keyed by parameter NAME, so no state_dict is shipped), the parameter names and the
fp64 decision margin of every layer.  Seeds are searched until every layer is certified with tau = 1e-4 (the GPU
run's upstream GEMMs differ from CPU in the last bits, so margins must be wide).

Three groups of fixtures (see `specs` in main): embed 32 / head dim 16 (generic kernels), embed 128 / head dim 64
(the production path: tome_match_keys on the qkv buffer's keys in-model) with proportional attention on and off,
and `apply_duplicate_patch` (videomae.py:154-157, timesformer.py:170-172, motionformer.py:230-232) per family.
    python tests/golden/generate_models.py [fixture names ...]     # no names: all of them
"""
from __future__ import annotations

import importlib
import importlib.util
import json
import math
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import synth  # noqa: E402

REF = os.environ.get("TOME_REFERENCE", "/root/reference")
TAU = 1e-4


def install_stubs():
    def pkg(name, path=None):
        m = types.ModuleType(name)
        m.__path__ = [path] if path else []
        sys.modules[name] = m
        return m

    pkg("slowfast", os.path.join(REF, "slowfast"))
    pkg("slowfast.models", os.path.join(REF, "slowfast", "models"))
    build = types.ModuleType("slowfast.models.build")

    class _Reg:
        def register(self):
            return lambda cls: cls
    build.MODEL_REGISTRY = _Reg()
    sys.modules["slowfast.models.build"] = build
    pkg("tome", os.path.join(REF, "tome"))
    pkg("tome.patch", os.path.join(REF, "tome", "patch"))

    timm = pkg("timm")
    models = pkg("timm.models")
    layers = types.ModuleType("timm.models.layers")

    def drop_path(x, drop_prob=0.0, training=False):
        assert not training or not drop_prob
        return x

    class DropPath(torch.nn.Module):
        def __init__(self, drop_prob=None):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            return drop_path(x, self.drop_prob, self.training)
    layers.drop_path, layers.DropPath = drop_path, DropPath
    layers.to_2tuple = lambda v: v if isinstance(v, tuple) else (v, v)
    layers.trunc_normal_ = torch.nn.init.trunc_normal_
    sys.modules["timm.models.layers"] = layers
    reg = types.ModuleType("timm.models.registry")
    reg.register_model = lambda f: f
    sys.modules["timm.models.registry"] = reg
    resnet = types.ModuleType("timm.models.resnet")
    resnet.resnet26d = resnet.resnet50d = None
    sys.modules["timm.models.resnet"] = resnet
    data = types.ModuleType("timm.data")
    data.IMAGENET_DEFAULT_MEAN, data.IMAGENET_DEFAULT_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    sys.modules["timm.data"] = data
    timm.models, models.layers, models.registry, models.resnet, timm.data = models, layers, reg, resnet, data
    tv = pkg("torchvision")
    tvu = types.ModuleType("torchvision.utils")
    tvu.make_grid = tvu.save_image = None  # names only
    sys.modules["torchvision.utils"] = tvu
    tv.utils = tvu


def certificate(metric: torch.Tensor, r: int, cls: bool):
    m = metric.double()
    m = m / m.norm(dim=-1, keepdim=True)
    s = m[:, ::2] @ m[:, 1::2].transpose(-1, -2)
    if cls:
        s[:, 0, :] = -math.inf
    nmax, _ = s.max(-1)
    order = nmax.argsort(dim=-1, descending=True, stable=True)
    snm = nmax.gather(-1, order)
    gaps = torch.nan_to_num(snm[:, :-1] - snm[:, 1:], nan=math.inf)
    g_src = gaps[:, :r].min().item() if r > 0 else math.inf
    g_unm = gaps[:, r:].min().item() if gaps.shape[1] > r else math.inf
    rows = order[:, :r]
    if s.shape[2] > 1:
        top2 = s.gather(1, rows[..., None].expand(-1, -1, s.shape[2])).topk(2, dim=-1).values
        g_dst = torch.nan_to_num(top2[..., 0] - top2[..., 1], nan=math.inf).min().item()
    else:
        g_dst = math.inf
    return min(g_src, g_dst, g_src if cls else g_unm)


def certificate_parts(metric: torch.Tensor, r: int, cls: bool = False):
    """fp64 margins of one matching, per group: (gap at the r boundary of the sorted row maxima, smallest top-2 gap of
    the r selected rows, per-position gaps of the sorted row maxima [n, T1-1])."""
    m = metric.double()
    m = m / m.norm(dim=-1, keepdim=True)
    s = m[:, ::2] @ m[:, 1::2].transpose(-1, -2)
    if cls:
        s[:, 0, :] = -math.inf
    nmax, _ = s.max(-1)
    order = nmax.argsort(dim=-1, descending=True, stable=True)
    snm = nmax.gather(-1, order)
    gaps = torch.nan_to_num(snm[:, :-1] - snm[:, 1:], nan=math.inf)
    rows = order[:, :r]
    top2 = s.gather(1, rows[..., None].expand(-1, -1, s.shape[2])).topk(2, dim=-1).values
    g_dst = torch.nan_to_num(top2[..., 0] - top2[..., 1], nan=math.inf).min(-1).values
    return gaps[:, r - 1], g_dst, gaps


def closure_vars(fn):
    return dict(zip(fn.__code__.co_freevars, (c.cell_contents for c in fn.__closure__)))


def run_traced(patch_mod, model, clip, r, cls=False, partitions=None, keep_node_max=False):
    """Forward of the patched reference model, recording every matching (metric, indices) -- whichever of the three
    matching functions the patch's mode calls (merge.py:17, :215, :274).  partitions: a list that receives the
    canonical_partition of the source matrix after every layer (trace_source patches in merge / hybrid mode)."""
    layers = []
    names = ("bipartite_soft_matching", "bipartite_soft_matching_drop", "bipartite_soft_matching_hybrid")
    originals = {k: getattr(patch_mod, k) for k in names}

    def record(metric, fn):
        if getattr(fn, "__closure__", None):
            cv = closure_vars(fn)
            unm = cv["unm_idx"] if "unm_idx" in cv else cv["und_idx"]
            rec = dict(T=metric.shape[1], n=metric.shape[0], r=int(cv["r"]), metric=metric.detach().clone(),
                       src=cv["src_idx"][..., 0].numpy().astype(np.int16), unm=unm[..., 0].numpy().astype(np.int16))
            if "dst_idx" in cv:
                rec["dst"] = cv["dst_idx"][..., 0].numpy().astype(np.int16)
            if keep_node_max and "node_max" in cv:
                rec["node_max"] = cv["node_max"].detach().clone()
            layers.append(rec)

    def spy_merge(metric, r_, class_token=False, distill_token=False, mode="merge"):
        pair = originals["bipartite_soft_matching"](metric, r_, class_token, distill_token, mode)
        record(metric, pair[0])
        return pair

    def spy_drop(metric, r_, class_token=False, distill_token=False, mode="drop"):
        drop = originals["bipartite_soft_matching_drop"](metric, r_, class_token, distill_token, mode)
        record(metric, drop if not isinstance(drop, tuple) else None)
        return drop

    def spy_hybrid(metric, r_, class_token=False, distill_token=False, mode="merge", threshold=0.0):
        pair = originals["bipartite_soft_matching_hybrid"](metric, r_, class_token, distill_token, mode, threshold)
        record(metric, pair[0])
        return pair
    patch_mod.bipartite_soft_matching = spy_merge
    patch_mod.bipartite_soft_matching_drop = spy_drop
    patch_mod.bipartite_soft_matching_hybrid = spy_hybrid
    orig_ms = patch_mod.merge_source
    if partitions is not None:
        def spy_source(merge, x, source=None):
            out = orig_ms(merge, x, source)
            if getattr(merge, "__closure__", None):  # (a clamped r of 0 hands do_nothing through merge_source)
                partitions.append(canonical_partition(out))
            return out
        patch_mod.merge_source = spy_source
    try:
        model.r = r
        with torch.no_grad():
            out = model([clip])
    finally:
        for k, v in originals.items():
            setattr(patch_mod, k, v)
        patch_mod.merge_source = orig_ms
    return out, layers


def clip_of(clip_shape, seeds):
    """The fixture's clip batch: clip i is synth.uniform01 of its own seed (clips of a batch are independent, so
    each one is searched on its own -- a batch is certified when each of its clips is)."""
    one = (1,) + tuple(clip_shape[1:])
    return torch.from_numpy(np.concatenate([synth.uniform01(one, s) for s in seeds], axis=0))


def emit(name, model, wrapper_info, clip_shape, r, patch_mod, cls, extra, out_dir, prop_attn, max_attempts=6000,
         l0_set_gap=0.0):
    """Search clip seeds until every layer's matching is certified (margin > TAU) for every clip of the batch, then
    store the traced forward of the batch.  l0_set_gap > 0: every clip's layer-0 gap between the r-th and (r+1)-th
    largest row maximum must also exceed it in every group (so that a 16-bit run of the fixture, whose cosine noise
    is ~3e-3, has source SETS to be held to)."""
    seeds, best = [], None
    for attempt in range(max_attempts):
        seed = 4000 + 7 * attempt
        out, layers = run_traced(patch_mod, model, clip_of(clip_shape, [seed]), r, cls)
        margin = min(certificate(l["metric"], l["r"], cls) for l in layers)
        if l0_set_gap > 0 and float(certificate_parts(layers[0]["metric"], layers[0]["r"], cls)[0].min()) <= l0_set_gap:
            continue
        if best is None or margin > best[0]:
            best = (margin, seed)
        if margin > 1.02 * TAU:  # (a hair above tau: the batched forward's GEMMs may round differently)
            seeds.append(seed)
            if len(seeds) == clip_shape[0]:
                break
    while len(seeds) < clip_shape[0]:
        seeds.append(best[1])
    out, layers = run_traced(patch_mod, model, clip_of(clip_shape, seeds), r, cls)
    margins = [certificate(l["metric"], l["r"], cls) for l in layers]
    arrays = {"logits": out.numpy(), "size": wrapper_info()["size"].numpy()}
    for i, l in enumerate(layers):
        arrays[f"L{i}_src"], arrays[f"L{i}_dst"], arrays[f"L{i}_unm"] = l["src"], l["dst"], l["unm"]
    np.savez_compressed(os.path.join(out_dir, f"models_{name}.npz"), **arrays)
    meta = dict(name=name, clip_shape=list(clip_shape), seeds=seeds, r=r, margin=min(margins), margins=margins,
                certified=bool(min(margins) > TAU), tokens=[l["T"] for l in layers], r_eff=[l["r"] for l in layers],
                groups=[l["n"] for l in layers], prop_attn=prop_attn, **extra)
    if l0_set_gap > 0:
        meta["l0_set_gap"] = float(certificate_parts(layers[0]["metric"], layers[0]["r"], cls)[0].min())
    print({k: v for k, v in meta.items() if k != "param_names"}, flush=True)
    return meta


def canonical_partition(source: torch.Tensor) -> np.ndarray:
    """[n, T, T0] 0/1 source matrix (tome/merge.py:372-384) -> [n, T0] int16: for every original token the smallest
    original token of its merged group.  Independent of the ORDER of the merged rows (which the reference's unstable
    argsort leaves open on near-ties), so two forwards that merged the same tokens agree on it exactly."""
    n, T, T0 = source.shape
    ids = torch.arange(T0).view(1, 1, T0).expand(n, T, T0)
    first = torch.where(source > 0.5, ids, torch.full_like(ids, T0)).min(-1).values  # [n, T]: smallest member
    owner = (source > 0.5).float().transpose(1, 2) @ first.float().unsqueeze(-1)     # [n, T0, 1]: one group per token
    return owner[..., 0].round().numpy().astype(np.int16)


TAU_FULL = 2e-5


def emit_config0(out_dir, attempts=30):
    """BASELINE.json configs[0] at FULL size from the real reference: VideoMAE-B (embed 768, depth 12, 12 heads,
    1568 tokens), 2 synth.uniform01 clips 3x16x224x224, synth.fill_parameters weights, fp32, r = 8, prop_attn False,
    trace_source True.  Stored: logits [2, 400]; tokens per layer; final sizes; after EVERY layer the partition of the
    1568 original tokens into merged groups (canonical_partition: free of the order of the merged rows, which the
    reference's unstable argsort leaves open on near-ties); layer 0's src / dst / unm with per-position certificates.
    Random-init VideoMAE keys are nearly parallel (cosines 0.999..): decision margins are 1e-7 .. 1e-4 where the small
    fixtures have 1e-3.  The two clips are the best of `attempts` seeds by their smallest fp64 margin over all 12
    layers' r-boundary and selected rows' top-2 gaps, and must exceed TAU_FULL = 2e-5 -- two orders above what fp32
    evaluation of these cosines can move (1e-7), so every merge of the forward is defined."""
    vm = importlib.import_module("slowfast.models.videomae_video_model_builder")
    pv = importlib.import_module("tome.patch.videomae")
    cfg = dict(img_size=224, patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True,
               num_classes=400, all_frames=16, tubelet_size=2, init_values=0.0, init_scale=1.0)
    torch.manual_seed(21)
    ref = _wrap(vm.VisionTransformer(norm_layer=lambda d: torch.nn.LayerNorm(d, eps=1e-6), **cfg).eval(), "VideoMAEWrap")
    wseed, r = 77, 8
    names = synth.fill_parameters(ref, wseed)
    pv.apply_patch(ref, prop_attn=False, trace_source=True)
    shape = (2, 3, 16, 224, 224)
    tried = []
    for attempt in range(attempts):
        seed = 9000 + 13 * attempt
        _, layers = run_traced(pv, ref, clip_of(shape, [seed]), r)
        parts = [certificate_parts(l["metric"], l["r"]) for l in layers]
        margin = min(min(float(p[0].min()), float(p[1].min())) for p in parts)
        tried.append((margin, seed))
        print(f"config0 seed {seed}: smallest src/dst margin over 12 layers {margin:.3e}", flush=True)
    seeds = [sd for _, sd in sorted(tried, reverse=True)[:2]]
    # the forward of the pair, recording the source matrix after every layer
    partitions = []
    orig_ms = pv.merge_source

    def spy_source(merge, x, source=None):
        out = orig_ms(merge, x, source)
        partitions.append(canonical_partition(out))
        return out
    pv.merge_source = spy_source
    try:
        out, layers = run_traced(pv, ref, clip_of(shape, seeds), r)
    finally:
        pv.merge_source = orig_ms
    info = ref._tome_info
    parts = [certificate_parts(l["metric"], l["r"]) for l in layers]
    margins = [min(float(p[0].min()), float(p[1].min())) for p in parts]
    assert min(margins) > TAU_FULL, margins
    assert len(partitions) == len(layers) and np.array_equal(partitions[-1], canonical_partition(info["source"]))
    unm_gaps0 = parts[0][2][:, layers[0]["r"]:]                      # gaps between consecutive unmerged rows, layer 0
    unm_ok = torch.ones(unm_gaps0.shape[0], unm_gaps0.shape[1] + 1, dtype=torch.bool)
    unm_ok[:, :-1] &= unm_gaps0 > TAU_FULL
    unm_ok[:, 1:] &= unm_gaps0 > TAU_FULL
    arrays = {"logits": out.numpy(), "size": info["size"].numpy(), "partitions": np.stack(partitions),
              "L0_src": layers[0]["src"], "L0_dst": layers[0]["dst"], "L0_unm": layers[0]["unm"],
              "L0_unm_certified": unm_ok.numpy()}
    np.savez_compressed(os.path.join(out_dir, "models_config0_videomae_b.npz"), **arrays)
    meta = dict(name="config0_videomae_b", host="videomae", cfg=cfg, clip_shape=list(shape), seeds=seeds, r=r,
                weight_seed=wseed, param_names=names, prop_attn=False, tokens=[l["T"] for l in layers],
                r_eff=[l["r"] for l in layers], margins_src_dst=margins, margin=min(margins), tau=TAU_FULL,
                certified=bool(min(margins) > TAU_FULL), unm_positions_certified_layer0=int(unm_ok.sum()),
                seeds_tried=[[sd, m] for m, sd in tried],
                what="BASELINE.json configs[0]: the reference's tome/patch/videomae.py over "
                     "slowfast/models/videomae_video_model_builder.py at full size, fp32, CPU, trace_source=True")
    print({k: v for k, v in meta.items() if k not in ("param_names", "seeds_tried")}, flush=True)
    return meta


def l0_certificates(metric: torch.Tensor, r: int, tau: float) -> dict:
    """Per-position fp64 certificates of one matching (no class token): which of the reference's layer-0 answers are
    DEFINED, i.e. decided by margins above `tau` (the reference's argsort is unstable and its bmm sums in another
    order than any other machine's).  Returns numpy bool arrays
      set_ok  [n]        the r selected rows as a SET (gap at the r boundary of the sorted row maxima)
      src_ok  [n, r]     position k of src_idx (gaps to both neighbours in the sorted order)
      dst_ok  [n, r]     the destination of the source at position k (top-2 gap of that row)
      unm_ok  [n, T1-r]  position k of unm_idx
    and the smallest boundary / destination margin per group."""
    m = metric.double()
    m = m / m.norm(dim=-1, keepdim=True)
    s = m[:, ::2] @ m[:, 1::2].transpose(-1, -2)
    n, t1, t2 = s.shape
    nmax, _ = s.max(-1)
    order = nmax.argsort(dim=-1, descending=True, stable=True)
    snm = nmax.gather(-1, order)
    gaps = torch.nan_to_num(snm[:, :-1] - snm[:, 1:], nan=math.inf)              # [n, t1-1]
    big = torch.full((n, 1), math.inf, dtype=gaps.dtype)
    left = torch.cat([big, gaps], 1)                                             # gap to the previous position
    right = torch.cat([gaps, big], 1)                                            # gap to the next position
    pos_ok = (left > tau) & (right > tau)                                        # [n, t1]
    set_gap = gaps[:, r - 1] if r < t1 else torch.full((n,), math.inf, dtype=gaps.dtype)
    rows = order[:, :r]
    if t2 > 1:
        top2 = s.gather(1, rows[..., None].expand(-1, -1, t2)).topk(2, dim=-1).values
        dst_gap = torch.nan_to_num(top2[..., 0] - top2[..., 1], nan=math.inf)
    else:
        dst_gap = torch.full((n, r), math.inf, dtype=gaps.dtype)
    return dict(set_ok=(set_gap > tau).numpy(), src_ok=pos_ok[:, :r].numpy(), dst_ok=(dst_gap > tau).numpy(),
                unm_ok=pos_ok[:, r:].numpy(), set_gap=set_gap.numpy(), dst_gap=dst_gap.min(-1).values.numpy())


def _full_size_model(family):
    """The reference's own model classes at the sizes BASELINE.json names (embed 768, depth 12, 12 heads), random
    init by name (synth.fill_parameters).  Returns (patched-model object, patch module, cfg for the host side, clip
    shape of ONE clip, weight seed, parameter names)."""
    ln = lambda d: torch.nn.LayerNorm(d, eps=1e-6)  # noqa: E731
    if family == "videomae":
        vm = importlib.import_module("slowfast.models.videomae_video_model_builder")
        patch = importlib.import_module("tome.patch.videomae")
        cfg = dict(img_size=224, patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True,
                   num_classes=400, all_frames=16, tubelet_size=2, init_values=0.0, init_scale=1.0)
        torch.manual_seed(21)
        model = _wrap(vm.VisionTransformer(norm_layer=ln, **cfg).eval(), "VideoMAEWrap")
        shape, wseed = (1, 3, 16, 224, 224), 77
    elif family == "timesformer":
        tsm = importlib.import_module("slowfast.models.timesformer")
        patch = importlib.import_module("tome.patch.timesformer")
        cfg = dict(img_size=224, patch_size=16, num_classes=400, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4,
                   qkv_bias=True, num_frames=8, attention_type="divided_space_time")
        torch.manual_seed(22)
        model = _wrap(tsm.VisionTransformer(norm_layer=ln, drop_path_rate=0.0, **cfg).eval(), "TimeSformerWrap")
        shape, wseed = (1, 3, 8, 224, 224), 78
    elif family == "motionformer":
        mb = importlib.import_module("slowfast.models.motionformer_video_model_builder")
        patch = importlib.import_module("tome.patch.motionformer")
        mcfg = SimpleNamespace(
            DATA=SimpleNamespace(TRAIN_CROP_SIZE=224), MODEL=SimpleNamespace(NUM_CLASSES=400),
            EPICKITCHENS=SimpleNamespace(NUM_CLASSES=None),
            MOTIONFORMER=SimpleNamespace(PATCH_SIZE=16, CHANNELS=3, EMBED_DIM=768, DEPTH=12, NUM_HEADS=12, MLP_RATIO=4,
                                         QKV_BIAS=True, DROP=0.0, DROP_PATH=0.0, HEAD_DROPOUT=0.0, VIDEO_INPUT=True,
                                         TEMPORAL_RESOLUTION=8, USE_MLP=True, ATTN_DROPOUT=0.0, HEAD_ACT="tanh",
                                         PATCH_SIZE_TEMP=2, POS_DROPOUT=0.0, POS_EMBED="separate",
                                         ATTN_LAYER="trajectory", USE_ORIGINAL_TRAJ_ATTN_CODE=True,
                                         APPROX_ATTN_TYPE="none", APPROX_ATTN_DIM=128))
        cfg = dict(img_size=224, patch_size=16, patch_size_temp=2, temporal_resolution=8, num_classes=400,
                   embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0, qkv_bias=True, use_mlp=True, head_act="tanh")
        torch.manual_seed(23)
        model = mb.Motionformer(mcfg).eval()
        shape, wseed = (1, 3, 16, 224, 224), 79
    else:
        raise ValueError(family)
    names = synth.fill_parameters(model, wseed)  # (re-randomises Motionformer's zeroed patch_embed_3d)
    return model, patch, cfg, shape, wseed, names


FULL_SPECS = {
    # key: (family, r, prop_attn, BASELINE.json config it pins)
    "config1_videomae_b_r16": ("videomae", 16, False, "configs[1]: VideoMAE-B 16x224x224, r=16 (the metric's config)"),
    "config2_timesformer_r8": ("timesformer", 8, True, "configs[2]: TimeSformer divST 8x224, r=8"),
    "config2_timesformer_r16": ("timesformer", 16, True, "configs[2]: TimeSformer divST 8x224, r=16"),
    "config2_timesformer_r32": ("timesformer", 32, True, "configs[2]: TimeSformer divST 8x224, r=32 (196 -> ... -> 1)"),
    "config4_motionformer_r16": ("motionformer", 16, True, "configs[4]: Motionformer 224 16x4, r=16"),
}


def emit_full(out_dir, key, attempts=24, clips=2):
    """One BASELINE.json config at FULL size from the real reference (as emit_config0, generalised): the reference's
    tome/patch/<family>.py over its slowfast model class, `clips` synth.uniform01 clips, fill_parameters weights, fp32,
    CPU, trace_source=True, prop_attn as the reference defaults it for the family.  Stored: logits; tokens / r_eff /
    groups per layer; final sizes; the canonical partition of every group's original tokens after EVERY layer; layer
    0's src / dst / unm with per-position certificates (l0_certificates, tau = TAU_FULL); per layer the smallest
    boundary / destination margin.  The clips are the best of `attempts` seeds by the number of layer-0 groups that
    are fully certified, then by the smallest margin over all layers."""
    family, r, prop, what = FULL_SPECS[key]
    model, patch, cfg, shape1, wseed, names = _full_size_model(family)
    patch.apply_patch(model, prop_attn=prop, trace_source=True)
    info_of = lambda: model._tome_info  # noqa: E731
    tried = []
    for attempt in range(attempts):
        seed = 9100 + 13 * attempt + 1000 * (sorted(FULL_SPECS).index(key))
        _, layers = run_traced(patch, model, clip_of(shape1, [seed]), r)
        c0 = l0_certificates(layers[0]["metric"], layers[0]["r"], TAU_FULL)
        groups_ok = int((c0["set_ok"] & c0["dst_ok"].all(-1)).sum())
        margin = min(min(float(c["set_gap"].min()), float(c["dst_gap"].min()))
                     for c in (l0_certificates(l["metric"], l["r"], TAU_FULL) for l in layers))
        tried.append((groups_ok, margin, seed))
        print(f"{key} seed {seed}: layer-0 groups fully certified {groups_ok}/{layers[0]['n']}, smallest src/dst "
              f"margin over {len(layers)} layers {margin:.3e}", flush=True)
    seeds = [sd for _, _, sd in sorted(tried, reverse=True)[:clips]]
    shape = (clips,) + tuple(shape1[1:])
    partitions = []
    out, layers = run_traced(patch, model, clip_of(shape, seeds), r, partitions=partitions)
    info = info_of()
    assert len(partitions) == len(layers) and np.array_equal(partitions[-1], canonical_partition(info["source"]))
    c0 = l0_certificates(layers[0]["metric"], layers[0]["r"], TAU_FULL)
    per_layer = [l0_certificates(l["metric"], l["r"], TAU_FULL) for l in layers]
    margins = [min(float(c["set_gap"].min()), float(c["dst_gap"].min())) for c in per_layer]
    group_ok = c0["set_ok"] & c0["dst_ok"].all(-1)
    arrays = {"logits": out.numpy(), "size": info["size"].numpy(),
              "L0_src": layers[0]["src"], "L0_dst": layers[0]["dst"], "L0_unm": layers[0]["unm"],
              "L0_set_ok": c0["set_ok"], "L0_src_ok": c0["src_ok"], "L0_dst_ok": c0["dst_ok"], "L0_unm_ok": c0["unm_ok"],
              "L0_group_ok": group_ok}
    for i, p in enumerate(partitions):     # (the groups keep their original-token count T0, the layers differ in rows)
        arrays[f"P{i}"] = p
    np.savez_compressed(os.path.join(out_dir, f"models_{key}.npz"), **arrays)
    meta = dict(name=key, host=family, cfg=cfg, clip_shape=list(shape), seeds=seeds, r=r, weight_seed=wseed,
                param_names=names, prop_attn=prop, tokens=[l["T"] for l in layers], r_eff=[l["r"] for l in layers],
                groups=[l["n"] for l in layers], margins_src_dst=margins, tau=TAU_FULL,
                l0_groups_certified=int(group_ok.sum()), l0_groups=int(group_ok.size),
                l0_src_positions_certified=int(c0["src_ok"].sum()), l0_dst_certified=int(c0["dst_ok"].sum()),
                l0_unm_positions_certified=int(c0["unm_ok"].sum()), l0_unm_positions=int(c0["unm_ok"].size),
                seeds_tried=[[sd, g, m] for g, m, sd in tried],
                what=f"BASELINE.json {what}: the reference's tome/patch/{family}.py over its slowfast model at full "
                     "size, fp32, CPU, trace_source=True")
    print({k: v for k, v in meta.items() if k not in ("param_names", "seeds_tried")}, flush=True)
    return meta


def emit_mode(name, build, patch_mod, r, mode, prop, out_dir, max_attempts=3000):
    """A reduced-width fixture of the patch-level DROP / HYBRID glue (tome/patch/videomae.py:102-151,
    timesformer.py:111-185, motionformer.py:172-245): apply_patch(mode=..., trace_source=True[, threshold]) on the
    reference's model, every layer's matching certified (margin > TAU; hybrid: no selected edge within TAU of the
    threshold either).  The hybrid threshold is the median of layer 0's selected edges' scores in a first plain forward of
    the chosen clip, so that some destinations keep their own contribution and some lose it (merge.py:326).  Stored:
    logits, final sizes (drop: the fp32 ones the patch resets them to), the final source matrix (drop: `drop(eye)`),
    per layer src / unm (/ dst) and, for hybrid, the per-edge keep flags in src_idx order."""
    model, cfg, clip_shape, wseed = build(128)
    names = synth.fill_parameters(model, wseed)
    patch_mod.apply_patch(model, prop_attn=prop, mode=mode, trace_source=True, threshold=-1.0)
    info = model._tome_info
    one = (1,) + tuple(clip_shape[1:])
    seeds, thr = [], None

    def margins_of(layers, threshold):
        ms = [certificate(l["metric"], l["r"], False) for l in layers]
        if threshold is not None:
            for l in layers:
                sel = l["node_max"].sort(-1, descending=True).values[:, :l["r"]].double()
                ms.append(float((sel - threshold).abs().min()))
        return ms
    for attempt in range(max_attempts):
        seed = 5000 + 11 * attempt
        if mode == "hybrid":
            cand = thr
            if cand is None:  # fixed by the first clip that certifies; later clips must certify against the same value
                info["threshold"] = -1.0  # (the config default, custom_config.py:166: equal to plain merge)
                _, layers = run_traced(patch_mod, model, clip_of(one, [seed]), r, keep_node_max=True)
                # the median of LAYER 0's selected edges (scores grow with depth: a median over all layers keeps
                # nothing in layer 0): half of the first layer's destinations lose their own contribution, a few do later
                l0 = layers[0]
                sel0 = l0["node_max"].sort(-1, descending=True).values[:, :l0["r"]].flatten().double().sort().values
                cand = float(0.5 * (sel0[sel0.numel() // 2 - 1] + sel0[sel0.numel() // 2]))  # between two scores
            info["threshold"] = cand
            _, layers = run_traced(patch_mod, model, clip_of(one, [seed]), r, keep_node_max=True)
            kept0 = (layers[0]["node_max"].sort(-1, descending=True).values[:, :layers[0]["r"]] >= cand).sum().item()
            if min(margins_of(layers, cand)) <= 1.02 * TAU or not (0 < kept0 < layers[0]["n"] * layers[0]["r"]):
                continue
            thr = cand
        else:
            _, layers = run_traced(patch_mod, model, clip_of(one, [seed]), r, keep_node_max=True)
            if min(margins_of(layers, None)) <= 1.02 * TAU:
                continue
        seeds.append(seed)
        if len(seeds) == clip_shape[0]:
            break
    assert len(seeds) == clip_shape[0], f"{name}: no certified clip in {max_attempts} attempts"
    info["threshold"] = thr if mode == "hybrid" else 0.0
    out, layers = run_traced(patch_mod, model, clip_of(clip_shape, seeds), r, keep_node_max=True)
    margins = margins_of(layers, thr)
    arrays = {"logits": out.numpy(), "size": info["size"].numpy(), "source": info["source"].numpy().astype(np.uint8)}
    kept = []
    for i, l in enumerate(layers):
        arrays[f"L{i}_src"], arrays[f"L{i}_unm"] = l["src"], l["unm"]
        if "dst" in l:
            arrays[f"L{i}_dst"] = l["dst"]
        if mode == "hybrid":
            keep = l["node_max"].sort(-1, descending=True).values[:, :l["r"]] >= thr
            arrays[f"L{i}_keep"] = keep.numpy()
            kept.append(int(keep.sum()))
    np.savez_compressed(os.path.join(out_dir, f"models_{name}.npz"), **arrays)
    meta = dict(name=name, host=name.split("_")[0], cfg=cfg, weight_seed=wseed, param_names=names, mode=mode,
                clip_shape=list(clip_shape), seeds=seeds, r=r, margin=min(margins), margins=margins,
                certified=bool(min(margins) > TAU), tokens=[l["T"] for l in layers], r_eff=[l["r"] for l in layers],
                groups=[l["n"] for l in layers], prop_attn=prop, trace_source=True, size_dtype=str(info["size"].dtype))
    if mode == "hybrid":
        meta["threshold"], meta["edges_kept"] = thr, kept
        meta["edges"] = [l["n"] * l["r"] for l in layers]
    print({k: v for k, v in meta.items() if k != "param_names"}, flush=True)
    return meta


def _wrap(inner, tag):
    wrap = torch.nn.Module()
    wrap.__class__ = type(tag, (torch.nn.Module,), {"forward": lambda self, x: self.model(x)})
    wrap.model = inner
    return wrap


def main():
    install_stubs()
    torch.manual_seed(0)
    metas = []
    vm = importlib.import_module("slowfast.models.videomae_video_model_builder")
    pv = importlib.import_module("tome.patch.videomae")
    tsm = importlib.import_module("slowfast.models.timesformer")
    pt = importlib.import_module("tome.patch.timesformer")
    mb = importlib.import_module("slowfast.models.motionformer_video_model_builder")
    pm = importlib.import_module("tome.patch.motionformer")
    ln = lambda d: torch.nn.LayerNorm(d, eps=1e-6)  # noqa: E731

    def videomae(embed):
        cfg = dict(img_size=32, patch_size=8, embed_dim=embed, depth=4, num_heads=2, mlp_ratio=4, qkv_bias=True,
                   num_classes=10, all_frames=8, tubelet_size=2, init_values=0.0, init_scale=1.0)
        torch.manual_seed(11)
        wrap = _wrap(vm.VisionTransformer(norm_layer=ln, **cfg).eval(), "VideoMAEWrap")
        return wrap, cfg, (2, 3, 8, 32, 32), 101

    def timesformer(embed):
        cfg = dict(img_size=48, patch_size=8, num_classes=10, embed_dim=embed, depth=4, num_heads=2, mlp_ratio=4,
                   qkv_bias=True, num_frames=4, attention_type="divided_space_time")
        torch.manual_seed(12)
        wrap = _wrap(tsm.VisionTransformer(norm_layer=ln, drop_path_rate=0.0, **cfg).eval(), "TimeSformerWrap")
        return wrap, cfg, (2, 3, 4, 48, 48), 202

    def motionformer(embed):
        mcfg = SimpleNamespace(
            DATA=SimpleNamespace(TRAIN_CROP_SIZE=224), MODEL=SimpleNamespace(NUM_CLASSES=10),
            EPICKITCHENS=SimpleNamespace(NUM_CLASSES=None),
            MOTIONFORMER=SimpleNamespace(PATCH_SIZE=32, CHANNELS=3, EMBED_DIM=embed, DEPTH=3, NUM_HEADS=2, MLP_RATIO=4,
                                         QKV_BIAS=True, DROP=0.0, DROP_PATH=0.0, HEAD_DROPOUT=0.0, VIDEO_INPUT=True,
                                         TEMPORAL_RESOLUTION=4, USE_MLP=True, ATTN_DROPOUT=0.0, HEAD_ACT="tanh",
                                         PATCH_SIZE_TEMP=2, POS_DROPOUT=0.0, POS_EMBED="separate",
                                         ATTN_LAYER="trajectory", USE_ORIGINAL_TRAJ_ATTN_CODE=True,
                                         APPROX_ATTN_TYPE="none", APPROX_ATTN_DIM=128))
        cfg = dict(img_size=224, patch_size=32, patch_size_temp=2, temporal_resolution=4, num_classes=10,
                   embed_dim=embed, depth=3, num_heads=2, mlp_ratio=4.0, qkv_bias=True, use_mlp=True, head_act="tanh")
        torch.manual_seed(13)
        return mb.Motionformer(mcfg).eval(), cfg, (2, 3, 8, 224, 224), 303

    families = {"videomae": (videomae, pv, 5), "timesformer": (timesformer, pt, 6), "motionformer": (motionformer, pm, 5)}
    # layer-0 r-boundary gap demanded of the three VideoMAE head-dim-64 fixtures (16-bit cosine noise is ~3e-3: the
    # bf16 run of these fixtures then has every group's source SET to answer for, not only destinations)
    L0_GAP = {"videomae_hd64_prop0": 1e-2, "videomae_hd64_prop1": 1e-2, "videomae_hd64_dup": 1e-2,
              "videomae_hd64_concat": 1e-2}
    # (fixture name, family, embed width, prop_attn, duplicate (layer, quantity) or None)
    #   embed 32  -> head dim 16: the first round's fixtures (generic kernels)
    #   embed 128 -> head dim 64: the production path in-model (tome_match_keys on the per-head keys of the qkv buffer,
    #                tome_prop_attention, the LayerNorm-fused merges in the 16-bit run of the same fixture)
    #   dup       -> apply_duplicate_patch(model, 1, 2) in front of apply_patch, r given per layer
    specs = [("videomae_prop0", "videomae", 32, False, None), ("videomae_prop1", "videomae", 32, True, None),
             ("timesformer_prop1", "timesformer", 32, True, None), ("timesformer_prop0", "timesformer", 32, False, None),
             ("motionformer_prop1", "motionformer", 32, True, None), ("motionformer_prop0", "motionformer", 32, False, None),
             ("videomae_hd64_prop0", "videomae", 128, False, None), ("videomae_hd64_prop1", "videomae", 128, True, None),
             ("timesformer_hd64_prop1", "timesformer", 128, True, None),
             ("timesformer_hd64_prop0", "timesformer", 128, False, None),
             ("motionformer_hd64_prop1", "motionformer", 128, True, None),
             ("motionformer_hd64_prop0", "motionformer", 128, False, None),
             ("videomae_hd64_dup", "videomae", 128, False, (1, 2)), ("timesformer_hd64_dup", "timesformer", 128, True, (1, 2)),
             ("motionformer_hd64_dup", "motionformer", 128, True, (1, 2)),
             # head_aggregation="concat" (videomae.py:74-75; experiments.sh:164-169): the metric is the keys of all
             # heads side by side (D = 128 here, 768 at full size)
             ("videomae_hd64_concat", "videomae", 128, False, None)]
    only = set(sys.argv[1:])
    man_path = os.path.join(HERE, "manifest.json")
    manifest = json.load(open(man_path))
    old = {m["name"]: m for m in manifest.get("models", [])}
    for name, fam, embed, prop, dup in specs:
        if only and name not in only:
            if name in old:
                metas.append(old[name])
            continue
        build, patch_mod, r = families[fam]
        model, cfg, clip_shape, wseed = build(embed)
        names = synth.fill_parameters(model, wseed)  # (the reference zero-inits Motionformer's tubelet conv: duplicate tokens)
        extra = dict(host=fam, cfg=cfg, weight_seed=wseed, param_names=names)
        r_arg = r
        if dup is not None:
            patch_mod.apply_duplicate_patch(model, layer_to_duplicate=dup[0], quantity=dup[1])
            # the per-layer list tools/test_net.py:274 builds, given directly (its tuple form makes parse_r raise)
            r_arg = [0] * dup[0] + [r] * dup[1] + [0] * (cfg["depth"] - 1 - dup[0])
            extra["duplicate"] = list(dup)
        if name.endswith("_concat"):
            extra["head_aggregation"] = "concat"
            patch_mod.apply_patch(model, prop_attn=prop, head_aggregation="concat")
        else:
            patch_mod.apply_patch(model, prop_attn=prop)
        metas.append(emit(name, model, lambda: model._tome_info, clip_shape, r_arg, patch_mod, False, extra, HERE, prop,
                          l0_set_gap=L0_GAP.get(name, 0.0)))
    manifest["models"] = metas
    if not only or "config0_videomae_b" in only:
        manifest["config0"] = emit_config0(HERE)
    mode_old = {m["name"]: m for m in manifest.get("modes", [])}
    mode_metas = []
    for fam in ("videomae", "timesformer", "motionformer"):
        for mode in ("drop", "hybrid"):
            name = f"{fam}_hd64_{mode}"
            if only and name not in only:
                if name in mode_old:
                    mode_metas.append(mode_old[name])
                continue
            build, patch_mod, r = families[fam]
            mode_metas.append(emit_mode(name, build, patch_mod, r, mode, fam != "videomae", HERE))
    manifest["modes"] = mode_metas
    full_old = {m["name"]: m for m in manifest.get("full", [])}
    manifest["full"] = [emit_full(HERE, key, attempts=int(os.environ.get("TOME_FULL_ATTEMPTS", "24")))
                        if (not only or key in only) else full_old[key]
                        for key in FULL_SPECS if (not only or key in only or key in full_old)]
    manifest["models_tau"] = TAU
    with open(man_path, "w") as f:
        json.dump(manifest, f, indent=1)


if __name__ == "__main__":
    main()
