"""Patched-model parity on the GPU (`-m gpu`): this repository's host models + `tome.patch.*` (HIP merge
path) against golden vectors of the reference's own patched models (tests/golden/generate_models.py ran
tome/patch/{videomae,timesformer,motionformer}.py over the slowfast model files on CPU).

Weights are filled by name (tests/synth.fill_parameters) on both sides, the clip comes from its seed.
Checked per layer: token counts, r_eff, src/dst/unm indices (bit-exact: the fixtures' seeds were chosen
so that every layer's decision margins are >= 5e-5, far above the fp32 GEMM noise between CPU and GPU);
at the end: token sizes exactly, logits within 2e-4 (fp32 attention/MLP run on different BLAS).
ViViT has no reference fixture (parity unpinned); it gets structural checks only.
"""
import os
import sys

import numpy as np
import pytest
import torch

import golden_io as G
import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _hosts():
    import tome  # noqa: F401
    from hosts import motionformer, timesformer, videomae, vivit
    return tome, dict(videomae=videomae, timesformer=timesformer, motionformer=motionformer, vivit=vivit)


def _build(meta):
    tome, H = _hosts()
    cfg = dict(meta["cfg"])
    if meta["host"] == "videomae":
        model = H["videomae"].VideoMAE(num_frames=cfg.pop("all_frames"), num_classes=cfg.pop("num_classes"),
                                       tubelet_size=cfg.pop("tubelet_size"), **cfg)
        patch = tome.patch.videomae
    elif meta["host"] == "timesformer":
        model = H["timesformer"].TimeSformer(num_frames=cfg.pop("num_frames"), num_classes=cfg.pop("num_classes"),
                                             attention_type=cfg.pop("attention_type"), **cfg)
        patch = tome.patch.timesformer
    else:
        model = H["motionformer"].Motionformer(**cfg)
        patch = tome.patch.motionformer
    names = synth.fill_parameters(model, meta["weight_seed"])
    # same parameter names as the reference model (its checkpoints load); the reference's Motionformer also
    # carries an unused 2-D patch_embed
    extra = set(meta["param_names"]) - set(names)
    assert set(names) <= set(meta["param_names"]) and all(n.startswith("patch_embed.") for n in extra), extra
    return tome, model.to(DEV).eval(), patch


def _trace(tome, model, clip, r, keep_metric=False):
    """Run the patched model, recording the plan of every matching (and, on request, the metric it saw) -- whichever
    of the three matching functions the patch's mode calls."""
    from tome.patch import _common
    plans = []
    names = ("bipartite_soft_matching", "bipartite_soft_matching_drop", "bipartite_soft_matching_hybrid")
    originals = {k: getattr(_common, k) for k in names}

    def spy_of(orig):
        def spy(metric, *a, **kw):
            got = orig(metric, *a, **kw)
            fn = got[0] if isinstance(got, tuple) else got
            if hasattr(fn, "plan"):
                rec = (metric.shape, fn.plan)
                if keep_metric:
                    m = metric.materialize() if hasattr(metric, "materialize") else metric
                    rec += (m.detach().double(),)
                plans.append(rec)
            return got
        return spy
    for k in names:
        setattr(_common, k, spy_of(originals[k]))
    try:
        model.r = r
        with torch.no_grad():
            out = model([clip])
    finally:
        for k in names:
            setattr(_common, k, originals[k])
    return out, plans


class _CallCounter:
    """Counts calls of the fused entry points of tome._abi during a forward (is the production path the one
    that ran?)."""
    NAMES = ("match_keys", "match", "prop_attention", "merge_wavg_ln", "merge_wavg_regrouped", "merge_wavg",
             "add_layernorm", "trajectory_mix", "short_attention")

    def __init__(self, monkeypatch):
        from tome import _abi
        self.n = {k: 0 for k in self.NAMES}
        for k in self.NAMES:
            monkeypatch.setattr(_abi, k, self._wrap(k, getattr(_abi, k)))

    def _wrap(self, k, fn):
        def counted(*a, **kw):
            self.n[k] += 1
            return fn(*a, **kw)
        return counted


def _patched(meta, dtype=torch.float32):
    tome, model, patch = _build(meta)
    if meta.get("duplicate"):
        getattr(tome.patch, "duplicate_" + meta["host"])(model, *meta["duplicate"])
    if meta.get("head_aggregation"):
        patch(model, prop_attn=meta["prop_attn"], head_aggregation=meta["head_aggregation"])
    else:
        patch(model, prop_attn=meta["prop_attn"])
    return tome, model.to(dtype)


def _clip(meta):
    """The fixture's clip batch: one synth.uniform01 clip per stored seed (tests/golden/generate_models.py)."""
    one = (1,) + tuple(meta["clip_shape"][1:])
    return torch.from_numpy(np.concatenate([synth.uniform01(one, s) for s in meta["seeds"]], axis=0)).to(DEV)


def _r_of(meta):
    return list(meta["r"]) if isinstance(meta["r"], list) else meta["r"]


@pytest.mark.parametrize("meta", G.manifest()["models"], ids=lambda m: m["name"])
def test_patched_model_matches_reference(meta, monkeypatch):
    """fp32 forward of the host model + tome.patch.* against the reference's own patched model (same weights by
    name, same clip): per layer token counts, r_eff and ALL THREE index tensors exactly, final sizes exactly, logits
    within 2e-4.  The head-dim-64 fixtures run the production matching in-model (tome_match_keys reads the per-head
    keys of the qkv buffer; asserted by the call counter); `*_dup` fixtures go through apply_duplicate_patch."""
    assert meta["certified"], "fixture margins below the generator's tau: re-seed it (tests/golden/generate_models.py)"
    tome, model = _patched(meta)
    calls = _CallCounter(monkeypatch)
    clip = _clip(meta)
    out, plans = _trace(tome, model, clip, _r_of(meta))
    z = np.load(os.path.join(G.GOLDEN, f"models_{meta['name']}.npz"))
    assert [s[1] for s, _ in plans] == meta["tokens"]
    assert [p.r for _, p in plans] == meta["r_eff"]
    assert [s[0] for s, _ in plans] == meta["groups"]
    for i, (_, p) in enumerate(plans):
        np.testing.assert_array_equal(p.src_idx.cpu().numpy()[..., 0], z[f"L{i}_src"], err_msg=f"layer {i} src")
        np.testing.assert_array_equal(p.dst_idx.cpu().numpy()[..., 0], z[f"L{i}_dst"], err_msg=f"layer {i} dst")
        np.testing.assert_array_equal(p.unm_idx.cpu().numpy()[..., 0], z[f"L{i}_unm"], err_msg=f"layer {i} unm")
    np.testing.assert_array_equal(model._tome_info["size"].cpu().numpy(), z["size"])
    np.testing.assert_allclose(out.cpu().numpy(), z["logits"], atol=2e-4, rtol=1e-4)
    head_dim = meta["cfg"]["embed_dim"] // meta["cfg"]["num_heads"]
    if head_dim == 64 and meta.get("head_aggregation") == "concat":
        # the metric is the heads' keys side by side (videomae.py:74-75): a strided [B, N, H*64] view of the qkv buffer
        # handed to tome_match as it lies there
        assert calls.n["match"] == len(plans) and calls.n["match_keys"] == 0, calls.n
        assert all(s[2] == meta["cfg"]["embed_dim"] for s, _ in plans)
    elif head_dim == 64:
        # the keys were averaged over the heads inside the matching kernel, never as a torch tensor
        assert calls.n["match_keys"] == len(plans) and calls.n["match"] == 0, calls.n
    if meta.get("duplicate"):
        lay, q = meta["duplicate"]
        blocks = model.blocks if meta["host"] == "motionformer" else model.model.blocks
        assert len(blocks) == meta["cfg"]["depth"] + q - 1 and len(plans) == q


_HD64 = [m for m in G.manifest()["models"] if m["cfg"]["embed_dim"] // m["cfg"]["num_heads"] == 64]
BF16_LOGIT_TOL = 0.08  # ceiling for a fixture that has no measured value yet (of the largest |logit| of the fixture)


def _measured():
    """tests/golden/measured.json: per fixture, what a run on MI355X measured (bf16 / fp32 logit error against the
    reference's logits, share of tokens in the reference's merged group per layer).  The tests hold a run to 1.5x the
    measured error (and to the measured agreement minus 2 points) instead of one global tolerance; a run with
    TOME_RECORD_MEASURED=1 writes what it sees to gpurun_out/measured_seen.json (tools/update_measured.py folds the
    largest values of several runs into the tracked file)."""
    path = os.path.join(G.GOLDEN, "measured.json")
    if not os.path.exists(path):
        return {}
    import json
    with open(path) as f:
        return json.load(f)


def _note_measured(name, **values):
    if os.environ.get("TOME_RECORD_MEASURED") != "1":
        return
    import json
    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                       "gpurun_out", "measured_seen.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    seen = json.load(open(out)) if os.path.exists(out) else {}
    seen.setdefault(name, {}).update(values)
    with open(out, "w") as f:
        json.dump(seen, f, indent=1)


# Runs of one build are bit-reproducible (three runs in a row, round 4: identical errors), so the factor is head-room
# for the measurement switches, which move rounding points on purpose (README table; `tools/test_switches.sh` runs
# this file under each: TOME_FUSE_NEXT=0 / TOME_ATTN_KERNEL=0 reach 1.52x on two TimeSformer fixtures).
_MEASURED_FACTOR = 1.5 if not any(os.environ.get(k) == "0" for k in os.environ if k.startswith("TOME_")) else 2.0


def _logit_tol(name, key, scale, ceiling):
    """1.5 x the error a run on MI355X measured for this fixture (2 x under a measurement switch; never below 0.2 % of
    the largest |logit|: the last bits of the library GEMMs differ between library builds), or `ceiling * scale` while
    no measurement is stored."""
    m = _measured().get(name, {}).get(key)
    if m is None:
        return ceiling * scale
    return max(_MEASURED_FACTOR * m, 2e-3 * scale)


@pytest.mark.parametrize("meta", _HD64, ids=lambda m: m["name"])
def test_production_path_bf16_against_reference_fixture(meta, monkeypatch):
    """The SAME reference fixtures, run the way the benchmark runs: bf16 weights and clips, every fused kernel on
    (tome_match_keys, tome_prop_attention, tome_merge_wavg[_regrouped]_ln with the residual, tome_add_layernorm,
    Motionformer's tome_trajectory_mix).  A 16-bit forward cannot reproduce fp32 decisions whose margin is below
    its own rounding noise, so the comparison is certificate-aware:
      * noise = max |cos_bf16 - cos_fp32| over the layer-0 score matrix, both evaluated in fp64 from the metric each
        run handed to the matching (measured here, not assumed);
      * a group's layer-0 SOURCE SET must equal the reference's when the gap between the r-th and (r+1)-th largest
        row maximum exceeds 2*noise; a source's DESTINATION must equal the reference's when that row's top-2 gap
        exceeds 2*noise;
      * token counts and r_eff of every layer exactly; logits within BF16_LOGIT_TOL * max|logit| of the fp32
        fixture (stated tolerance: bf16 has 8 bits of mantissa, the models are three to five blocks deep, and tokens
        whose margin is inside the noise may merge differently)."""
    z = np.load(os.path.join(G.GOLDEN, f"models_{meta['name']}.npz"))
    clip32 = _clip(meta)
    tome, model32 = _patched(meta)
    _, plans32 = _trace(tome, model32, clip32, _r_of(meta), keep_metric=True)
    tome, model16 = _patched(meta, torch.bfloat16)
    calls = _CallCounter(monkeypatch)
    out16, plans16 = _trace(tome, model16, clip32.bfloat16(), _r_of(meta), keep_metric=True)
    assert [s[1] for s, *_ in plans16] == meta["tokens"] and [p.r for _, p, _ in plans16] == meta["r_eff"]
    # the production kernels ran (asserted for the default configuration; under a TOME_* measurement switch this test
    # still checks the decisions and the logits of whatever path the switch selects -- tools/test_switches.sh)
    n_layers = len(plans16)
    from tome.patch import _common
    if all((_common._FUSE_LN, _common._FUSE_ADD, _common._FUSE_NEXT, _common._ATTN_KERNEL)):
        _assert_production_calls(calls, meta, n_layers)
    _check_layer0_and_logits(meta, z, plans32, plans16, out16)


def _assert_production_calls(calls, meta, n_layers, n_match=None):
    """n_layers: layers that merged; n_match: layers that ASKED for a matching (a layer whose clamped r is 0 -- one
    token left per group at the end of an r = 32 walk -- still calls the matching entry, which answers `nothing`)."""
    n_match = n_layers if n_match is None else n_match
    assert calls.n["prop_attention"] > 0 and calls.n["add_layernorm"] > 0, calls.n
    if meta.get("head_aggregation") == "concat":
        assert calls.n["match"] == n_match and calls.n["match_keys"] == 0, calls.n
    else:
        assert calls.n["match_keys"] == n_match and calls.n["match"] == 0, calls.n
    if meta["host"] == "motionformer":
        assert calls.n["trajectory_mix"] > 0, calls.n
    from hosts import timesformer as ts_host
    if meta["host"] == "timesformer" and ts_host._SHORT_KERNEL:  # the temporal attention of every space-time block
        assert calls.n["short_attention"] >= n_layers, calls.n
    if not meta.get("duplicate"):  # (a duplicate block only attends and merges: no LayerNorm behind that merge)
        fused = calls.n["merge_wavg_ln"] + calls.n["merge_wavg_regrouped"]
        assert fused == n_layers and calls.n["merge_wavg"] == 0, calls.n


def _check_layer0_and_logits(meta, z, plans32, plans16, out16):
    # layer 0: decisions whose fp64 margin exceeds the measured bf16 noise
    def cos(m):
        u = m / m.norm(dim=-1, keepdim=True)
        return u[:, ::2] @ u[:, 1::2].transpose(-1, -2)
    s32, s16 = cos(plans32[0][2]), cos(plans16[0][2])
    noise = float((s32 - s16).abs().max())
    r = plans32[0][1].r
    nm, _ = s32.max(-1)
    snm = nm.sort(dim=-1, descending=True).values
    set_ok = ((snm[:, r - 1] - snm[:, r]) > 2 * noise).cpu().numpy()
    want_src, want_dst = z["L0_src"].astype(np.int64), z["L0_dst"].astype(np.int64)
    got_src = plans16[0][1].src_idx.cpu().numpy()[..., 0]
    got_dst = plans16[0][1].dst_idx.cpu().numpy()[..., 0]
    checked_sets = checked_dst = 0
    top2 = s32.topk(2, dim=-1).values
    row_gap = (top2[..., 0] - top2[..., 1]).cpu().numpy()
    for g in range(want_src.shape[0]):
        if set_ok[g]:
            assert set(got_src[g].tolist()) == set(want_src[g].tolist()), f"group {g}: source set"
            checked_sets += 1
        got_map = dict(zip(got_src[g].tolist(), got_dst[g].tolist()))
        for i, j in zip(want_src[g].tolist(), want_dst[g].tolist()):
            if i in got_map and row_gap[g, i] > 2 * noise:
                assert got_map[i] == j, f"group {g}: destination of source row {i}"
                checked_dst += 1
    print(f"{meta['name']}: bf16 score noise {noise:.2e}; {checked_sets}/{want_src.shape[0]} source sets and "
          f"{checked_dst}/{want_src.size} destinations above it, all equal to the reference")
    assert checked_dst > 0, "no layer-0 decision lies above the bf16 noise: the fixture does not test this path"
    if meta.get("l0_set_gap"):
        # these fixtures' clips were chosen with a layer-0 r-boundary gap of >= 1e-2 in every group (the generator's
        # l0_set_gap): every group's source SET is then above the 16-bit noise and has been compared
        assert checked_sets == want_src.shape[0], (checked_sets, noise, meta["l0_set_gap"])
    ref = z["logits"]
    err = float(np.abs(out16.float().cpu().numpy() - ref).max())
    scale = float(np.abs(ref).max())
    tol = _logit_tol(meta["name"], "bf16_logit_err", scale, BF16_LOGIT_TOL)
    print(f"{meta['name']}: bf16 logits max |diff| {err:.3e} (largest |logit| {scale:.3e}, tolerance {tol:.3e})")
    _note_measured(meta["name"], bf16_logit_err=err)
    assert err <= tol, (err, tol, scale)


def test_videomae_schedule_and_modes():
    """r schedules (tuple form as the drivers set it), drop / hybrid / random modes and trace_source run end
    to end and walk the expected token counts (VideoMAE 1568 -> 1376 at r=16)."""
    tome, H = _hosts()
    torch.manual_seed(0)
    model = H["videomae"].VideoMAE(num_frames=16, img_size=224, patch_size=16, embed_dim=64, depth=12, num_heads=2,
                                   num_classes=7).to(DEV).eval()
    clip = [torch.rand(1, 3, 16, 224, 224, device=DEV)]
    for mode in ("merge", "drop", "hybrid", "random_merge", "random_drop"):
        tome.patch.videomae(model, mode=mode, threshold=0.5, trace_source=(mode == "merge"))
        model.r = (16, 0)
        with torch.no_grad():
            out = model(clip)
        assert out.shape == (1, 7) and torch.isfinite(out).all()
        size = model._tome_info["size"]
        assert size.shape[1] == 1376
        if mode in ("merge", "random_merge"):
            assert float(size.sum()) == 1568.0
        if mode == "merge":
            src = model._tome_info["source"]
            assert src.shape == (1, 1376, 1568) and float(src.sum()) == 1568.0
    tome.patch.videomae(model)
    model.r = (16, -1)  # decreasing schedule: 32, 29, ..., 2, 0
    with torch.no_grad():
        model(clip)
    assert model._tome_info["size"].shape[1] == 1568 - sum([32, 29, 26, 23, 20, 17, 14, 11, 8, 5, 2, 0])


@pytest.mark.parametrize("family", ["timesformer", "motionformer"])
def test_regrouped_models_drop_and_hybrid_modes(family):
    """TimeSformer / Motionformer in drop, random_drop and hybrid mode (kernel-addressed groups,
    tome_drop_regrouped / tome_merge_wavg_regrouped with edge flags) == the same model run through the
    reference's own sequence rearrange -> reduce -> rearrange -> cat built from this package's grouped calls."""
    tome, H = _hosts()
    from tome.patch import _common as C
    torch.manual_seed(0)
    if family == "timesformer":
        model = H["timesformer"].TimeSformer(num_frames=4, img_size=96, patch_size=16, embed_dim=64, depth=3, num_heads=2,
                                             num_classes=5).to(DEV).eval()
        clip = [torch.rand(2, 3, 4, 96, 96, device=DEV)]
        patch = tome.patch.timesformer
    else:
        model = H["motionformer"].Motionformer(img_size=96, patch_size=16, patch_size_temp=2, temporal_resolution=4,
                                               embed_dim=64, depth=3, num_heads=2, num_classes=5).to(DEV).eval()
        clip = [torch.rand(2, 3, 8, 96, 96, device=DEV)]
        patch = tome.patch.motionformer
    for mode in ("drop", "hybrid"):
        patch(model, mode=mode, threshold=0.9, prop_attn=True)
        model.r = (6, 0)
        with torch.no_grad():
            got = model(clip)
        sizes = model._tome_info["size"].clone()
        # the same forward with the regrouped entry points replaced by explicit regrouping around the grouped ones
        real_drop, real_merge = C.reduce_drop_regrouped, C.reduce_merge_regrouped

        def via_views(reduce_grouped):
            def f(metric, x_full, info, r, frames, **kw):
                B, N, Cc = x_full.shape
                P = (N - 1) // frames
                body = x_full[:, 1:].reshape(B, P, frames, Cc).permute(0, 2, 1, 3).reshape(B * frames, P, Cc)
                y = reduce_grouped(metric, body.contiguous(), info, r)
                P2 = y.shape[1]
                y = y.reshape(B, frames, P2, Cc).permute(0, 2, 1, 3).reshape(B, P2 * frames, Cc)
                return torch.cat((x_full[:, :1], y), dim=1)
            return f
        C.reduce_drop_regrouped = via_views(C.reduce_drop)
        C.reduce_merge_regrouped = via_views(C.reduce_hybrid)
        try:
            model.r = (6, 0)
            with torch.no_grad():
                want = model(clip)
        finally:
            C.reduce_drop_regrouped, C.reduce_merge_regrouped = real_drop, real_merge
        assert torch.equal(model._tome_info["size"], sizes)
        torch.testing.assert_close(got, want, rtol=0, atol=0)


def test_vivit_structural():
    """ViViT: class token protected and first, token counts 3137 -> 3137 - 64*layers at reduced depth/width."""
    tome, H = _hosts()
    torch.manual_seed(0)
    model = H["vivit"].ViViT(num_classes=5, image_size=224, num_frames=32, hidden_size=64, num_hidden_layers=3,
                             num_attention_heads=2, intermediate_size=128).to(DEV).eval()
    tome.patch.vivit(model)
    assert model._tome_info["class_token"] is True
    out, plans = _trace(tome, model, torch.rand(1, 3, 32, 224, 224, device=DEV), 64)
    assert out.shape == (1, 5) and torch.isfinite(out).all()
    assert [s[1] for s, _ in plans] == [3137, 3073, 3009]
    for _, p in plans:
        assert int(p.unm_idx[0, 0, 0]) == 0 and bool((p.src_idx != 0).all())
    assert float(model._tome_info["size"].sum()) == 3137.0


def test_timesformer_r32_walks_to_one_token():
    """TimeSformer r=32 on 196 spatial tokens: 196,164,132,100,68,36,18,9,5,3,2,1."""
    tome, H = _hosts()
    torch.manual_seed(0)
    model = H["timesformer"].TimeSformer(num_frames=2, img_size=224, patch_size=16, embed_dim=32, depth=12,
                                         num_heads=2, num_classes=3).to(DEV).eval()
    tome.patch.timesformer(model)
    out, plans = _trace(tome, model, torch.rand(1, 3, 2, 224, 224, device=DEV), 32)
    assert [s[1] for s, _ in plans] == [196, 164, 132, 100, 68, 36, 18, 9, 5, 3, 2]
    assert [p.r for _, p in plans] == [32, 32, 32, 32, 32, 18, 9, 4, 2, 1, 1]
    assert torch.isfinite(out).all()


def test_duplicate_layer_patches():
    tome, H = _hosts()
    torch.manual_seed(0)
    model = H["videomae"].VideoMAE(num_frames=4, img_size=32, patch_size=8, embed_dim=32, depth=3, num_heads=2,
                                   num_classes=3).to(DEV).eval()
    tome.patch.duplicate_videomae(model, 1, 3)
    assert len(model.model.blocks) == 5
    tome.patch.videomae(model)
    model.r = [2, 2, 2, 2, 2]
    with torch.no_grad():
        out = model([torch.rand(2, 3, 4, 32, 32, device=DEV)])
    assert torch.isfinite(out).all() and model._tome_info["size"].shape[1] == 32 - 10


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_vivit_duplicate_layer_patch(dtype):
    """tome.patch.duplicate_vivit (tome/patch/vivit.py:207-211): deep copies that only attend + merge are inserted in
    front of the duplicated layer, `config.num_hidden_layers` grows by `quantity` as in the reference, the per-layer r
    list walks the expected token counts, and the result equals running the layer's attention + merge by hand.  (No
    reference fixture: the reference's ViViT patch cannot be imported here, SURVEY.md 8c -- parity unpinned.)  The
    16-bit run also crosses a duplicated layer with the fused residual / LayerNorm hand-over switched on."""
    tome, H = _hosts()
    torch.manual_seed(0)
    model = H["vivit"].ViViT(num_classes=5, image_size=64, num_frames=8, hidden_size=128, num_hidden_layers=3,
                             num_attention_heads=2, intermediate_size=256).to(DEV).to(dtype).eval()
    before = model.vivit.config.num_hidden_layers
    tome.patch.duplicate_vivit(model, 1, 3)
    assert len(model.vivit.encoder.layer) == 5 and model.vivit.config.num_hidden_layers == before + 3
    tome.patch.vivit(model)
    clip = torch.rand(2, 3, 8, 64, 64, device=DEV).to(dtype)
    tokens = 1 + 4 * 16
    out, plans = _trace(tome, model, clip, [0, 6, 6, 6, 0])
    assert out.shape == (2, 5) and torch.isfinite(out).all()
    assert [s[1] for s, _ in plans] == [tokens, tokens - 6, tokens - 12]
    assert float(model._tome_info["size"].float().sum()) == 2.0 * tokens
    for _, p in plans:  # the class token is never merged and stays first
        assert int(p.unm_idx[0, 0, 0]) == 0 and bool((p.src_idx != 0).all())
    # r = 0 everywhere: the duplicates only attend (their output is dropped), the model equals the un-duplicated one
    torch.manual_seed(0)
    plain = H["vivit"].ViViT(num_classes=5, image_size=64, num_frames=8, hidden_size=128, num_hidden_layers=3,
                             num_attention_heads=2, intermediate_size=256).to(DEV).to(dtype).eval()
    tome.patch.vivit(plain)
    plain.r = 0
    model.r = 0
    with torch.no_grad():
        want, got = plain([clip]), model([clip])
    assert torch.equal(want, got)


def test_vivit_fused_qkv_projection_follows_its_weights(monkeypatch):
    """The ViViT patch runs query / key / value (tome/patch/vivit.py:95-101 of the reference: three Linear calls) as one
    projection over a cached side-by-side copy of the three weights.  The cached copy must follow the weights (in-place
    update, replaced parameter), hooked or grad-enabled modules must run themselves, and the result must match the three
    separate GEMMs within the rounding of a GEMM of another shape."""
    tome, H = _hosts()
    import sys
    vv = sys.modules["tome.patch.vivit"]
    torch.manual_seed(0)
    model = H["vivit"].ViViT(num_classes=5, image_size=64, num_frames=8, hidden_size=128, num_hidden_layers=2,
                             num_attention_heads=2, intermediate_size=256).to(DEV).to(torch.bfloat16).eval()
    tome.patch.vivit(model)
    model.r = 6
    clip = [torch.rand(2, 3, 8, 64, 64, device=DEV).to(torch.bfloat16)]
    att = model.vivit.encoder.layer[0].attention.attention

    def run(fused):
        monkeypatch.setattr(vv, "_FUSE_QKV", fused)
        with torch.no_grad():
            return model(clip).float()

    a, b = run(True), run(False)
    assert "_tome_qkv" in att.__dict__ and "_tome_qkv" not in model.state_dict()
    assert float((a - b).abs().max()) <= 0.02 * max(1.0, float(b.abs().max()))
    with torch.no_grad():
        att.query.weight.mul_(-1.0)  # in place: the version counter moves, the cached copy must be rebuilt
    a2, b2 = run(True), run(False)
    assert float((a2 - b2).abs().max()) <= 0.02 * max(1.0, float(b2.abs().max()))
    assert float((a2 - a).abs().max()) > 0.0
    att.key.weight = torch.nn.Parameter(att.key.weight.detach().clone() * 0.5, requires_grad=False)  # replaced
    a3, b3 = run(True), run(False)
    assert float((a3 - b3).abs().max()) <= 0.02 * max(1.0, float(b3.abs().max()))
    # a hooked projection runs itself
    seen = []
    hk = att.value.register_forward_hook(lambda m, i, o: seen.append(1))
    run(True)
    hk.remove()
    assert len(seen) == 1


def test_graphed_forward_replays_the_merge_path():
    """The whole patched forward, merge kernels included, captured in a HIP graph and replayed on new clips:
    same logits as the eager run (the merge path launches on torch's current stream, allocates through
    torch and never synchronises, so it is capturable)."""
    tome, H = _hosts()
    from hosts.graphed import GraphedForward
    torch.manual_seed(0)
    model = H["videomae"].VideoMAE(num_frames=8, img_size=64, patch_size=16, embed_dim=64, depth=4, num_heads=1,
                                   num_classes=9).to(DEV).eval()
    tome.patch.videomae(model)
    model.r = 6
    a = torch.rand(2, 3, 8, 64, 64, device=DEV)
    b = torch.rand(2, 3, 8, 64, 64, device=DEV)
    fwd = GraphedForward(model, [a])
    with torch.no_grad():
        want_b = model([b]).clone()
        want_a = model([a]).clone()
    got_b = fwd([b]).clone()
    got_a = fwd([a]).clone()
    assert torch.equal(got_b, want_b) and torch.equal(got_a, want_a)
    assert not torch.equal(got_a, got_b)


def _partition_of(source: torch.Tensor) -> np.ndarray:
    """tests/golden/generate_models.py canonical_partition on the device: [n, T, T0] 0/1 -> [n, T0] smallest member of
    every original token's merged group (free of the order of the merged rows)."""
    n, T, T0 = source.shape
    ids = torch.arange(T0, device=source.device).view(1, 1, T0).expand(n, T, T0)
    first = torch.where(source > 0.5, ids, torch.full_like(ids, T0)).min(-1).values
    owner = (source > 0.5).float().transpose(1, 2) @ first.float().unsqueeze(-1)
    return owner[..., 0].round().cpu().numpy().astype(np.int16)


def test_config0_videomae_b_full_size_against_the_reference(monkeypatch):
    """BASELINE.json configs[0] AT FULL SIZE against the REAL reference (tests/golden/generate_models.py emit_config0:
    tome/patch/videomae.py over slowfast's VideoMAE-B, 2 clips of 16x224x224, fill_parameters weights, fp32, r = 8,
    CPU, trace_source on).  The fixture's clips were chosen so that in all 12 layers the r-boundary and the selected
    rows' top-2 gaps exceed 2e-5 in fp64 (they are 6e-5 .. 2.6e-4), two orders above what fp32 evaluation moves such
    cosines: given its INPUT ORDER, every merge of a layer is defined.  What stays open is the ORDER in which a layer
    emits its ~770 unmerged rows per clip (sorted by row maxima that lie 1e-8 .. 1e-5 apart: the reference's unstable
    argsort over fp32 values that differ between any two BLAS), and that order decides which tokens are even (sources)
    and odd (destinations) in the NEXT layer -- so from the second or third layer on the reference itself would merge
    other tokens on another machine.  Hence:
      * tokens per layer and r_eff exactly; layer 0's src_idx / dst_idx exactly, unm_idx exactly at the positions the
        fixture certifies (1049 of 1552; the rest as a set); the partition of the 1568 original tokens into merged
        groups after layer 0 exactly (trace_source through tome_source_init / merge(., "max"), compared in the
        order-free form of generate_models.canonical_partition);
      * later layers: the share of original tokens whose merged group equals the reference's is printed per layer
        (measured: identical through layer 1, then 99.7 % falling to 91.9 % after layer 11) and must stay >= 90 %;
        sizes sum to 1568 per clip;
      * logits within 0.5 % of the largest |logit| (4.8): measured 8.4e-3 = 0.18 % -- fp32 GEMMs and attention on
        different BLAS / summation orders through 12 layers, plus the legitimately different late merges (92 % of the
        tokens end in the reference's group; the others were merged with a near-identical neighbour instead).
    The same clips in bf16 through the production kernels: token trace exact, logits within BF16_LOGIT_TOL of the
    reference's largest logit."""
    meta = G.manifest()["config0"]
    assert meta["certified"]
    z = np.load(os.path.join(G.GOLDEN, "models_config0_videomae_b.npz"))
    tome, H = _hosts()
    cfg = dict(meta["cfg"])
    model = H["videomae"].VideoMAE(num_frames=cfg.pop("all_frames"), num_classes=cfg.pop("num_classes"),
                                   tubelet_size=cfg.pop("tubelet_size"), **cfg)
    names = synth.fill_parameters(model, meta["weight_seed"])
    assert set(names) == set(meta["param_names"])
    model = model.to(DEV).eval()
    tome.patch.videomae(model, prop_attn=False, trace_source=True)
    clip = _clip(meta)
    # the source matrix after every layer
    from tome.patch import _common
    parts = []
    real_ms = _common.merge_source

    def spy_source(merge, x, source=None):
        out = real_ms(merge, x, source)
        parts.append(_partition_of(out))
        return out
    monkeypatch.setattr(_common, "merge_source", spy_source)
    out, plans = _trace(tome, model, clip, meta["r"])
    assert [s[1] for s, _ in plans] == meta["tokens"] and [p.r for _, p in plans] == meta["r_eff"]
    p0 = plans[0][1]
    np.testing.assert_array_equal(p0.src_idx.cpu().numpy()[..., 0], z["L0_src"])
    np.testing.assert_array_equal(p0.dst_idx.cpu().numpy()[..., 0], z["L0_dst"])
    got_unm, want_unm, ok = p0.unm_idx.cpu().numpy()[..., 0], z["L0_unm"].astype(np.int64), z["L0_unm_certified"]
    np.testing.assert_array_equal(got_unm[ok], want_unm[ok])
    np.testing.assert_array_equal(np.sort(got_unm, 1), np.sort(want_unm, 1))
    assert len(parts) == len(meta["tokens"])
    np.testing.assert_array_equal(parts[0], z["partitions"][0], err_msg="merged groups after layer 0")
    agree = [float((got == z["partitions"][layer]).mean()) for layer, got in enumerate(parts)]
    exact = next((i for i, a in enumerate(agree) if a < 1.0), len(agree))
    print(f"config0 full size fp32: merged groups identical to the reference's through layer {exact - 1}; share of "
          f"original tokens with the same group per layer: {[round(a, 4) for a in agree]}")
    assert min(agree) >= 0.9, agree
    sizes = model._tome_info["size"].cpu().numpy()[..., 0]
    assert sizes.shape == z["size"].shape[:2] and (sizes.sum(1) == 1568.0).all() and (z["size"].sum((1, 2)) == 1568.0).all()
    err = float(np.abs(out.cpu().numpy() - z["logits"]).max())
    print(f"config0 full size fp32: logits max |diff| {err:.3e} (largest |logit| {np.abs(z['logits']).max():.3e})")
    assert err <= 5e-3 * float(np.abs(z["logits"]).max()), err
    # ---- the same clips as the benchmark runs them: bf16, every fused kernel on
    monkeypatch.setattr(_common, "merge_source", real_ms)
    model16 = H["videomae"].VideoMAE(num_frames=16, num_classes=400, tubelet_size=2, **cfg)
    synth.fill_parameters(model16, meta["weight_seed"])
    model16 = model16.to(DEV).to(torch.bfloat16).eval()
    tome.patch.videomae(model16, prop_attn=False)
    out16, plans16 = _trace(tome, model16, clip.bfloat16(), meta["r"])
    assert [s[1] for s, _ in plans16] == meta["tokens"]
    err16 = float(np.abs(out16.float().cpu().numpy() - z["logits"]).max())
    tol16 = _logit_tol("config0_videomae_b", "bf16_logit_err", float(np.abs(z["logits"]).max()), BF16_LOGIT_TOL)
    print(f"config0 full size bf16: logits max |diff| {err16:.3e} (tolerance {tol16:.3e})")
    _note_measured("config0_videomae_b", bf16_logit_err=err16, fp32_logit_err=err, agree=agree)
    assert err16 <= tol16, (err16, tol16)


def _full_metas():
    return G.manifest().get("full", [])


@pytest.mark.parametrize("meta", _full_metas(), ids=lambda m: m["name"])
def test_full_size_config_against_the_reference(meta, monkeypatch):
    """BASELINE.json configs[1] (the metric's own config: VideoMAE-B 16x224, r = 16), configs[2] (TimeSformer divST
    8x224, r = 8 / 16 / 32 incl. the 196 -> ... -> 1 walk) and configs[4] (Motionformer 224 16x4, r = 16) AT FULL SIZE
    against the REAL reference (tests/golden/generate_models.py emit_full: the reference's tome/patch/<family>.py over
    its own slowfast model class, embed 768 / depth 12 / 12 heads, 2 synth.uniform01 clips, fill_parameters weights,
    fp32, CPU, trace_source=True, the family's default prop_attn).  Random-init keys are nearly parallel, so decision
    margins are small (1e-7 .. 1e-4); the fixture carries per-position fp64 certificates (tau = 2e-5, two orders above
    what fp32 evaluation moves a cosine) and the test holds exactly what they define:
      * tokens, r_eff and groups of every layer exactly;
      * layer 0: the source SET of every group whose r-boundary is certified; src_idx at every certified position;
        the destination of every source whose top-2 gap is certified; unm_idx at every certified position (and as a set
        where the boundary is certified); the merged-group partition after layer 0 exactly for every group that is
        fully certified;
      * later layers (the order in which a layer emits its unmerged rows -- row maxima 1e-8 .. 1e-5 apart under the
        reference's unstable argsort -- decides which tokens are even / odd in the next layer, so the reference itself
        would merge other tokens on another machine): the share of original tokens whose merged group equals the
        reference's, per layer, printed, and held to the share a run on MI355X measured minus 2 points
        (tests/golden/measured.json; >= 0.5 while none is stored); sizes sum to the tokens of a group;
      * fp32 logits within 1.5x the measured error; the same clips in bf16 through the production kernels (asserted by
        call count): token trace exact, logits within 1.5x the measured bf16 error."""
    z = np.load(os.path.join(G.GOLDEN, f"models_{meta['name']}.npz"))
    tome, model, patch = _build(meta)
    patch(model, prop_attn=meta["prop_attn"], trace_source=True)
    clip = _clip(meta)
    from tome.patch import _common
    parts = []
    real_ms = _common.merge_source

    def spy_source(merge, x, source=None):
        out = real_ms(merge, x, source)
        if hasattr(merge, "plan"):
            parts.append(_partition_of(out))
        return out
    monkeypatch.setattr(_common, "merge_source", spy_source)
    out, plans = _trace(tome, model, clip, meta["r"])
    assert [s[1] for s, _ in plans] == meta["tokens"] and [p.r for _, p in plans] == meta["r_eff"]
    assert [s[0] for s, _ in plans] == meta["groups"]
    # ---- layer 0 against the certificates
    p0 = plans[0][1]
    got_src, got_dst, got_unm = (t.cpu().numpy()[..., 0] for t in (p0.src_idx, p0.dst_idx, p0.unm_idx))
    want_src, want_dst, want_unm = (z[k].astype(np.int64) for k in ("L0_src", "L0_dst", "L0_unm"))
    set_ok, src_ok, dst_ok, unm_ok = z["L0_set_ok"], z["L0_src_ok"], z["L0_dst_ok"], z["L0_unm_ok"]
    np.testing.assert_array_equal(got_src[src_ok], want_src[src_ok], err_msg="layer 0 src_idx at certified positions")
    np.testing.assert_array_equal(got_unm[unm_ok], want_unm[unm_ok], err_msg="layer 0 unm_idx at certified positions")
    n_dst = 0
    for g in range(want_src.shape[0]):
        if set_ok[g]:
            assert set(got_src[g].tolist()) == set(want_src[g].tolist()), f"group {g}: source set"
            assert set(got_unm[g].tolist()) == set(want_unm[g].tolist()), f"group {g}: unmerged set"
        got_map = dict(zip(got_src[g].tolist(), got_dst[g].tolist()))
        for k, (i, j) in enumerate(zip(want_src[g].tolist(), want_dst[g].tolist())):
            if dst_ok[g, k] and i in got_map:
                assert got_map[i] == j, f"group {g}: destination of source row {i}"
                n_dst += 1
    assert n_dst >= int(dst_ok[set_ok].sum()) and n_dst > 0
    group_ok = z["L0_group_ok"]
    assert len(parts) == len(meta["tokens"])
    np.testing.assert_array_equal(parts[0][group_ok], z["P0"][group_ok], err_msg="merged groups after layer 0")
    agree = [float((got == z[f"P{layer}"]).mean()) for layer, got in enumerate(parts)]
    exact = next((i for i, a in enumerate(agree) if a < 1.0), len(agree))
    print(f"{meta['name']} fp32: layer 0 -- {int(set_ok.sum())}/{set_ok.size} source sets, {int(src_ok.sum())}/"
          f"{src_ok.size} src positions, {n_dst}/{dst_ok.size} destinations, {int(unm_ok.sum())}/{unm_ok.size} unm "
          f"positions, {int(group_ok.sum())}/{group_ok.size} group partitions certified, all equal to the reference; "
          f"merged groups identical through layer {exact - 1}; share of original tokens in the reference's group per "
          f"layer: {[round(a, 4) for a in agree]}")
    known = _measured().get(meta["name"], {})
    floor = [max(0.0, a - 0.02) for a in known["agree"]] if "agree" in known else [0.5] * len(agree)
    assert all(a >= f for a, f in zip(agree, floor)), (agree, floor)
    sizes = model._tome_info["size"].cpu().numpy()[..., 0]
    t0 = meta["tokens"][0]
    assert sizes.shape == z["size"].shape[:2] and (sizes.sum(1) == float(t0)).all()
    assert (z["size"].sum((1, 2)) == float(t0)).all()
    scale = float(np.abs(z["logits"]).max())
    err = float(np.abs(out.cpu().numpy() - z["logits"]).max())
    tol = _logit_tol(meta["name"], "fp32_logit_err", scale, 0.02)
    print(f"{meta['name']} fp32: logits max |diff| {err:.3e} (largest |logit| {scale:.3e}, tolerance {tol:.3e})")
    assert err <= tol, (err, tol)
    # ---- the same clips as the benchmark runs them: bf16, every fused kernel on
    monkeypatch.setattr(_common, "merge_source", real_ms)
    del model
    tome, model16, patch = _build(meta)
    model16 = model16.to(torch.bfloat16)
    patch(model16, prop_attn=meta["prop_attn"])
    calls = _CallCounter(monkeypatch)
    out16, plans16 = _trace(tome, model16, clip.bfloat16(), meta["r"])
    assert [s[1] for s, _ in plans16] == meta["tokens"] and [p.r for _, p in plans16] == meta["r_eff"]
    if all((_common._FUSE_LN, _common._FUSE_ADD, _common._FUSE_NEXT, _common._ATTN_KERNEL)):
        from tome.utils import parse_r
        asked = sum(1 for r in parse_r(meta["cfg"]["depth"], meta["r"]) if r > 0)
        _assert_production_calls(calls, meta, len(plans16), n_match=asked)
    err16 = float(np.abs(out16.float().cpu().numpy() - z["logits"]).max())
    tol16 = _logit_tol(meta["name"], "bf16_logit_err", scale, BF16_LOGIT_TOL)
    print(f"{meta['name']} bf16: logits max |diff| {err16:.3e} (tolerance {tol16:.3e})")
    _note_measured(meta["name"], fp32_logit_err=err, bf16_logit_err=err16, agree=agree)
    assert err16 <= tol16, (err16, tol16)


@pytest.mark.parametrize("meta", G.manifest().get("modes", []), ids=lambda m: m["name"])
def test_patch_level_drop_and_hybrid_against_the_reference(meta):
    """The patch-level glue of the DROP and HYBRID modes against the reference's own patched models
    (tome/patch/videomae.py:102-151, timesformer.py:111-185, motionformer.py:172-245; fixtures:
    generate_models.py emit_mode, reduced width, head dim 64, trace_source=True, every layer certified): per layer
    token counts, r_eff and the index tensors exactly; the sizes exactly -- drop resets them to ones of **fp32**
    whatever the tokens' dtype (videomae.py:123), hybrid sums what the kept destinations and the sources carry; the
    final source matrix exactly (drop: `drop(eye)`, a dropped token has no column entry; hybrid: a destination with an
    incoming edge below the threshold loses its own column, merge.py:326-331); logits within 2e-4.  Hybrid fixtures
    carry a threshold INSIDE the selected edges' scores (some destinations kept, some zeroed in every fixture:
    `edges_kept` of `edges` in the manifest), plumbed through `_tome_info["threshold"]`, and the per-edge keep flags
    of every layer are compared too."""
    assert meta["certified"]
    z = np.load(os.path.join(G.GOLDEN, f"models_{meta['name']}.npz"))
    tome, model, patch = _build(meta)
    kw = dict(prop_attn=meta["prop_attn"], mode=meta["mode"], trace_source=True)
    if meta["mode"] == "hybrid":
        kw["threshold"] = meta["threshold"]
    patch(model, **kw)
    info = model._tome_info
    assert info["mode"] == meta["mode"] and info["trace_source"] is True
    out, plans = _trace(tome, model, _clip(meta), meta["r"])
    assert [s[1] for s, _ in plans] == meta["tokens"] and [p.r for _, p in plans] == meta["r_eff"]
    assert [s[0] for s, _ in plans] == meta["groups"]
    for i, (_, p) in enumerate(plans):
        np.testing.assert_array_equal(p.src_idx.cpu().numpy()[..., 0], z[f"L{i}_src"], err_msg=f"layer {i} src")
        np.testing.assert_array_equal(p.unm_idx.cpu().numpy()[..., 0], z[f"L{i}_unm"], err_msg=f"layer {i} unm")
        if meta["mode"] == "hybrid":
            np.testing.assert_array_equal(p.dst_idx.cpu().numpy()[..., 0], z[f"L{i}_dst"], err_msg=f"layer {i} dst")
            keep = p.edge_keep.cpu().numpy().astype(bool).reshape(z[f"L{i}_keep"].shape)
            np.testing.assert_array_equal(keep, z[f"L{i}_keep"], err_msg=f"layer {i} edge keep flags")
            assert 0 < int(z["L0_keep"].sum()) < z["L0_keep"].size
    size = info["size"]
    assert str(size.dtype) == meta["size_dtype"]
    np.testing.assert_array_equal(size.cpu().numpy(), z["size"])
    np.testing.assert_array_equal(info["source"].cpu().numpy().astype(np.uint8), z["source"])
    np.testing.assert_allclose(out.cpu().numpy(), z["logits"], atol=2e-4, rtol=1e-4)


def test_config0_videomae_b_fp32_vs_cpu_port():
    """BASELINE.json configs[0]: VideoMAE-B random-init, 2 clips of 16x224x224, r=8, fp32 -- the reference's own
    CPU-runnable case.  The CPU side is oracle/torch_port.py (the reference's op sequence, pinned to the golden
    vectors); the GPU side is the patched host model on the HIP merge path with the same weights.  Token
    schedule identical (1568 -> 1472); first-layer src/dst indices identical (their margins are far above GEMM
    noise; the order of the unmerged rows is not defined for near-ties, SURVEY 7.1, so later layers may take
    different but equivalent paths); logits within 2e-3 of a logit scale of ~1e-1."""
    tome, H = _hosts()
    from oracle import torch_port
    torch.manual_seed(0)
    cpu_model = H["videomae"].videomae_base(16).eval()
    clip = torch.rand(2, 3, 16, 224, 224)
    trace = []
    want = torch_port.videomae_forward(cpu_model, clip, 8, trace=trace)
    assert [t for t, _ in trace] == [1568 - 8 * i for i in range(12)]
    import copy
    gpu_model = copy.deepcopy(cpu_model).to(DEV).eval()
    tome.patch.videomae(gpu_model)  # prop_attn False, as every VideoMAE run of the reference
    got, plans = _trace(tome, gpu_model, clip.to(DEV), 8)
    assert [s[1] for s, _ in plans] == [t for t, _ in trace]
    p0, ref0 = plans[0][1], trace[0][1]
    assert torch.equal(p0.src_idx.cpu().sort(1).values, ref0.src_idx.sort(1).values)
    order = p0.src_idx.cpu()[..., 0].argsort(1)
    rorder = ref0.src_idx[..., 0].argsort(1)
    assert torch.equal(p0.dst_idx.cpu()[..., 0].gather(1, order), ref0.dst_idx[..., 0].gather(1, rorder))
    assert float(gpu_model._tome_info["size"].sum()) == 2 * 1568.0
    err = (got.cpu() - want).abs().max().item()
    scale = want.abs().max().item()
    assert err <= 2e-3 + 2e-2 * scale, (err, scale)


@pytest.mark.parametrize("host_name", ["videomae", "vivit", "timesformer", "motionformer"])
def test_fused_block_equals_unfused_block_bf16(host_name, monkeypatch):
    """bf16 forward with the residual add + merge + LayerNorm fused into one kernel (tome_merge_wavg_ln) vs the
    same forward running the three steps separately: identical token schedule and first-layer indices, final
    sizes identical, logits within bf16 rounding noise."""
    tome, H = _hosts()
    from tome.patch import _common
    torch.manual_seed(0)
    if host_name == "videomae":
        model = H["videomae"].VideoMAE(num_frames=8, img_size=64, patch_size=16, embed_dim=64, depth=4, num_heads=1,
                                       num_classes=9)
        patch, frames = tome.patch.videomae, 8
    elif host_name == "vivit":
        model = H["vivit"].ViViT(num_classes=9, image_size=64, num_frames=8, hidden_size=64, num_hidden_layers=4,
                                 num_attention_heads=1, intermediate_size=128)
        patch, frames = tome.patch.vivit, 8
    elif host_name == "timesformer":
        model = H["timesformer"].TimeSformer(num_frames=4, img_size=64, patch_size=8, embed_dim=64, depth=4,
                                             num_heads=1, num_classes=9)
        patch, frames = tome.patch.timesformer, 4
    else:
        model = H["motionformer"].Motionformer(img_size=64, patch_size=8, patch_size_temp=2, temporal_resolution=4,
                                               embed_dim=64, depth=3, num_heads=1, num_classes=9)
        patch, frames = tome.patch.motionformer, 8
    model = model.to(DEV).to(torch.bfloat16).eval()
    patch(model)
    clip = torch.rand(3, 3, frames, 64, 64, device=DEV).to(torch.bfloat16)
    outs = {}
    for fused in (True, False):
        monkeypatch.setattr(_common, "_FUSE_LN", fused)
        monkeypatch.setattr(_common, "_FUSE_ADD", fused)
        monkeypatch.setattr(_common, "_FUSE_NEXT", fused)
        out, plans = _trace(tome, model, clip, 5)
        outs[fused] = (out.float(), plans, model._tome_info["size"].float().clone())
    (o1, p1, s1), (o0, p0, s0) = outs[True], outs[False]
    assert [s[1] for s, _ in p1] == [s[1] for s, _ in p0]
    assert torch.equal(p1[0][1].src_idx, p0[0][1].src_idx) and torch.equal(p1[0][1].dst_idx, p0[0][1].dst_idx)
    assert float(s1.sum()) == float(s0.sum())
    assert float((o1 - o0).abs().max()) <= 0.05 * max(1.0, float(o0.abs().max()))


@pytest.mark.parametrize("host_name", ["videomae", "vivit"])
def test_matching_beside_the_attention_on_the_side_stream(host_name, monkeypatch):
    """tome/_overlap.py: a patched attention issues its layer's matching on a second HIP stream behind its qkv GEMM, and
    the merge waits for it.  Same kernels on the same keys, so the forward is BIT-identical to the one that keeps the
    matching on the caller's stream -- also when the side stream is held up by a spin kernel in front of every
    matching (a merge that did not wait would read index tensors nobody has written yet) -- every merging layer's
    matching did run on the side stream, nothing is left un-joined, and a HIP graph captures the fork and the join.
    (VideoMAE and ViViT: long sequences, streaming attention kernel.  TimeSformer / Motionformer keep the matching on
    the caller's stream: measured +-0 at large batches and -15..-20 % at batch 8, where they are bound by the host.)"""
    tome, H = _hosts()
    from tome import _abi, _overlap
    from hosts.graphed import GraphedForward
    torch.manual_seed(0)
    if host_name == "videomae":
        model = H["videomae"].VideoMAE(num_frames=8, img_size=64, patch_size=16, embed_dim=128, depth=4, num_heads=2,
                                       num_classes=9)
        patch, frames, layers = tome.patch.videomae, 8, 4
    else:
        model = H["vivit"].ViViT(num_classes=9, image_size=64, num_frames=8, hidden_size=128, num_hidden_layers=4,
                                 num_attention_heads=2, intermediate_size=128)
        patch, frames, layers = tome.patch.vivit, 8, 4
    model = model.to(DEV).to(torch.bfloat16).eval()
    patch(model)
    model.r = 5
    clip = torch.rand(3, 3, frames, 64, 64, device=DEV).to(torch.bfloat16)
    streams = []
    real = _abi.match_keys

    def spy(*a, **k):
        streams.append(torch.cuda.current_stream().cuda_stream)  # launches AND allocations of the matching go there
        return real(*a, **k)

    main = torch.cuda.current_stream().cuda_stream
    monkeypatch.setattr(_abi, "match_keys", spy)
    with torch.no_grad():
        monkeypatch.setattr(_overlap, "ENABLED", False)
        want = model([clip]).clone()
        assert streams and set(streams) == {main}
        n_match = len(streams)
        assert n_match == layers
        streams.clear()
        monkeypatch.setattr(_overlap, "ENABLED", True)
        monkeypatch.setattr(_overlap, "MIN_WORK", int(6e7))  # (the default, whatever TOME_MATCH_STREAM_MIN says)
        model([clip])
        assert set(streams) == {main} and len(streams) == n_match  # too small a forward to fork in eager mode ...
        streams.clear()
        monkeypatch.setattr(_overlap, "MIN_WORK", 0)
        got = model([clip]).clone()
        side = _overlap.side_stream(torch.device(DEV)).cuda_stream
        assert side != main and streams == [side] * n_match
        assert not _overlap._open
        assert torch.equal(got, want)
        # the side stream held up in front of every matching
        real_beside = _overlap.match_beside

        def late(metric, ready, info):
            if ready is not None:
                with torch.cuda.stream(_overlap.side_stream(metric.keys.device)):
                    torch.cuda._sleep(20_000_000)  # ~10 ms
            return real_beside(metric, ready, info)

        from tome.patch import _common
        monkeypatch.setattr(_common, "match_beside", late)
        streams.clear()
        got_late = model([clip]).clone()
        assert streams == [side] * n_match and torch.equal(got_late, want)
        monkeypatch.setattr(_common, "match_beside", real_beside)
        # a block that raises between the fork and the join: the model forward joins on its way out
        def boom(*a, **k):
            raise RuntimeError("boom")
        monkeypatch.setattr(_common, "bipartite_soft_matching", boom)
        seen_open = []
        real_join = _overlap.join
        monkeypatch.setattr(_overlap, "join", lambda dev: (seen_open.append(bool(_overlap._open)), real_join(dev))[1])
        with pytest.raises(RuntimeError, match="boom"):
            model([clip])
        assert seen_open == [True] and not _overlap._open
        monkeypatch.undo()
        torch.cuda.synchronize()
    # captured: the side stream joins the capture through the two events (... but always inside a capture)
    monkeypatch.setattr(_overlap, "ENABLED", True)  # (monkeypatch.undo() above restored the environment's settings)
    monkeypatch.setattr(_overlap, "MIN_WORK", int(6e7))
    main = torch.cuda.current_stream().cuda_stream
    seen = []
    real = _abi.match_keys
    monkeypatch.setattr(_abi, "match_keys",
                        lambda *a, **k: (seen.append(torch.cuda.current_stream().cuda_stream), real(*a, **k))[1])
    fwd = GraphedForward(model, [clip], warmup=1)
    side = _overlap.side_stream(torch.device(DEV)).cuda_stream
    assert side not in seen[:layers] and seen[layers:] == [side] * layers
    other = torch.rand(3, 3, frames, 64, 64, device=DEV).to(torch.bfloat16)
    with torch.no_grad():
        want_other = model([other]).clone()
    assert torch.equal(fwd([other]).clone(), want_other) and torch.equal(fwd([clip]).clone(), want)


@pytest.mark.parametrize("host_name,batch,r", [("videomae", 16, 16), ("timesformer", 64, 16), ("motionformer", 16, 16),
                                               ("vivit", 4, 64)])
def test_forward_reads_no_memory_it_did_not_write(host_name, batch, r):
    """Every kernel of this package takes its buffers from `torch.empty`: nothing may be read before it is written.
    The caching allocator's pool is pre-filled with a byte pattern (one big tensor filled and released: every later
    allocation of the forward is carved out of it); the full-size bf16 forward -- filter path of the matching at
    TimeSformer's 512 groups, the side-stream fork where it applies -- must not depend on the pattern, plans included."""
    tome, H = _hosts()
    from tome import _abi
    torch.manual_seed(0)
    build, patch, frames, kw = {
        "videomae": (lambda: H["videomae"].videomae_base(16), tome.patch.videomae, 16, {"prop_attn": False}),
        "timesformer": (lambda: H["timesformer"].timesformer_base(8), tome.patch.timesformer, 8, {}),
        "motionformer": (lambda: H["motionformer"].motionformer_base(), tome.patch.motionformer, 16, {}),
        "vivit": (lambda: H["vivit"].vivit_base(32), tome.patch.vivit, 32, {}),
    }[host_name]
    model = build().to(DEV).to(torch.bfloat16).eval()
    patch(model, **kw)
    model.r = r
    clip = [torch.rand(batch, 3, frames, 224, 224, device=DEV).to(torch.bfloat16)]
    torch.cuda.reset_peak_memory_stats()
    with torch.no_grad():
        model(clip)
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated()
    runs = []
    for fill in (0, 255, "rand"):
        torch.cuda.empty_cache()
        junk = torch.empty(int(peak * 1.2), dtype=torch.uint8, device=DEV)
        junk.random_(0, 256) if fill == "rand" else junk.fill_(fill)
        torch.cuda.synchronize()
        del junk
        with torch.no_grad():
            out, plans = _trace(tome, model, clip[0], r)
        runs.append((out.clone(), [(p.src_idx.clone(), p.dst_idx.clone(), p.unm_idx.clone()) for _, p in plans]))
    torch.cuda.synchronize()
    for out, plans in runs[1:]:
        assert not out.isnan().any()
        assert all(torch.equal(a, b) for pa, pb in zip(plans, runs[0][1]) for a, b in zip(pa, pb))
        assert torch.equal(out, runs[0][0])
    del model, clip, runs
    torch.cuda.empty_cache()


@pytest.mark.parametrize("host_name,batch", [("timesformer", 64), ("videomae", 32)])
def test_side_stream_matching_shares_no_memory_with_kernels_in_flight(host_name, batch, monkeypatch):
    """The matching beside the attention must take its plan and scratch from the SIDE stream's allocator pool.  Taken
    from the main stream's pool (round 4's first lean form: launches by raw stream handle) they can be blocks a kernel
    still in flight on the main stream reads -- the attention wrapper's log(size) temporary, released a microsecond
    before -- and the matching writes into them beside that reader.  Seen on exactly this case: full-size TimeSformer,
    64 clips (512 groups: the filter path), prop_attn, the fork forced in eager mode -- the attention output of block
    4 changed.  The forward with the fork must equal the forward without it bit for bit, twice.  (Second case:
    full-size VideoMAE with the size bias, at a batch that forks by itself in eager mode.)"""
    tome, H = _hosts()
    from tome import _overlap
    from tome.patch import _common
    torch.manual_seed(0)
    if host_name == "timesformer":
        model = H["timesformer"].timesformer_base(8).to(DEV).to(torch.bfloat16).eval()
        tome.patch.timesformer(model)
        frames = 8
    else:
        model = H["videomae"].videomae_base(16).to(DEV).to(torch.bfloat16).eval()
        tome.patch.videomae(model, prop_attn=True)
        frames = 16
    model.r = 16
    clip = [torch.rand(batch, 3, frames, 224, 224, device=DEV).to(torch.bfloat16)]
    real_ready = _overlap.keys_ready
    forks = []

    def ready(keys, info, capture_only=False):
        ev = real_ready(keys, info, False)  # fork in eager mode too
        forks.append(ev is not None)
        return ev

    monkeypatch.setattr(_common, "keys_ready", ready)
    monkeypatch.setattr(_overlap, "MIN_WORK", 0)
    with torch.no_grad():
        monkeypatch.setattr(_overlap, "ENABLED", False)
        want = model(clip).clone()
        assert not any(forks)
        monkeypatch.setattr(_overlap, "ENABLED", True)
        for _ in range(2):
            forks.clear()
            got = model(clip).clone()
            assert all(forks) and len(forks) == 12
            assert torch.equal(got, want), float((got.float() - want.float()).abs().max())
    torch.cuda.synchronize()


@pytest.mark.parametrize("host_name", ["videomae", "vivit", "timesformer", "motionformer"])
def test_attention_kernel_equals_framework_attention_bf16(host_name, monkeypatch):
    """Proportional attention through tome_prop_attention (size bias per key inside the kernel; Motionformer: one
    launch per frame of the trajectory attention's first stage) vs the framework's attention with the bias tensor
    the reference builds: same token schedule, same first-layer matching, logits within bf16 noise."""
    tome, H = _hosts()
    from tome.patch import _common
    torch.manual_seed(0)
    if host_name == "videomae":
        model = H["videomae"].VideoMAE(num_frames=8, img_size=64, patch_size=8, embed_dim=128, depth=3, num_heads=2,
                                       num_classes=9)
        patch, frames = tome.patch.videomae, 8
    elif host_name == "vivit":
        model = H["vivit"].ViViT(num_classes=9, image_size=64, num_frames=8, hidden_size=128, num_hidden_layers=3,
                                 num_attention_heads=2, intermediate_size=256, tubelet_size=(2, 8, 8))
        patch, frames = tome.patch.vivit, 8
    elif host_name == "timesformer":
        model = H["timesformer"].TimeSformer(num_frames=4, img_size=64, patch_size=8, embed_dim=128, depth=3,
                                             num_heads=2, num_classes=9)
        patch, frames = tome.patch.timesformer, 4
    else:
        model = H["motionformer"].Motionformer(img_size=64, patch_size=8, patch_size_temp=2, temporal_resolution=4,
                                               embed_dim=128, depth=3, num_heads=2, num_classes=9)
        patch, frames = tome.patch.motionformer, 8
    model = model.to(DEV).to(torch.bfloat16).eval()
    patch(model, prop_attn=True)
    clip = torch.rand(3, 3, frames, 64, 64, device=DEV).to(torch.bfloat16)
    outs = {}
    for kernel in (True, False):
        monkeypatch.setattr(_common, "_ATTN_KERNEL", kernel)
        out, plans = _trace(tome, model, clip, 5)
        outs[kernel] = (out.float(), plans, model._tome_info["size"].float().clone())
    (o1, p1, s1), (o0, p0, s0) = outs[True], outs[False]
    assert [s[1] for s, _ in p1] == [s[1] for s, _ in p0]
    assert torch.equal(p1[0][1].src_idx, p0[0][1].src_idx) and torch.equal(p1[0][1].dst_idx, p0[0][1].dst_idx)
    assert float(s1.sum()) == float(s0.sum())
    assert float((o1 - o0).abs().max()) <= 0.05 * max(1.0, float(o0.abs().max()))


@pytest.mark.parametrize("original", [True, False])
def test_motionformer_key_half_of_proj_kv(original, monkeypatch):
    """use_original_code (the reference's default) sums over the trajectory tokens themselves: the value half of proj_kv
    is computed and never read (tome/patch/motionformer.py:123,130-134).  The patch then evaluates only the key half --
    the proj_kv MODULE is not called, the attention output matches the full projection's within bf16 rounding of one
    GEMM -- and with use_original_code=False it calls the module as the reference does."""
    tome, H = _hosts()
    import sys
    mf_patch = sys.modules["tome.patch.motionformer"]  # (tome.patch.motionformer the NAME is apply_patch, as in the reference)
    torch.manual_seed(0)
    model = H["motionformer"].Motionformer(img_size=64, patch_size=8, patch_size_temp=2, temporal_resolution=4,
                                           embed_dim=128, depth=2, num_heads=2, num_classes=9, use_original_code=original)
    model = model.to(DEV).to(torch.bfloat16).eval()
    tome.patch.motionformer(model, prop_attn=True)
    model.r = 4
    clip = [torch.rand(2, 3, 8, 64, 64, device=DEV).to(torch.bfloat16)]
    calls = []
    hooks = [b.attn.proj_kv.register_forward_pre_hook(lambda m, i: calls.append(1)) for b in model.blocks]
    with torch.no_grad():
        model(clip)
    # (a hooked module is not "stock": its forward must run -- the hand-made call is refused, for both settings)
    assert len(calls) == len(model.blocks)
    for hk in hooks:
        hk.remove()
    outs = {}
    for keys_only in (True, False):
        monkeypatch.setattr(mf_patch, "_KEYS_ONLY", keys_only)
        seen = []
        real = torch.nn.Linear.forward

        def spy(self, x, real=real, seen=seen):
            if any(self is b.attn.proj_kv for b in model.blocks):
                seen.append(1)
            return real(self, x)
        monkeypatch.setattr(torch.nn.Linear, "forward", spy)
        with torch.no_grad():
            outs[keys_only] = (model(clip).float(), len(seen))
        monkeypatch.setattr(torch.nn.Linear, "forward", real)
    (o1, n1), (o0, n0) = outs[True], outs[False]
    assert n0 == len(model.blocks)
    assert n1 == (0 if original else len(model.blocks)), (original, n1)
    assert float((o1 - o0).abs().max()) <= 0.02 * max(1.0, float(o0.abs().max()))


_CFG_YAML = """\
TRAIN:
  ENABLE: True
DATA:
  NUM_FRAMES: 16
  TEST_CROP_SIZE: 224
  INPUT_CHANNEL_NUM: [3]
MOTIONFORMER:
  PATCH_SIZE: 16
  PATCH_SIZE_TEMP: 2
  EMBED_DIM: 64
  DEPTH: 3
  NUM_HEADS: 4
  TEMPORAL_RESOLUTION: 8
  USE_MLP: True
  HEAD_ACT: tanh
MODEL:
  NUM_CLASSES: 13
  ARCH : motionformer
  MODEL_NAME: Motionformer
TEST:
  ENABLE: True
  BATCH_SIZE: 4
  NUM_ENSEMBLE_VIEWS: 2
  NUM_SPATIAL_CROPS: 1
NUM_GPUS: 1
RNG_SEED: 0
"""


def test_reference_command_lines_run(tmp_path, capsys):
    """the reference's `--cfg <yaml> --opts ...` command lines (experiments.sh:16-19,81-92)
    drive the patched models: MODEL_BENCHMARK harness and the multi-view test loop, on synthetic clips."""
    from hosts import harness
    y = tmp_path / "tome_motionformer_tiny.yaml"
    y.write_text(_CFG_YAML)
    res = harness.main_benchmark(["--cfg", str(y), "--opts", "TRAIN.ENABLE", "False", "TOME.ENABLE", "True",
                                  "TOME.R_VALUE", "16", "MODEL_BENCHMARK.WARMUP_ITERATIONS", "1",
                                  "MODEL_BENCHMARK.ITERATIONS", "2", "TEST.BATCH_SIZE", "2"])
    assert res["average_fps"] > 0 and res["iterations"] == 2 and res["batch"] == 2
    assert "Average fps is" in capsys.readouterr().out
    out = {}
    for opts in (["TOME.ENABLE", "True", "TOME.R_VALUE", "16", "TOME.SCHEDULE", "-1"],
                 ["TOME.ENABLE", "True", "TOME.R_VALUE", "8", "TOME.MODE", "hybrid", "TOME.THRESHOLD", "0.8"],
                 ["TOME.ENABLE", "True", "TOME.R_VALUE", "0"], []):
        r = harness.main_run_net(["--cfg", str(y), "--dtype", "fp32", "--opts", "TRAIN.ENABLE", "False",
                                  "TEST.NUM_SYNTHETIC_VIDEOS", "6"] + opts)
        assert r["videos"] == 6 and r["all_clips_seen"] and r["views_per_video"] == 2
        assert 0.0 <= r["top1_acc"] <= r["top5_acc"] <= 100.0
        out[tuple(opts)] = r
    # r = 0 is the unpatched model (same seeds -> same weights, clips and labels)
    assert out[("TOME.ENABLE", "True", "TOME.R_VALUE", "0")] == out[()]
    with pytest.raises(SystemExit):
        harness.main_run_net(["--cfg", str(y)])  # TRAIN.ENABLE True: training is out of scope, said loudly


def test_second_forward_on_another_stream_is_ordered_not_concurrent(monkeypatch):
    """Two forwards in flight on two HIP streams of one process never finish on this platform (two concurrent library
    GEMM grids -- persistent Stream-K kernels under either BLAS preference, profiles/r04_two_stream_probe_rocblas_kernel.txt;
    tools/probes/two_stream_gemm.py is the minimal reproduction, round 2's tools/two_stream.py the original observation).  The patched forward therefore orders itself behind a patched
    forward that is still in flight on another stream (`_common._guard_one_forward_in_flight`: the stream waits for
    that forward's end event, one RuntimeWarning), or refuses with TOME_ONE_FORWARD=raise.  Made deterministic here by
    a spin kernel in front of the first forward: its end event cannot be complete when the second forward is issued."""
    import warnings
    tome, H = _hosts()
    from tome.patch import _common
    torch.manual_seed(0)
    models = []
    for _ in range(2):
        torch.manual_seed(0)
        m = H["videomae"].VideoMAE(num_frames=8, img_size=64, patch_size=16, embed_dim=128, depth=3, num_heads=2,
                                   num_classes=9).to(DEV).to(torch.bfloat16).eval()
        tome.patch.videomae(m, prop_attn=False)
        m.r = 4
        models.append(m)
    clip = torch.rand(4, 3, 8, 64, 64, device=DEV).to(torch.bfloat16)
    with torch.no_grad():
        want = models[0]([clip]).clone()
        torch.cuda.synchronize()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        monkeypatch.setattr(_common, "_warned_two_streams", False)
        _common._in_flight.clear()
        with torch.cuda.stream(s1):
            torch.cuda._sleep(2_000_000_000)  # ~1 s: the first forward is certainly still in flight below, however
            # slowly a busy host issues it (0.1 s was not enough on one box in round 4)
            out_a = models[0]([clip])
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            with torch.cuda.stream(s2):
                out_b = models[1]([clip])
        assert any(issubclass(w.category, RuntimeWarning) and "in flight" in str(w.message) for w in caught)
        torch.cuda.synchronize()
        assert torch.equal(out_a, want) and torch.equal(out_b, want)
        # the refusing form
        monkeypatch.setenv("TOME_ONE_FORWARD", "raise")
        _common._in_flight.clear()
        with torch.cuda.stream(s1):
            torch.cuda._sleep(2_000_000_000)
            models[0]([clip])
        with torch.cuda.stream(s2), pytest.raises(RuntimeError, match="in flight"):
            models[1]([clip])
        torch.cuda.synchronize()
        _common._in_flight.clear()


@pytest.mark.parametrize("host_name", ["videomae", "vivit", "timesformer", "motionformer"])
def test_block_output_with_and_without_the_fc2_fold_bf16(host_name, monkeypatch):
    """The fc2 fold changes the rounding order of a block's last residual against the reference: the reference rounds
    x' + round(h W + b), the fold rounds round(x' + b) + h W accumulated in the GEMM (bias put into the stream by the
    merge kernel's x_out_bias, `finish_linear` with beta = 1).  A doubled or missing bias in ONE block would be hidden
    by a model-level logit tolerance, so this holds single blocks to each other: every block's OUTPUT with the fold on
    vs off (same input to the block, captured by hooks in the unfolded run and replayed) within 2 bf16 ulps of the
    largest magnitude of that output row... and the fold must actually have been taken and consumed (`_folded`)."""
    tome, H = _hosts()
    from tome.patch import _common
    if not all((_common._FUSE_LN, _common._FUSE_ADD, _common._FUSE_NEXT)):
        pytest.skip("the fold rides on the fused merge + LayerNorm hand-over, which a TOME_FUSE_* switch has turned off")
    torch.manual_seed(0)
    if host_name == "videomae":
        model = H["videomae"].VideoMAE(num_frames=8, img_size=64, patch_size=16, embed_dim=128, depth=3, num_heads=2,
                                       num_classes=9)
        patch, frames = tome.patch.videomae, 8
    elif host_name == "vivit":
        model = H["vivit"].ViViT(num_classes=9, image_size=64, num_frames=8, hidden_size=128, num_hidden_layers=3,
                                 num_attention_heads=2, intermediate_size=256)
        patch, frames = tome.patch.vivit, 8
    elif host_name == "timesformer":
        model = H["timesformer"].TimeSformer(num_frames=4, img_size=64, patch_size=8, embed_dim=128, depth=3,
                                             num_heads=2, num_classes=9)
        patch, frames = tome.patch.timesformer, 4
    else:
        model = H["motionformer"].Motionformer(img_size=64, patch_size=8, patch_size_temp=2, temporal_resolution=4,
                                               embed_dim=128, depth=3, num_heads=2, num_classes=9)
        patch, frames = tome.patch.motionformer, 8
    model = model.to(DEV).to(torch.bfloat16).eval()
    # give every bias a size that would show if it were applied twice or not at all
    with torch.no_grad():
        for name, p_ in model.named_parameters():
            if name.endswith("bias"):
                p_.copy_(torch.randn_like(p_) * 0.5)
    patch(model)
    clip = torch.rand(3, 3, frames, 64, 64, device=DEV).to(torch.bfloat16)
    blocks = [m for m in model.modules() if getattr(m.__class__, "_tome_tag", None) in ("ToMeBlock", "ToMeVivitLayer")]
    assert len(blocks) == 3
    outs = {}
    for fold in (False, True):
        monkeypatch.setattr(_common, "_FUSE_FC2", fold)
        seen = []
        hooks = [b.register_forward_hook(lambda m, i, o, seen=seen: seen.append((o[0] if isinstance(o, tuple) else o).float().clone()))
                 for b in blocks]
        folded_taken = []
        real_fl = _common.finish_linear

        def spy_fl(block, x, h, linear, info):
            folded_taken.append(info.get("_folded") is not None)
            return real_fl(block, x, h, linear, info)
        monkeypatch.setattr(_common, "finish_linear", spy_fl)
        logits, plans = _trace(tome, model, clip, 4)
        logits = logits.float()
        for hk in hooks:
            hk.remove()
        monkeypatch.setattr(_common, "finish_linear", real_fl)
        outs[fold] = (seen, logits, folded_taken, model._tome_info.get("_folded"), [p for _, p in plans])
    (s0, l0, f0, left0, p0), (s1, l1, f1, left1, p1) = outs[False], outs[True]
    assert not any(f0) and left0 is None
    if host_name in ("videomae", "vivit"):
        assert f1 and all(f1), "the fold was not taken in every block"
    print(f"{host_name}: finish_linear calls with a folded bias: {sum(f1)} of {len(f1)}")
    assert left1 is None, "a folded bias was left unconsumed at the end of the forward"
    assert len(s0) == len(s1) == 3
    compared = 0
    for k, (a, b) in enumerate(zip(s0, s1)):
        assert a.shape == b.shape
        # block 0 sees identical input in both runs; a later block is comparable as long as every matching up to and
        # including its own chose the same tokens (an ulp of difference upstream may legitimately merge others)
        same = all(torch.equal(getattr(x, n_), getattr(y, n_)) for x, y in zip(p0[:k + 1], p1[:k + 1])
                   for n_ in ("src_idx", "dst_idx", "unm_idx"))
        if k > 0 and not same:
            break
        tol = 2.0 * 2 ** -8 * float(a.abs().max()) * (k + 1)
        assert float((a - b).abs().max()) <= tol, (k, float((a - b).abs().max()), tol)
        compared += 1
    print(f"{host_name}: {compared} block outputs compared")
    assert compared >= 1


def test_closures_and_patches_are_differentiable_when_tokens_require_grad():
    """SURVEY 8b / tome/merge.py:49: only the matching is no_grad in the reference; `tools/train_net.py:727-741` patches
    models for training.  With tokens that require grad the closures of this package run on the framework's
    differentiable ops (the matching still on the HIP kernels): same values as the kernels' inference path, the same
    gradients as the reference's op sequence (oracle/torch_port.py on CPU with the same index tensors), and a patched
    model's forward + backward reaches every parameter -- also through the regrouped reductions of TimeSformer."""
    tome, H = _hosts()
    from oracle import torch_port
    from tome import merge as M
    g = torch.Generator(device=DEV).manual_seed(5)
    metric = torch.randn(4, 197, 64, device=DEV, generator=g)
    x = torch.randn(4, 197, 96, device=DEV, generator=g)
    size = torch.randint(1, 4, (4, 197, 1), device=DEV, generator=g).float()
    merge, unmerge = M.bipartite_soft_matching(metric, 16, class_token=True)
    with torch.no_grad():
        want_x, want_s = M.merge_wavg(merge, x, size)           # the kernels
    xg = x.clone().requires_grad_(True)
    got_x, got_s = M.merge_wavg(merge, xg, size)                # framework ops, differentiable
    assert got_x.requires_grad and torch.equal(got_s, want_s)
    torch.testing.assert_close(got_x, want_x, rtol=1e-6, atol=1e-6)
    got_x.square().sum().backward()
    p = merge.plan
    tp = torch_port.TorchPlan(p.r, p.src_idx.cpu(), p.dst_idx.cpu(), p.unm_idx.cpu(), p.T)
    xc = x.cpu().requires_grad_(True)
    ref_x, _ = torch_port.merge_wavg(tp, xc, size.cpu())
    ref_x.square().sum().backward()
    torch.testing.assert_close(xg.grad.cpu(), xc.grad, rtol=1e-5, atol=1e-6)
    back = unmerge(got_x.detach().requires_grad_(True))
    assert back.shape == x.shape and back.requires_grad
    # a patched model trains: VideoMAE (grouped reduction) and TimeSformer (regrouped reduction by views)
    torch.manual_seed(0)
    for name, model, clip, patch in (
            ("videomae", H["videomae"].VideoMAE(num_frames=8, img_size=64, patch_size=16, embed_dim=64, depth=3,
                                                num_heads=1, num_classes=9), torch.rand(2, 3, 8, 64, 64), tome.patch.videomae),
            ("timesformer", H["timesformer"].TimeSformer(num_frames=4, img_size=64, patch_size=8, embed_dim=64, depth=3,
                                                         num_heads=1, num_classes=9), torch.rand(2, 3, 4, 64, 64),
             tome.patch.timesformer)):
        model = model.to(DEV).train()
        patch(model)
        model.r = 6
        out = model([clip.to(DEV)])
        assert out.requires_grad, name
        out.square().sum().backward()
        missing = [n for n, q in model.named_parameters() if q.grad is None or not torch.isfinite(q.grad).all()]
        # (parameters the architecture never reads in this configuration may legitimately have no gradient)
        assert not [n for n in missing if "blocks" in n and ("qkv" in n or "mlp" in n or "norm" in n)], (name, missing)
        assert model._tome_info["size"].shape[1] < (196 if name == "videomae" else 64)
