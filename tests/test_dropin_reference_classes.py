"""The drop-in claim of INTEGRATION.md section 1, executed: THIS package's `tome.patch.*` applied to the REFERENCE's own
model classes (slowfast's VideoMAE `VisionTransformer` under a `.model` wrapper, the inner TimeSformer
`VisionTransformer(attention_type='divided_space_time')`, `Motionformer`) -- the objects tools/test_net.py:259-283
hands to `tome.patch.<arch>(model, ...)`.

Build container only: needs /root/reference (skipped where it is absent, e.g. on the GPU box); nothing of the
reference is shipped or copied -- its model files are imported where they lie, through the name-only stand-ins of
tests/golden/generate_models.py (timm / fvcore / torchvision names, no arithmetic).  CPU, no kernel runs: the assertions
are structural (every Block / Attention instance swizzled, one shared `_tome_info` with the reference's keys, `model.r`
parsed into the reference's table), and a CPU forward must end in the package's loud `TomeHipError("... no CPU path")`
-- not in an AttributeError from a detector that never saw these classes."""
import importlib
import os
import sys

import pytest
import torch

REF = os.environ.get("TOME_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "slowfast", "models")),
                                reason="the reference checkout is not present (build container only)")

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_INFO_KEYS = {"r", "size", "source", "trace_source", "prop_attn", "verbose", "class_token", "distill_token",
                       "mode", "head_aggregation", "threshold"}  # tome/patch/videomae.py:180-193


@pytest.fixture(scope="module")
def ref_models():
    """The reference's model modules, imported next to (not instead of) this package's `tome`: the stand-ins put
    name-only `tome` / `tome.patch` packages into sys.modules for the reference's own patch files, so the product
    package is loaded first under its real name and restored afterwards."""
    import tome as product_tome          # this repository's package (conftest put it on sys.path)
    import tome.patch as product_patch   # noqa: F401
    saved = {k: v for k, v in sys.modules.items() if k == "tome" or k.startswith("tome.") or k.startswith("slowfast")
             or k.startswith("timm") or k.startswith("torchvision")}
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import generate_models as GM
    GM.install_stubs()
    try:
        vm = importlib.import_module("slowfast.models.videomae_video_model_builder")
        tsm = importlib.import_module("slowfast.models.timesformer")
        mb = importlib.import_module("slowfast.models.motionformer_video_model_builder")
        mvh = importlib.import_module("slowfast.models.motionformer_vit_helper")
    finally:
        for k in [k for k in sys.modules if k == "tome" or k.startswith("tome.")]:
            del sys.modules[k]
        sys.modules.update({k: v for k, v in saved.items() if k == "tome" or k.startswith("tome.")})
    assert sys.modules["tome"] is product_tome
    yield dict(GM=GM, vm=vm, tsm=tsm, mb=mb, mvh=mvh, tome=product_tome)
    for k in [k for k in sys.modules if k.startswith("slowfast") or k.startswith("timm") or k.startswith("torchvision")]:
        if k not in saved:
            del sys.modules[k]


def _ln(d):
    return torch.nn.LayerNorm(d, eps=1e-6)


def _small_videomae(R):
    inner = R["vm"].VisionTransformer(img_size=32, patch_size=8, embed_dim=128, depth=3, num_heads=2, mlp_ratio=4,
                                      qkv_bias=True, num_classes=10, all_frames=8, tubelet_size=2, init_values=0.0,
                                      norm_layer=_ln).eval()
    return R["GM"]._wrap(inner, "VideoMAEWrap"), (1, 3, 8, 32, 32), R["vm"].Block, R["vm"].Attention


def _small_timesformer(R):
    inner = R["tsm"].VisionTransformer(img_size=48, patch_size=8, num_classes=10, embed_dim=128, depth=3, num_heads=2,
                                       mlp_ratio=4, qkv_bias=True, num_frames=4, drop_path_rate=0.0,
                                       attention_type="divided_space_time", norm_layer=_ln).eval()
    return R["GM"]._wrap(inner, "TimeSformerWrap"), (1, 3, 4, 48, 48), R["tsm"].Block, R["tsm"].Attention


def _small_motionformer(R):
    from types import SimpleNamespace as NS
    cfg = NS(DATA=NS(TRAIN_CROP_SIZE=64), MODEL=NS(NUM_CLASSES=10), EPICKITCHENS=NS(NUM_CLASSES=None),
             MOTIONFORMER=NS(PATCH_SIZE=16, CHANNELS=3, EMBED_DIM=128, DEPTH=3, NUM_HEADS=2, MLP_RATIO=4, QKV_BIAS=True,
                             DROP=0.0, DROP_PATH=0.0, HEAD_DROPOUT=0.0, VIDEO_INPUT=True, TEMPORAL_RESOLUTION=4,
                             USE_MLP=True, ATTN_DROPOUT=0.0, HEAD_ACT="tanh", PATCH_SIZE_TEMP=2, POS_DROPOUT=0.0,
                             POS_EMBED="separate", ATTN_LAYER="trajectory", USE_ORIGINAL_TRAJ_ATTN_CODE=True,
                             APPROX_ATTN_TYPE="none", APPROX_ATTN_DIM=128))
    model = R["mb"].Motionformer(cfg).eval()
    return model, (1, 3, 8, 64, 64), R["mvh"].Block, R["mvh"].TrajectoryAttention


FAMILIES = {"videomae": _small_videomae, "timesformer": _small_timesformer, "motionformer": _small_motionformer}


@pytest.mark.parametrize("family", sorted(FAMILIES))
def test_patch_applies_to_the_references_own_classes(family, ref_models):
    R = ref_models
    tome = R["tome"]
    model, clip_shape, block_cls, attn_cls = FAMILIES[family](R)
    patch = getattr(tome.patch, family)
    assert patch.__module__.startswith("tome.patch.") and "video-how-do-your-tokens-merge_amd" in \
        sys.modules[patch.__module__].__file__, "the product package's patch, not the reference's"
    inner = model if family == "motionformer" else model.model
    blocks = [m for m in inner.modules() if isinstance(m, block_cls)]
    attns = [m for m in inner.modules() if isinstance(m, attn_cls)]
    assert len(blocks) == 3 and len(attns) >= 3
    R["GM"].synth.fill_parameters(model, 55)   # (the reference zero-inits Motionformer's tubelet projection)
    torch.manual_seed(0)
    clip = torch.rand(*clip_shape)
    with torch.no_grad():
        unpatched = model([clip]).clone()
    patch(model)     # tools/test_net.py:276/282: tome.patch.<arch>(model, ...)
    # every Block / attention instance of the reference's classes is swizzled (still an instance of its own class)
    for b in blocks:
        assert isinstance(b, block_cls) and getattr(type(b), "_tome_tag", None) == "ToMeBlock", type(b)
    tagged = [a for a in attns if getattr(type(a), "_tome_tag", None) is not None]
    if family == "timesformer":
        # timesformer.py:283-285: only `block.attn` (the spatial attention) becomes a ToMeAttention; temporal_attn stays
        assert {id(a) for a in tagged} == {id(b.attn) for b in blocks}
    else:
        assert len(tagged) == len(attns) == 3
    assert all(isinstance(a, attn_cls) for a in tagged)
    # one shared state dict, the reference's keys (videomae.py:180-193), the reference's defaults
    info = model._tome_info
    assert all(b._tome_info is info for b in blocks)
    # (timesformer.py:263-274 / motionformer.py:256-267 carry no "head_aggregation": they ignore that argument)
    assert set(info) >= REFERENCE_INFO_KEYS - (set() if family == "videomae" else {"head_aggregation"})
    assert info["prop_attn"] is (family != "videomae") and info["mode"] == "merge" and info["trace_source"] is False
    assert info["class_token"] is False and info["distill_token"] is False and model.r == 0
    assert getattr(type(model), "_tome_tag", None) == "ToMeVisionTransformer"
    assert all(callable(b.reduction_function) for b in blocks)
    # r = 0: the patched classes run the reference model's forward on CPU unchanged
    with torch.no_grad():
        out0 = model([clip])
    assert out0.shape == (1, 10) and torch.isfinite(out0).all()
    torch.testing.assert_close(out0, unpatched, rtol=1e-4, atol=1e-5)  # r = 0 merges nothing: the model's own output
    assert info["r"] == []  # parse_r's per-layer list, consumed block by block
    # model.r = (r, -1): parsed into the reference's decreasing table at the next forward (utils.py:83-108), and the first
    # layer that merges refuses loudly on CPU
    model.r = (4, -1)
    from tome._abi import TomeHipError
    with torch.no_grad(), pytest.raises(TomeHipError, match="no CPU path"):
        model([clip])
    from tome.utils import parse_r
    table = parse_r(3, (4, -1))
    assert table == [8, 4, 0] and info["r"] == table[1:]   # block 0 popped its r and raised in the matching


def test_duplicate_patches_apply_to_the_references_classes(ref_models):
    """tome.patch.duplicate_<arch> (tools/test_net.py:259-274) on the reference's classes: VideoMAE inserts deep copies
    that only attend + merge (videomae.py:154-157), TimeSformer / Motionformer re-insert the same block object
    (timesformer.py:170-172, motionformer.py:230-232)."""
    R = ref_models
    tome = R["tome"]
    model, _, block_cls, _ = _small_videomae(R)
    tome.patch.duplicate_videomae(model, 1, 2)
    tome.patch.videomae(model)
    kinds = [getattr(type(b), "_tome_tag", None) for b in model.model.blocks]
    assert kinds == ["ToMeBlock", "ToMeDuplicateBlock", "ToMeBlock", "ToMeBlock"]
    assert all(b._tome_info is model._tome_info for b in model.model.blocks)
    model, _, block_cls, _ = _small_timesformer(R)
    tome.patch.duplicate_timesformer(model, 1, 2)
    tome.patch.timesformer(model)
    assert len(model.model.blocks) == 4 and model.model.blocks[1] is model.model.blocks[2]
    model, _, block_cls, _ = _small_motionformer(R)
    tome.patch.duplicate_motionformer(model, 1, 2)
    tome.patch.motionformer(model)
    assert len(model.blocks) == 4 and model.blocks[1] is model.blocks[2]
    assert all(getattr(type(b), "_tome_tag", None) == "ToMeBlock" for b in model.blocks)
