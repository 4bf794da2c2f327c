"""Host-model plumbing that runs without a GPU: the matrix-product patch embeddings equal the convolutions
they replace (same weights, same token order)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "video-how-do-your-tokens-merge_amd")]

from hosts._patchify import patch_tokens, tubelet_tokens  # noqa: E402


@pytest.mark.parametrize("shape,kernel", [((2, 3, 4, 32, 48), (2, 16, 16)), ((1, 3, 8, 16, 16), (2, 8, 4)),
                                          ((3, 2, 3, 12, 12), (1, 4, 4))])
def test_tubelet_tokens_equal_conv3d(shape, kernel):
    torch.manual_seed(0)
    conv = torch.nn.Conv3d(shape[1], 24, kernel_size=kernel, stride=kernel)
    x = torch.randn(shape)
    want = conv(x).flatten(2).transpose(1, 2)
    got = tubelet_tokens(conv, x)
    assert got.shape == want.shape
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)


def test_patch_tokens_equal_conv2d_and_fallbacks():
    torch.manual_seed(1)
    conv = torch.nn.Conv2d(3, 16, kernel_size=8, stride=8)
    x = torch.randn(5, 3, 32, 24)
    torch.testing.assert_close(patch_tokens(conv, x), conv(x).flatten(2).transpose(1, 2), rtol=1e-5, atol=1e-5)
    # overlapping / padded / ragged convolutions keep the library path
    over = torch.nn.Conv2d(3, 16, kernel_size=8, stride=4)
    assert torch.equal(patch_tokens(over, x), over(x).flatten(2).transpose(1, 2))
    pad = torch.nn.Conv3d(3, 8, kernel_size=(2, 4, 4), stride=(2, 4, 4), padding=(0, 1, 1))
    x3 = torch.randn(1, 3, 4, 8, 8)
    assert torch.equal(tubelet_tokens(pad, x3), pad(x3).flatten(2).transpose(1, 2))
    ragged = torch.nn.Conv3d(3, 8, kernel_size=(2, 4, 4), stride=(2, 4, 4))
    x4 = torch.randn(1, 3, 5, 9, 8)
    assert torch.equal(tubelet_tokens(ragged, x4), ragged(x4).flatten(2).transpose(1, 2))


def test_harness_config_defaults_file_and_opts(tmp_path):
    """The reference's `--cfg file --opts KEY VALUE ...` interface (slowfast/utils/parser.py + yacs semantics):
    defaults <- YAML <- opts, values parsed as YAML scalars."""
    from hosts import harness
    y = tmp_path / "c.yaml"
    y.write_text("TRAIN:\n  ENABLE: True\nDATA:\n  NUM_FRAMES: 16\n  INPUT_CHANNEL_NUM: [3]\nMODEL:\n  MODEL_NAME: VideoMAE\n"
                 "TEST:\n  BATCH_SIZE: 32\nNUM_GPUS: 2\nUNKNOWN_SECTION:\n  KEY: 1\n")
    a = harness.parse_args(["--cfg", str(y), "--opts", "TRAIN.ENABLE", "False", "TOME.ENABLE", "True", "TOME.R_VALUE",
                            "150", "TOME.PROP_ATTN", "False", "TOME.SCHEDULE", "-1", "TOME.THRESHOLD", "0.8",
                            "MODEL_BENCHMARK.ITERATIONS", "100", "WANDB.ENABLE", "True"])
    cfg = harness.load_cfg(a.cfg_file, a.opts)
    assert cfg.TRAIN.ENABLE is False and cfg.TOME.ENABLE is True and cfg.TOME.R_VALUE == 150
    assert cfg.TOME.PROP_ATTN is False and cfg.TOME.SCHEDULE == -1 and cfg.TOME.THRESHOLD == 0.8
    assert cfg.DATA.NUM_FRAMES == 16 and cfg.TEST.BATCH_SIZE == 32 and cfg.NUM_GPUS == 2
    assert cfg.TOME.MODE == "merge" and cfg.TOME.HEAD_AGGREGATION == "mean"  # defaults of custom_config.py
    assert cfg.MODEL_BENCHMARK.ITERATIONS == 100 and cfg.MODEL_BENCHMARK.WARMUP_ITERATIONS == 0
    assert cfg.WANDB.ENABLE is True and cfg.UNKNOWN_SECTION.KEY == 1
    with pytest.raises(ValueError):
        harness.load_cfg(None, ["TOME.ENABLE"])


def test_harness_builds_the_four_families():
    from hosts import harness
    tiny = ["MOTIONFORMER.EMBED_DIM", "48", "MOTIONFORMER.DEPTH", "2", "MOTIONFORMER.NUM_HEADS", "4",
            "MODEL.MODEL_NAME", "Motionformer", "MODEL.NUM_CLASSES", "7"]
    m = harness.build_model(harness.load_cfg(None, tiny))
    assert len(m.blocks) == 2 and m.head.out_features == 7 if hasattr(m.head, "out_features") else True
    for name, attr in (("VideoMAE", "model"), ("TimeSformer", "model"), ("ViViT", "vivit")):
        cfg = harness.load_cfg(None, ["MODEL.MODEL_NAME", name, "DATA.NUM_FRAMES", "8", "MODEL.NUM_CLASSES", "11"])
        with torch.device("meta"):
            mod = harness.build_model(cfg)
        assert hasattr(mod, attr)
    with pytest.raises(ValueError):
        harness.build_model(harness.load_cfg(None, ["MODEL.MODEL_NAME", "SlowFast"]))


def test_fold_decisions_of_the_block_glue():
    """The host logic that decides whether a block's last Linear may accumulate onto the residual stream in place
    (tome/patch/_common.py: _plain_mlp, foldable, finish_linear, mlp_residual) -- no kernel runs here."""
    from tome.patch import _common as C

    class Mlp(torch.nn.Module):
        def __init__(self, act):
            super().__init__()
            self.fc1, self.act, self.fc2 = torch.nn.Linear(8, 16), act, torch.nn.Linear(16, 8)
            self.drop = torch.nn.Dropout(0.0)

        def forward(self, x):
            return self.drop(self.fc2(self.act(self.fc1(x))))

    plain = Mlp(torch.nn.GELU()).eval()
    assert C._plain_mlp(plain)
    assert not C._plain_mlp(Mlp(torch.nn.GELU(approximate="tanh")).eval())   # not the exact-erf form
    assert not C._plain_mlp(Mlp(torch.nn.ReLU()).eval())
    assert not C._plain_mlp(Mlp(torch.nn.GELU()).train())                    # dropout may be live
    extra = Mlp(torch.nn.GELU()).eval()
    extra.norm = torch.nn.LayerNorm(16)                                      # a child the fast path does not know
    assert not C._plain_mlp(extra)

    with torch.no_grad():
        assert C.foldable(plain.fc2) is plain.fc2
        assert C.foldable(plain.fc2, eval_mode=False) is None
        assert C.foldable(torch.nn.Linear(4, 4, bias=False)) is None
        assert C.foldable(torch.nn.Identity()) is None
    assert C.foldable(plain.fc2) is None                                     # grad mode with trainable weights
    frozen = Mlp(torch.nn.GELU()).eval().requires_grad_(False)
    assert C.foldable(frozen.fc2) is frozen.fc2

    # finish_linear without a folded bias is the plain residual (on CPU: x + linear(h)), and clears nothing it does not own
    class Block(torch.nn.Module):
        pass
    blk = Block().eval()
    x, h, info = torch.randn(2, 5, 8), torch.randn(2, 5, 16), {}
    with torch.no_grad():
        torch.testing.assert_close(C.finish_linear(blk, x, h, plain.fc2, info), x + plain.fc2(h))
        # a folded bias that belongs to another Linear, or an MLP the fold was not made for, is refused loudly
        info["_folded"] = (x, torch.nn.Linear(16, 8))
        with pytest.raises(RuntimeError):
            C.finish_linear(blk, x, h, plain.fc2, info)
        info["_folded"] = (x, plain.fc2)
        with pytest.raises(RuntimeError):
            C.mlp_residual(blk, Mlp(torch.nn.ReLU()).eval(), x, torch.randn(2, 5, 8), info)
        # a fold recorded for ANOTHER tensor is not this block's business: plain path
        other = {"_folded": (torch.randn(2, 5, 8), plain.fc2)}
        y = torch.randn(2, 5, 8)
        torch.testing.assert_close(C.mlp_residual(blk, plain, x, y, other), x + plain(y))
