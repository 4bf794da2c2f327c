"""The PyTorch-CPU port that bench.py times as `cpu_baseline` (oracle/torch_port.py) against the golden
vectors of the real reference: same op sequence => same indices (certificate-aware) and bit-identical
fp32 values.  Also: the port's VideoMAE forward walks the expected token schedule."""
import numpy as np
import pytest
import torch

import golden_io as G
from oracle import torch_port


@pytest.mark.parametrize("case", [c for c in G.match_cases() if c["T"] <= 800 and not c["distill"]],
                         ids=lambda c: c["id"])
def test_port_matching(case):
    plan = torch_port.match(torch.from_numpy(G.metric_of(case)), case["r"], case["cls"], case["distill"])
    if case["r_eff"] == 0:
        assert plan is None
        return
    z = G.arrays("match")
    G.check_indices(case, plan.src_idx.numpy(), plan.dst_idx.numpy(), plan.unm_idx.numpy(),
                    z[case["id"] + "_src"], z[case["id"] + "_dst"], z[case["id"] + "_unm"])


@pytest.mark.parametrize("case", [c for c in G.value_cases("wavg") if not c["distill"]], ids=lambda c: c["id"])
def test_port_merge_wavg(case):
    plan = torch_port.match(torch.from_numpy(G.metric_of(case)), case["r"], case["cls"], False)
    size = G.size_of(case)
    xo, so = torch_port.merge_wavg(plan, torch.from_numpy(G.x_of(case)), None if size is None else torch.from_numpy(size))
    z = G.arrays("values")
    np.testing.assert_array_equal(xo.numpy(), z[case["id"] + "_x"])
    np.testing.assert_array_equal(so.numpy(), z[case["id"] + "_size"])


def test_port_videomae_token_schedule():
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "video-how-do-your-tokens-merge_amd"))
    from hosts.videomae import VideoMAE
    torch.manual_seed(0)
    host = VideoMAE(num_frames=4, img_size=64, patch_size=16, embed_dim=32, depth=4, num_heads=2, num_classes=7).eval()
    trace = []
    out = torch_port.videomae_forward(host, torch.rand(2, 3, 4, 64, 64), 5, trace=trace)
    assert out.shape == (2, 7) and torch.isfinite(out).all()
    assert [t for t, _ in trace] == [32, 27, 22, 17]  # 2*4*4 tokens, minus 5 per layer
    assert torch_port.schedule(12, (16, -1)) == [32, 29, 26, 23, 20, 17, 14, 11, 8, 5, 2, 0]
