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
