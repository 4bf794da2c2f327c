"""GPU tests of tome_tubelet_rows (the regrouping in front of the hosts' patch-embedding GEMM): a pure move, so the
result must EQUAL the framework's reshape / permute of the same clip bit for bit, for every element size, tubelet
shape and input view the four hosts use -- and the embeddings built on it must equal the convolution the reference
runs (slowfast/models/videomae_video_model_builder.py:137-166 `PatchEmbed.proj` and the other three models')."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _rows_by_views(x, kt, kh, kw):
    B, C, T, H, W = x.shape
    nt, nh, nw = T // kt, H // kh, W // kw
    return x.reshape(B, C, nt, kt, nh, kh, nw, kw).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B, nt * nh * nw, C * kt * kh * kw)


CASES = [
    # dtype, B, C, T, H, W, kt, kh, kw, view
    (torch.bfloat16, 3, 3, 16, 224, 224, 2, 16, 16, "plain"),     # VideoMAE / Motionformer / ViViT-16 tubelets
    (torch.bfloat16, 2, 3, 8, 224, 224, 1, 16, 16, "plain"),      # TimeSformer: per-frame patches, clip read in place
    (torch.bfloat16, 2, 3, 32, 224, 224, 2, 16, 16, "btchw"),     # ViViT: [B, T, C, H, W] clips through a permuted view
    (torch.float16, 2, 3, 4, 64, 64, 2, 8, 8, "plain"),           # one 16-byte chunk per run
    (torch.float32, 2, 3, 4, 64, 96, 2, 16, 16, "plain"),         # four chunks per run
    (torch.float32, 1, 2, 2, 8, 20, 1, 2, 4, "plain"),            # W' = 5: the strip loop's tail
    (torch.bfloat16, 2, 1, 6, 32, 368, 3, 16, 8, "plain"),        # W' = 46 > several unrolled rounds, kt = 3
    (torch.bfloat16, 2, 3, 4, 64, 64, 2, 16, 16, "batch_slice"),  # a view with a larger batch stride
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{str(c[0])[6:]}-{c[1]}x{c[2]}x{c[3]}x{c[4]}x{c[5]}-k{c[6]}.{c[7]}.{c[8]}-{c[9]}")
def test_tubelet_rows_equal_the_permuted_copy(case):
    from tome import _abi
    dtype, B, C, T, H, W, kt, kh, kw, view = case
    torch.manual_seed(0)
    if view == "btchw":
        x = torch.randn(B, T, C, H, W, device=DEV).to(dtype).permute(0, 2, 1, 3, 4)
    elif view == "batch_slice":
        x = torch.randn(2 * B, C, T, H, W, device=DEV).to(dtype)[::2]
    else:
        x = torch.randn(B, C, T, H, W, device=DEV).to(dtype)
    assert _abi.tubelet_rows_ok(x, kt, kh, kw)
    got = _abi.tubelet_rows(x, kt, kh, kw)
    want = _rows_by_views(x, kt, kh, kw)
    assert got.shape == want.shape and got.is_contiguous() and torch.equal(got, want)


def test_tubelet_rows_refuses_what_it_cannot_move():
    from tome import _abi
    x = torch.randn(2, 3, 4, 32, 32, device=DEV).bfloat16()
    assert not _abi.tubelet_rows_ok(x, 2, 16, 4)          # runs of 8 bytes
    assert not _abi.tubelet_rows_ok(x, 3, 16, 16)         # clip is not whole tubelets
    assert not _abi.tubelet_rows_ok(x[..., 1:], 2, 16, 8)  # W 31: not whole tubelets either, and misaligned rows
    assert not _abi.tubelet_rows_ok(x.transpose(3, 4), 2, 16, 16)  # no unit stride along W
    assert not _abi.tubelet_rows_ok(x.cpu(), 2, 16, 16)
    with pytest.raises(_abi.TomeHipError):
        _abi.tubelet_rows(x, 2, 16, 4)
    with pytest.raises(_abi.TomeHipError):
        _abi.tubelet_rows(x.cpu(), 2, 16, 16)
    # the C entry checks for itself
    import ctypes
    rows = torch.empty(2 * 2 * 2 * 8 * 3 * 2 * 16 * 4, device=DEV, dtype=torch.bfloat16)
    strides = (ctypes.c_int64 * 4)(*x.stride()[:4])
    rc = _abi.lib().tome_tubelet_rows(x.data_ptr(), 2, 2, 3, 4, 32, 32, strides, 2, 16, 4, rows.data_ptr(), 0)
    assert rc != 0 and b"16-byte" in _abi.lib().tome_last_error()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_embeddings_on_the_kernel_equal_the_convolution(dtype, monkeypatch):
    """tubelet_tokens / frame_patch_tokens (hosts/_patchify.py) against the convolution itself, with the regrouping
    kernel and with the framework's permute: the two GEMM forms are the SAME product (bit-equal), and both are the
    convolution up to its different summation order."""
    from hosts import _patchify
    torch.manual_seed(0)
    conv3 = torch.nn.Conv3d(3, 96, kernel_size=(2, 16, 16), stride=(2, 16, 16)).to(DEV).to(dtype)
    conv2 = torch.nn.Conv2d(3, 96, kernel_size=16, stride=16).to(DEV).to(dtype)
    x = torch.rand(2, 3, 8, 64, 96, device=DEV).to(dtype)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    with torch.no_grad():
        outs = {}
        for on in (True, False):
            monkeypatch.setattr(_patchify, "_ROWS_KERNEL", on)
            outs[on] = (_patchify.tubelet_tokens(conv3, x), _patchify.frame_patch_tokens(conv2, x))
        assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])
        want3 = conv3(x).flatten(2).transpose(1, 2)
        want2 = conv2(x.transpose(1, 2).reshape(16, 3, 64, 96)).flatten(2).transpose(1, 2)
        assert outs[True][0].shape == want3.shape and outs[True][1].shape == want2.shape
        assert float((outs[True][0].float() - want3.float()).abs().max()) <= tol
        assert float((outs[True][1].float() - want2.float()).abs().max()) <= tol
