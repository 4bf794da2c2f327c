"""Minimal PyTorch host models that give the ToMe patches something to sit in (benchmarks and tests).
They are plumbing around the merge path, not the product: attention and MLP run on PyTorch-ROCm's
own kernels.  Attribute names follow the reference's checkpoints so its state_dicts load."""
