"""Minimal PyTorch host models that give the ToMe patches something to sit in (benchmarks and tests).
They are plumbing around the merge path, not the product: unpatched, their attention and MLP run on
PyTorch-ROCm's own kernels (the patches bring their own attention, tome_prop_attention).  Attribute names
follow the reference's checkpoints so its state_dicts load."""
