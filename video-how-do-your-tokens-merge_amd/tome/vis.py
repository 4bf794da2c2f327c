"""Merged-token visualisations from the `source` matrix a `trace_source=True` forward leaves in
``model._tome_info["source"]`` (reference: tome/vis.py -- make_visualization :32-79,
make_spatial_video_visualization :81-130, make_spatiotemporal_video_visualization :132-177,
concatenate_images :179-187).  Offline CPU plotting, not part of the MI355X hot path; it is here so that the
reference's notebooks find every name they import from ``tome``.

Same pictures as the reference, computed once per pixel instead of once per merged token: every pixel
belongs to exactly one group (the merged token its patch went into); it shows the group's mean colour -- or
the video itself -- unless it lies on the group's outline (it or one of its four neighbours in the frame
belongs to another group, or it touches the frame border: a 3x3-cross erosion with zero padding), where it
shows the group's colour-map entry.  Work is O(pixels) rather than O(pixels * groups).
"""
from __future__ import annotations

import random
from typing import List, Tuple

import numpy as np
import torch
import torch.nn.functional as F


def generate_colormap(N: int, seed: int = 0) -> List[Tuple[float, float, float]]:
    """N colours from Python's Mersenne twister seeded with `seed` (the stream `random.seed(seed)` gives)."""
    rng = random.Random(seed)
    return [(rng.random(), rng.random(), rng.random()) for _ in range(N)]


def _labels(source: torch.Tensor, class_token: bool) -> torch.Tensor:
    """[n, merged, original(+cls)] 0/1 matrix -> [n, original]: index of the merged token each original went to."""
    source = source.detach().cpu()
    if class_token:
        source = source[:, :, 1:]
    return source, source.argmax(dim=1)


def _upsample(lab: torch.Tensor, size) -> np.ndarray:
    """Nearest-neighbour upsampling of an integer label volume [d, h, w] (torch's index rule)."""
    up = F.interpolate(lab.to(torch.float32)[None, None], size=tuple(size), mode="nearest")
    return up[0, 0].to(torch.int64).numpy()


def _interior(lab: np.ndarray) -> np.ndarray:
    """[t, h, w] labels -> bool: pixel and its four in-frame neighbours share a label and none is off-frame."""
    same = np.zeros(lab.shape, dtype=bool)
    core = lab[:, 1:-1, 1:-1]
    same[:, 1:-1, 1:-1] = ((core == lab[:, :-2, 1:-1]) & (core == lab[:, 2:, 1:-1]) &
                           (core == lab[:, 1:-1, :-2]) & (core == lab[:, 1:-1, 2:]))
    return same


def _group_colours(lab: np.ndarray, pixels: np.ndarray, groups: int) -> np.ndarray:
    """Mean colour of every group over the pixels it covers ([groups, 3], zeros for empty groups)."""
    flat = lab.reshape(-1)
    valid = flat < groups
    count = np.bincount(flat[valid], minlength=groups).astype(np.float64)
    out = np.zeros((groups, pixels.shape[-1]))
    for c in range(pixels.shape[-1]):
        s = np.bincount(flat[valid], weights=pixels[..., c].reshape(-1)[valid].astype(np.float64), minlength=groups)
        with np.errstate(invalid="ignore", divide="ignore"):
            out[:, c] = s / count
    out[~np.isfinite(out).all(axis=1)] = 0.0
    return out


def _paint(lab: np.ndarray, pixels: np.ndarray, groups: int, average_colour: bool = True) -> np.ndarray:
    """Float picture [t, h, w, 3]: group fill inside, colour-map outline, zero where lab >= groups."""
    colours = _group_colours(lab, pixels, groups)
    cmap = np.asarray(generate_colormap(groups), dtype=np.float64).reshape(groups, 3)
    drawn = lab < groups
    safe = np.where(drawn, lab, 0)
    inside = _interior(lab) & drawn
    fill = colours[safe] if average_colour else pixels.astype(np.float64)
    out = np.where(inside[..., None], fill, cmap[safe] if groups else 0.0)
    return np.where(drawn[..., None], out, 0.0)


def make_visualization(img, source: torch.Tensor, patch_size: int = 16, class_token: bool = True):
    """PIL image in, PIL image of the same size out (tome/vis.py:32-79)."""
    from PIL import Image
    pixels = np.array(img.convert("RGB")) / 255.0
    h, w, _ = pixels.shape
    ph, pw = h // patch_size, w // patch_size
    _, vis = _labels(source, class_token)
    groups = int(vis.max().item()) + 1
    lab = _upsample(vis.reshape(1, ph, pw), (1, h, w))
    return Image.fromarray(np.uint8(_paint(lab, pixels[None], groups)[0] * 255))


def make_spatial_video_visualization(video: torch.Tensor, source: torch.Tensor,
                                     patch_size: Tuple[int, int, int] = (16, 16, 2), class_token: bool = True,
                                     average_colour: bool = True) -> np.ndarray:
    """Per-tubelet groups: `source[k]` describes the tokens of temporal slice k (tome/vis.py:81-130).
    video [t, c, h, w] in [0, 1]; returns uint8 [t, h, w, 3]."""
    frames = video.permute(0, 2, 3, 1).numpy()
    t, h, w, _ = frames.shape
    source, _ = _labels(source, class_token)
    step = patch_size[2]
    ph, pw = h // patch_size[0], w // patch_size[1]
    out = []
    for k, f0 in enumerate(range(0, t, step)):
        vis = source[k][None].argmax(dim=1)
        groups = int(vis.max().item()) + 1
        lab = _upsample(vis.reshape(1, ph, pw), (step, h, w))
        out.append(_paint(lab, frames[f0:f0 + step], groups, average_colour))
    return np.uint8(np.concatenate(out) * 255)


def make_spatiotemporal_video_visualization(video: torch.Tensor, source: torch.Tensor,
                                            patch_size: Tuple[int, int, int] = (16, 16, 2),
                                            class_token: bool = True, average_colour: bool = True,
                                            separate: bool = False) -> Tuple[np.ndarray, List]:
    """Groups span space and time: one `source` [1, merged, pt*ph*pw (+cls)] for the whole clip
    (tome/vis.py:132-177).  Tokens no merged token accounts for stay black.  With `separate`, also one uint8
    picture per group (the reference's 225 scale included)."""
    frames = video.permute(0, 2, 3, 1).numpy()
    t, h, w, _ = frames.shape
    ph, pw, pt = h // patch_size[0], w // patch_size[1], t // patch_size[2]
    source, vis = _labels(source, class_token)
    groups = int(vis.max().item()) + 1
    vis = vis.clone()
    vis[source.sum(dim=1) == 0] = groups
    lab = _upsample(vis.reshape(pt, ph, pw), (t, h, w))
    pic = _paint(lab, frames, groups, average_colour)
    tokens = []
    if separate:
        colours = _group_colours(lab, frames, groups)
        inside = _interior(lab)
        for g in range(groups):
            m = (inside & (lab == g))[..., None]
            fill = colours[g] if average_colour else frames  # float64 colour / float32 video, as the reference multiplies
            tokens.append(np.uint8(np.where(m, fill, np.zeros((), dtype=fill.dtype)) * 225))
    return np.uint8(pic * 255), tokens


def concatenate_images(video: np.ndarray, ncols: int = 8, nrows: int = 4):
    """Contact sheet of the first ncols*nrows frames of a uint8 [n, h, w, 3] video (tome/vis.py:179-187)."""
    from PIL import Image
    n, h, w, c = video.shape
    sheet = video[:ncols * nrows].reshape(nrows, ncols, h, w, c).transpose(0, 2, 1, 3, 4).reshape(nrows * h, ncols * w, c)
    return Image.fromarray(np.ascontiguousarray(sheet))
