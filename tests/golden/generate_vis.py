#!/usr/bin/env python3
"""Golden pictures for tome/vis.py.  Runs ONLY in the build container (reference mounted read-only at
/root/reference): imports the reference's ``tome/vis.py`` by file path, feeds it the deterministic inputs of
``vis_inputs`` below (seeds only -- shared with tests/test_vis_cpu.py) and stores the pictures it drew in
``vis.npz``.  No reference source text is stored."""
from __future__ import annotations

import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import synth  # noqa: E402
from vis_cases import CASES, vis_inputs  # noqa: E402

REF = os.environ.get("TOME_REFERENCE", "/root/reference")
spec = importlib.util.spec_from_file_location("_ref_tome_vis", os.path.join(REF, "tome/vis.py"))
ref_vis = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_vis)


def main():
    out = {}
    for c in CASES:
        kind, cid = c["kind"], c["id"]
        pixels, source = vis_inputs(c)
        if kind == "image":
            from PIL import Image
            pic = ref_vis.make_visualization(Image.fromarray(pixels), source, patch_size=c["patch"][0],
                                             class_token=c["cls"])
            out[cid] = np.array(pic)
        elif kind == "spatial":
            out[cid] = ref_vis.make_spatial_video_visualization(torch.from_numpy(pixels), source, patch_size=c["patch"],
                                                                class_token=c["cls"], average_colour=c["avg"])
        else:
            vid, toks = ref_vis.make_spatiotemporal_video_visualization(
                torch.from_numpy(pixels), source, patch_size=c["patch"], class_token=c["cls"], average_colour=c["avg"],
                separate=True)
            out[cid] = vid
            out[cid + "_tokens"] = np.stack(toks)
    sheet = ref_vis.concatenate_images(out["st0"].repeat(2, axis=0), ncols=4, nrows=2)
    out["sheet"] = np.array(sheet)
    out["colormap7"] = np.asarray(ref_vis.generate_colormap(7))
    np.savez_compressed(os.path.join(HERE, "vis.npz"), **out)
    print("wrote vis.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
