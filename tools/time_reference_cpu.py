#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (reads /root/reference; the GPU box never runs this).

Times the REAL reference -- tome/patch/videomae.py over slowfast/models/videomae_video_model_builder.py, imported
through the name-only stand-ins of tests/golden/generate_models.py -- next to oracle/torch_port.py (what bench.py
reports as cpu_baseline, kind "port") on the same clips, same weights, same thread count, fp32, PyTorch CPU:
VideoMAE-B 16x224x224, r=16, 2 clips per forward.  Backs the claim that the port costs what the reference costs;
the pair goes to profiles/r02_cpu_reference_vs_port.json.
    python tools/time_reference_cpu.py [--iters 4] [--threads 8]
"""
import argparse
import importlib
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd"), os.path.join(ROOT, "tests")]
import torch  # noqa: E402

import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--clips", type=int, default=2)
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    spec = importlib.util.spec_from_file_location("_gen_models", os.path.join(ROOT, "tests", "golden", "generate_models.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    gen.install_stubs()
    vm = importlib.import_module("slowfast.models.videomae_video_model_builder")
    pv = importlib.import_module("tome.patch.videomae")
    cfg = dict(img_size=224, patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True,
               num_classes=400, all_frames=16, tubelet_size=2, init_values=0.0, init_scale=1.0)
    ref = gen._wrap(vm.VisionTransformer(norm_layer=lambda d: torch.nn.LayerNorm(d, eps=1e-6), **cfg).eval(), "VideoMAEWrap")
    synth.fill_parameters(ref, 77)
    pv.apply_patch(ref, prop_attn=False)
    ref.r = 16

    from hosts.videomae import videomae_base
    from oracle import torch_port
    host = videomae_base(num_frames=16, init_scale=1.0).eval()
    synth.fill_parameters(host, 77)

    clips = torch.from_numpy(synth.uniform01((a.clips, 3, 16, 224, 224), 5))

    def timed(fn):
        fn()  # warm-up
        t0 = time.perf_counter()
        for _ in range(a.iters):
            out = fn()
        return (time.perf_counter() - t0) / a.iters, out

    with torch.no_grad():
        t_ref, out_ref = timed(lambda: ref([clips]))
        t_port, out_port = timed(lambda: torch_port.videomae_forward(host, clips, 16))
    diff = float((out_ref - out_port).abs().max())
    res = {"workload": f"VideoMAE-B 16x224x224, r=16, fp32, {a.clips} clips per forward, {a.iters} timed forwards, "
                       f"torch {torch.__version__} CPU, {torch.get_num_threads()} threads (build container)",
           "reference": {"what": "tome/patch/videomae.py + slowfast/models/videomae_video_model_builder.py (imported from "
                                 "/root/reference)", "s_per_forward": round(t_ref, 3),
                         "clips_per_s": round(a.clips / t_ref, 3)},
           "port": {"what": "oracle/torch_port.py over hosts/videomae.py (bench.py cpu_baseline, kind 'port')",
                    "s_per_forward": round(t_port, 3), "clips_per_s": round(a.clips / t_port, 3)},
           "port_over_reference_time": round(t_port / t_ref, 3), "max_abs_logit_difference": diff}
    print(json.dumps(res, indent=1))
    with open(os.path.join(ROOT, "profiles", "r02_cpu_reference_vs_port.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
