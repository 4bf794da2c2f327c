"""Command-line compatible drivers for the patched models: the reference's
``--cfg <yaml> --opts KEY VALUE ...`` interface for

* the throughput harness   tools/model_benchmark.py -> slowfast/utils/model_benchmark.py:20-113
  (MODEL_BENCHMARK.WARMUP_ITERATIONS / ITERATIONS, TEST.BATCH_SIZE, one event pair per forward, average
  time per frame and frames per second), and
* the multi-view test loop  tools/run_net.py -> tools/test_net.py:27-207,259-283 (TOME.* options, the
  ensembling of slowfast/utils/meters.py:324-359, top-k of slowfast/utils/metrics.py:9-41),

so `experiments.sh`-style command lines run unchanged against the MI355X merge path.  The reference's dataset
decoders, checkpoints and yacs are not part of the hot path and not rebuilt: the config is read with
yaml.safe_load over the defaults below, models are the random-init hosts of this package, and clips are
synthetic (`torch.rand`, as model_benchmark.py:36 does) -- labels are random, so the accuracies only exercise
the counting.  One process per GPU; under torch.distributed.run every rank takes its shard of the videos and
the only collective is the final all-reduce of the counts (hosts/evalloop.py).
"""
from __future__ import annotations

import argparse
import copy
import json
import os
from types import SimpleNamespace

import torch
import torch.distributed as dist
import yaml

# the keys of slowfast/config/{defaults,custom_config}.py that the two drivers read, with the reference's defaults
DEFAULTS = {
    "TRAIN": {"ENABLE": True},
    "TEST": {"ENABLE": True, "BATCH_SIZE": 8, "NUM_ENSEMBLE_VIEWS": 10, "NUM_SPATIAL_CROPS": 3,
             "NUM_SYNTHETIC_VIDEOS": 32},  # NUM_SYNTHETIC_VIDEOS: ours (there is no dataset to count)
    "DATA": {"NUM_FRAMES": 8, "TEST_CROP_SIZE": 224, "INPUT_CHANNEL_NUM": [3], "ENSEMBLE_METHOD": "sum"},
    "MODEL": {"MODEL_NAME": "VideoMAE", "NUM_CLASSES": 400},
    "VIDEOMAE": {"TUBELET_SIZE": 2, "USE_MEAN_POOLING": True, "INIT_SCALE": 0.001},
    "TIMESFORMER": {"ATTENTION_TYPE": "divided_space_time"},
    "MOTIONFORMER": {"PATCH_SIZE": 16, "PATCH_SIZE_TEMP": 2, "EMBED_DIM": 768, "DEPTH": 12, "NUM_HEADS": 12,
                     "TEMPORAL_RESOLUTION": 8, "USE_MLP": True, "HEAD_ACT": "tanh"},
    "VIVIT": {"CONFIG_PATH": ""},
    "TOME": {"ENABLE": False, "R_VALUE": 0, "SCHEDULE": 0, "PROP_ATTN": True, "HEAD_AGGREGATION": "mean",
             "MODE": "merge", "THRESHOLD": -1.0, "LAYER_TO_DUPLICATE": 0, "LAYER_QUANTITY": 1},
    "MODEL_BENCHMARK": {"WARMUP_ITERATIONS": 0, "ITERATIONS": 0},
    "NUM_GPUS": 1,
    "NUM_SHARDS": 1,
    "DIST_BACKEND": "nccl",  # = RCCL on ROCm (slowfast/config/defaults.py:912)
    "RNG_SEED": 0,
}


def _merge(dst: dict, src: dict) -> dict:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


def _namespace(d):
    return SimpleNamespace(**{k: _namespace(v) if isinstance(v, dict) else v for k, v in d.items()})


def load_cfg(cfg_file: str | None, opts=()) -> SimpleNamespace:
    """Defaults <- YAML file <- `KEY.SUB VALUE` pairs (values parsed as YAML scalars/lists, like yacs does)."""
    d = copy.deepcopy(DEFAULTS)
    if cfg_file:
        with open(cfg_file) as f:
            _merge(d, yaml.safe_load(f) or {})
    opts = list(opts)
    if len(opts) % 2:
        raise ValueError("--opts takes KEY VALUE pairs")
    for key, val in zip(opts[::2], opts[1::2]):
        node = d
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = yaml.safe_load(val) if isinstance(val, str) else val
    return _namespace(d)


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description="ToMe video models on MI355X (reference-compatible command line)")
    ap.add_argument("--cfg", dest="cfg_file", default=None)
    ap.add_argument("--dtype", choices=("fp32", "bf16", "fp16"), default="bf16")
    ap.add_argument("--graph", action="store_true",
                    help="replay the forward from a HIP graph (hosts/graphed.py): static shapes, no launch cost")
    ap.add_argument("--opts", nargs=argparse.REMAINDER, default=[])
    ap.add_argument("--init_method", default="tcp://127.0.0.1:9999")  # accepted and ignored: torchrun env is used
    return ap.parse_args(argv)


def build_model(cfg) -> torch.nn.Module:
    name, classes, frames = cfg.MODEL.MODEL_NAME, cfg.MODEL.NUM_CLASSES, cfg.DATA.NUM_FRAMES
    if name == "VideoMAE":
        from .videomae import videomae_base
        return videomae_base(num_frames=frames, num_classes=classes, tubelet_size=cfg.VIDEOMAE.TUBELET_SIZE,
                             use_mean_pooling=cfg.VIDEOMAE.USE_MEAN_POOLING, init_scale=cfg.VIDEOMAE.INIT_SCALE)
    if name == "TimeSformer":
        from .timesformer import timesformer_base
        return timesformer_base(num_frames=frames, num_classes=classes,
                                attention_type=cfg.TIMESFORMER.ATTENTION_TYPE)
    if name == "Motionformer":
        from .motionformer import Motionformer
        m = cfg.MOTIONFORMER
        return Motionformer(img_size=cfg.DATA.TEST_CROP_SIZE, patch_size=m.PATCH_SIZE, patch_size_temp=m.PATCH_SIZE_TEMP,
                            temporal_resolution=m.TEMPORAL_RESOLUTION, num_classes=classes, embed_dim=m.EMBED_DIM,
                            depth=m.DEPTH, num_heads=m.NUM_HEADS, use_mlp=m.USE_MLP, head_act=m.HEAD_ACT)
    if name == "ViViT":
        from .vivit import ViViT, VivitConfig
        kw = {"num_frames": frames, "image_size": cfg.DATA.TEST_CROP_SIZE}
        path = cfg.VIVIT.CONFIG_PATH
        if path and os.path.exists(path):  # the HF-style JSON of configs/vivit/*/tome_vivit_*.json
            with open(path) as f:
                j = json.load(f)
            kw.update(num_frames=j["video_size"][0], image_size=j["video_size"][1], tubelet_size=j["tubelet_size"],
                      hidden_size=j["hidden_size"], num_hidden_layers=j["num_hidden_layers"],
                      num_attention_heads=j["num_attention_heads"], intermediate_size=j["intermediate_size"],
                      layer_norm_eps=j["layer_norm_eps"], qkv_bias=j.get("qkv_bias", True))
        return ViViT(VivitConfig(**kw), num_classes=classes)
    raise ValueError(f"unknown MODEL.MODEL_NAME {name!r}")


def apply_tome(model, cfg, with_threshold: bool) -> None:
    """tools/test_net.py:259-283 (threshold passed) / slowfast/utils/model_benchmark.py:82-103 (not passed)."""
    if not cfg.TOME.ENABLE:
        return
    import tome
    name = cfg.MODEL.MODEL_NAME
    patch_func = {"TimeSformer": tome.patch.timesformer, "Motionformer": tome.patch.motionformer,
                  "ViViT": tome.patch.vivit, "VideoMAE": tome.patch.videomae}[name]
    duplicate_func = {"TimeSformer": tome.patch.duplicate_timesformer, "Motionformer": tome.patch.duplicate_motionformer,
                      "ViViT": tome.patch.duplicate_vivit, "VideoMAE": tome.patch.duplicate_videomae}[name]
    t = cfg.TOME
    if t.LAYER_QUANTITY > 1:
        t.R_VALUE = [0] * t.LAYER_TO_DUPLICATE + [t.R_VALUE] * t.LAYER_QUANTITY + [0] * (11 - t.LAYER_TO_DUPLICATE)
        duplicate_func(model, layer_to_duplicate=t.LAYER_TO_DUPLICATE, quantity=t.LAYER_QUANTITY)
    kw = {"threshold": t.THRESHOLD} if with_threshold else {}
    patch_func(model, prop_attn=t.PROP_ATTN, mode=t.MODE, head_aggregation=t.HEAD_AGGREGATION, **kw)
    model.r = (t.R_VALUE, t.SCHEDULE)


_DTYPES = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}


def _setup(cfg, dtype: str):
    """One rank = one process = one GPU (slowfast/utils/multiprocessing.py:8-62).  cfg.NUM_GPUS is the job size:
    the ranks come from torch.distributed.run or from our own spawn (hosts/launch.py), never from this call."""
    from . import launch
    rank, local, world = launch.check_world(cfg.NUM_GPUS * cfg.NUM_SHARDS)
    if not torch.cuda.is_available():
        raise SystemExit("the merge path has no CPU implementation: an MI355X is required")
    backend = os.environ.get("TOME_DIST_BACKEND", getattr(cfg, "DIST_BACKEND", "nccl"))
    launch.require_one_gpu_per_rank(backend, world)
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    launch.init_process_group(backend, dev)
    torch.manual_seed(cfg.RNG_SEED)
    model = build_model(cfg).to(dev).to(_DTYPES[dtype]).eval()
    return model, dev, rank, world


def _input_shape(cfg, batch):
    return (batch, cfg.DATA.INPUT_CHANNEL_NUM[0], cfg.DATA.NUM_FRAMES, cfg.DATA.TEST_CROP_SIZE, cfg.DATA.TEST_CROP_SIZE)


@torch.no_grad()
def perform_benchmark(model, cfg, dev, dtype: str, world: int = 1, graph: bool = False) -> dict:
    """slowfast/utils/model_benchmark.py:20-58: a fresh random batch per iteration, generated outside the event
    pair; the per-iteration times of all ranks are summed like `sum(du.all_gather_unaligned(time))`."""
    mb = cfg.MODEL_BENCHMARK
    total = mb.ITERATIONS + mb.WARMUP_ITERATIONS
    shape = _input_shape(cfg, cfg.TEST.BATCH_SIZE // max(1, world))
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    times = []
    forward = model
    if graph:
        from .graphed import GraphedForward
        forward = GraphedForward(model, [torch.rand(shape, device=dev).to(_DTYPES[dtype])])
    for _ in range(total):
        clip = [torch.rand(shape, device=dev).to(_DTYPES[dtype])]
        start.record()
        forward(clip)
        end.record()
        torch.cuda.synchronize()
        t = torch.tensor([start.elapsed_time(end)], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t)
        times.append(float(t.item()))
    timed = times[mb.WARMUP_ITERATIONS:]
    frames = cfg.TEST.BATCH_SIZE * cfg.DATA.NUM_FRAMES * max(1, mb.ITERATIONS)
    frame_time = sum(timed) / frames / 1000.0
    return {"average_frame_time_s": frame_time, "average_fps": (1.0 / frame_time) if frame_time else float("nan"),
            "clips_per_s": (1.0 / (frame_time * cfg.DATA.NUM_FRAMES)) if frame_time else float("nan"),
            "iterations": mb.ITERATIONS, "warmup": mb.WARMUP_ITERATIONS, "batch": cfg.TEST.BATCH_SIZE}


@torch.no_grad()
def perform_test(model, cfg, dev, dtype: str, rank: int = 0, world: int = 1) -> dict:
    """tools/test_net.py:27-207 on synthetic videos: every video is seen through NUM_ENSEMBLE_VIEWS *
    NUM_SPATIAL_CROPS clips, clip scores are ensembled per video, top-1 / top-5 counted once at the end."""
    from .evalloop import ClipEnsembleMeter, shard_range
    views = cfg.TEST.NUM_ENSEMBLE_VIEWS * cfg.TEST.NUM_SPATIAL_CROPS
    videos = cfg.TEST.NUM_SYNTHETIC_VIDEOS
    meter = ClipEnsembleMeter(videos, views, cfg.MODEL.NUM_CLASSES, device=dev, ensemble_method=cfg.DATA.ENSEMBLE_METHOD)
    lo, hi = shard_range(videos, rank, world)
    gen = torch.Generator(device=dev).manual_seed(cfg.RNG_SEED)
    labels_all = torch.randint(0, cfg.MODEL.NUM_CLASSES, (videos,), device=dev, generator=gen)
    clip_ids = torch.arange(lo * views, hi * views, device=dev)
    per = max(1, cfg.TEST.BATCH_SIZE // max(1, world))
    for i in range(0, clip_ids.numel(), per):
        ids = clip_ids[i:i + per]
        clip = [torch.rand(_input_shape(cfg, ids.numel()), device=dev).to(_DTYPES[dtype])]
        meter.update(model(clip), labels_all[torch.div(ids, views, rounding_mode="floor")], ids)
    out = meter.finalize(ks=(1, 5), videos=(lo, hi))
    out.update(views_per_video=views)
    return out


def _finish(res: dict, dev, rank: int, out_path) -> dict:
    from . import launch
    res.update(launch.census(dev))  # ranks_seen / devices: what the job consisted of
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(res), flush=True)
        if out_path:
            with open(out_path, "w") as f:
                json.dump(res, f)
    return res


def _benchmark_rank(argv, out_path=None) -> dict:
    args = parse_args(argv)
    cfg = load_cfg(args.cfg_file, args.opts)
    model, dev, rank, world = _setup(cfg, args.dtype)
    apply_tome(model, cfg, with_threshold=False)
    res = perform_benchmark(model, cfg, dev, args.dtype, world, graph=args.graph)
    if rank == 0:
        print(f"Average time per frame is {res['average_frame_time_s']}(s) after {res['iterations']} iterations")
        print(f"Average fps is {res['average_fps']}(im/s) after {res['iterations']} iterations")
    return _finish(res, dev, rank, out_path)


def _run_net_rank(argv, out_path=None) -> dict:
    args = parse_args(argv)
    cfg = load_cfg(args.cfg_file, args.opts)
    model, dev, rank, world = _setup(cfg, args.dtype)
    apply_tome(model, cfg, with_threshold=True)
    return _finish(perform_test(model, cfg, dev, args.dtype, rank, world), dev, rank, out_path)


def _launch(rank_fn, argv) -> dict:
    """launch_job (slowfast/utils/misc.py:402-430): NUM_GPUS > 1 -> one spawned process per GPU over a TCP
    rendezvous, else the function is called in this process.  Nothing here touches the GPU, so the children start
    from a parent that never initialised it.  Rank 0's result comes back through a file (a spawned rank cannot
    return a value)."""
    import sys
    import tempfile
    from . import launch
    argv = list(sys.argv[1:] if argv is None else argv)
    cfg = load_cfg(parse_args(argv).cfg_file, parse_args(argv).opts)
    n = int(cfg.NUM_GPUS) * int(cfg.NUM_SHARDS)
    if launch.under_launcher() or n <= 1:
        return rank_fn(argv)
    if int(cfg.NUM_SHARDS) > 1:
        # the reference starts NUM_GPUS ranks PER SHARD (machine) with rank = SHARD_ID * NUM_GPUS + local
        # (slowfast/utils/misc.py:414-428); spawning all NUM_GPUS * NUM_SHARDS ranks here would stack them on this one
        # node's GPUs.  Several nodes take an external launcher (torch.distributed.run --nnodes ...).
        raise SystemExit("NUM_SHARDS > 1: start one launcher per node (torch.distributed.run --nnodes NUM_SHARDS "
                         "--nproc-per-node NUM_GPUS ...); the self-spawn covers one node")
    with tempfile.TemporaryDirectory() as tmp:
        out_path = os.path.join(tmp, "rank0.json")
        launch.run(rank_fn, n, (argv, out_path))
        with open(out_path) as f:
            return json.load(f)


def main_benchmark(argv=None) -> dict:
    return _launch(_benchmark_rank, argv)


def main_run_net(argv=None) -> dict:
    argv_l = list(__import__("sys").argv[1:] if argv is None else argv)
    cfg = load_cfg(parse_args(argv_l).cfg_file, parse_args(argv_l).opts)
    if cfg.TRAIN.ENABLE:
        raise SystemExit("training is outside the scope of this package (inference merge path): pass TRAIN.ENABLE False")
    if not cfg.TEST.ENABLE:
        return {}
    return _launch(_run_net_rank, argv_l)
