"""`python bench.py --gpus 2` started plainly (no torchrun) on a one-GPU box: the program spawns its two ranks
itself (hosts/launch.py), both compute on device 0, gloo carries the one all-reduce (RCCL needs one device per
rank).  The same code path the driver takes on an 8-GPU node with the default backend."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]  # rank 0 prints ONE line
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_two_self_spawned_ranks():
    out = _run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--batch", "4", "--steps", "2",
                "--warmup", "1", "--no-roofline", "--no-cpu-baseline", "--no-also"])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["rank_devices"] == [0, 0]
    assert out["backend"] == "gloo" and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["clips_per_gpu_per_step"] == 4


@pytest.mark.gpu
def test_model_benchmark_two_self_spawned_ranks(tmp_path):
    """tools/model_benchmark.py with NUM_GPUS 2 (the reference's own switch, slowfast/utils/misc.py:402-430)."""
    env_backend = dict(os.environ, TOME_DIST_BACKEND="gloo")
    cmd = [sys.executable, "tools/model_benchmark.py", "--cfg", "configs/videomae_b_16x224.yaml", "--opts",
           "TRAIN.ENABLE", "False", "NUM_GPUS", "2", "TOME.ENABLE", "True", "TOME.R_VALUE", "16", "TOME.PROP_ATTN",
           "False", "MODEL_BENCHMARK.WARMUP_ITERATIONS", "1", "MODEL_BENCHMARK.ITERATIONS", "2", "TEST.BATCH_SIZE", "4"]
    env = {k: v for k, v in env_backend.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert res["ranks_seen"] == 2 and res["devices"] == [0, 0] and res["batch"] == 4 and res["average_fps"] > 0


@pytest.mark.gpu
def test_bench_line_carries_the_other_workloads():
    """The default bench line (here at small batch, `--also-quick`): headline + roofline with both byte counts + the
    `also` object with the other BASELINE.json sizes, each with its merge kernel's roofline, and the reference's
    batch-8 protocol eager / HIP graph."""
    out = _run([sys.executable, "bench.py", "--batch", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                "--also-quick"])
    roof = out["roofline"]
    assert roof["bound"] == "hbm" and 0 < roof["frac_8d"] < roof["frac"] < 1 and roof["bytes_8d"] < roof["bytes_fused"]
    also = out["also"]
    for key, rs in (("videomae_b_8x224", ("r16",)), ("timesformer_divst_8x224", ("r8", "r16", "r32")),
                    ("vivit_b_32x224", ("r64",)), ("motionformer_224_16x4", ("r16",))):
        for r in rs:
            rec = also[key][r]
            assert rec["clips_per_s"] > 0 and 0 < rec["roofline"]["frac_8d"] <= rec["roofline"]["frac"] < 1, (key, r, rec)
    proto = also["reference_protocol_batch8"]
    for fam in ("videomae_b_16x224", "timesformer_divst_8x224", "vivit_b_32x224", "motionformer_224_16x4"):
        vals = [v for k, v in proto[fam].items() if not k.endswith("_error")]
        assert len(vals) == 3 and all(v is not None and v > 0 for v in vals), proto[fam]


@pytest.mark.gpu
def test_rccl_group_of_one_runs_the_job_collectives():
    """RCCL on the one-GPU box: `bench.py --gpus 1 --force-group` forms the torch.distributed group with backend
    "nccl" (= RCCL on ROCm; device_id given, as hosts/launch.py does for N > 1) and runs the job's collectives on it
    -- the barrier pair around the timed steps, the ONE all-reduce of [top1, top5, clips], the max-reduce of the
    elapsed time, the census (all-reduce of ones + all-gather of device indices).  No scaling figure comes out of a
    group of one; what this pins is that RCCL initialises and the collectives the 8-GPU run depends on execute on
    MI355X through exactly the code path of that run (slowfast/utils/distributed.py:47-63,96-101)."""
    out = _run([sys.executable, "bench.py", "--gpus", "1", "--force-group", "--batch", "4", "--steps", "2", "--warmup",
                "1", "--no-roofline", "--no-cpu-baseline", "--no-also"])
    assert out["process_group"] is True and out["backend"] == "rccl"
    assert out["n_gpus"] == 1 and out["ranks_seen"] == 1 and out["rank_devices"] == [0] and out["value"] > 0


@pytest.mark.gpu
def test_rccl_census_and_count_reduction_on_device_tensors():
    """hosts/launch.census and hosts/evalloop.all_reduce_counts (the eval loop's collective, tools/train_net.py:515-522
    pattern) on device tensors over an RCCL group of one, in a process of its own (a process group is global state)."""
    code = (
        "import os, sys, json\n"
        f"sys.path[:0] = [{ROOT!r}, os.path.join({ROOT!r}, 'video-how-do-your-tokens-merge_amd')]\n"
        "import torch, torch.distributed as dist\n"
        "from hosts import launch\n"
        "from hosts.evalloop import all_reduce_counts, topk_counts\n"
        "dev = torch.device('cuda', 0); torch.cuda.set_device(dev)\n"
        "launch.init_process_group('nccl', dev, force=True)\n"
        "assert dist.is_initialized() and dist.get_backend() == 'nccl' and dist.get_world_size() == 1\n"
        "c = launch.census(dev)\n"
        "logits = torch.randn(16, 400, device=dev); labels = logits.argmax(1)\n"
        "counts = all_reduce_counts(topk_counts(logits, labels))\n"
        "torch.cuda.synchronize()\n"
        "print(json.dumps({'census': c, 'counts': counts.tolist()}))\n"
        "dist.destroy_process_group()\n")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert res["census"] == {"ranks_seen": 1, "devices": [0]} and res["counts"] == [16, 16, 16]
