"""N>1 path on CPU: two gloo ranks shard a clip set, count top-1/top-5 locally and meet in ONE
all-reduce; the result must equal the single-process count."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd"))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd"))
    from hosts.evalloop import all_reduce_counts, shard_range, topk_counts
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(123)
    logits = torch.randn(total, 50, generator=g)
    labels = torch.randint(0, 50, (total,), generator=g)
    lo, hi = shard_range(total, rank, world)
    counts = torch.zeros(3, dtype=torch.int64)
    for a in range(lo, hi, 4):  # a few local steps, accumulate on "device"
        b = min(hi, a + 4)
        counts += topk_counts(logits[a:b], labels[a:b])
    all_reduce_counts(counts)
    torch.save(counts, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [37, 64])
def test_two_ranks_one_allreduce(tmp_path, total):
    from hosts.evalloop import shard_range, topk_counts
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, str(tmp_path)), nprocs=world, join=True)
    g = torch.Generator().manual_seed(123)
    logits = torch.randn(total, 50, generator=g)
    labels = torch.randint(0, 50, (total,), generator=g)
    want = topk_counts(logits, labels)
    for r in range(world):
        got = torch.load(os.path.join(str(tmp_path), f"r{r}.pt"), weights_only=True)
        assert torch.equal(got, want), (got, want)
    assert want[2].item() == total and want[1] >= want[0]
    # shards tile the range
    spans = [shard_range(total, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == total and spans[0][1] == spans[1][0]


def test_shard_range_properties():
    from hosts.evalloop import shard_range
    for total in (0, 1, 7, 8, 100):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_clip_ensemble_meter_matches_per_clip_loop():
    """ClipEnsembleMeter == the per-clip accumulation loop of slowfast's TestMeter (sum and max ensembling),
    fed in shuffled batches."""
    from hosts.evalloop import ClipEnsembleMeter
    g = torch.Generator().manual_seed(5)
    V, K, C = 23, 3, 11
    preds = torch.randn(V * K, C, generator=g)
    labels = torch.randint(0, C, (V,), generator=g).repeat_interleave(K)
    clip_ids = torch.arange(V * K)
    perm = torch.randperm(V * K, generator=g)
    for method in ("sum", "max"):
        meter = ClipEnsembleMeter(V, K, C, ensemble_method=method)
        for a in range(0, V * K, 7):
            idx = perm[a:a + 7]
            meter.update(preds[idx], labels[idx], clip_ids[idx])
        ref = torch.zeros(V, C)
        for i in range(V * K):  # the reference's loop (meters.py:337-358)
            v = int(clip_ids[i]) // K
            ref[v] = ref[v] + preds[i] if method == "sum" else torch.max(ref[v], preds[i])
        assert torch.allclose(meter.video_preds, ref, atol=1e-6)
        stats = meter.finalize()
        top = ref.topk(5, dim=1).indices
        lab = labels[::K]
        assert stats["videos"] == V and stats["all_clips_seen"]
        assert abs(stats["top1_acc"] - 100.0 * (top[:, 0] == lab).float().mean().item()) < 1e-4
        assert abs(stats["top5_acc"] - 100.0 * (top == lab[:, None]).any(1).float().mean().item()) < 1e-4


def _census_rank(out_dir):
    """A rank as bench.py / hosts.harness run it: environment -> process group -> census -> one all-reduce."""
    sys.path.insert(0, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd"))
    from hosts import launch
    from hosts.evalloop import all_reduce_counts
    rank, local, world = launch.check_world(2)
    launch.init_process_group("gloo")
    seen = launch.census(torch.device("cpu"))
    counts = torch.tensor([rank + 1, 10 * (rank + 1), 7], dtype=torch.int64)
    all_reduce_counts(counts)
    torch.save({"rank": rank, "local": local, "world": world, "seen": seen, "counts": counts,
                "port": os.environ["MASTER_PORT"], "addr": os.environ["MASTER_ADDR"]},
               os.path.join(out_dir, f"census{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_launcher_spawns_its_ranks(tmp_path, monkeypatch):
    """hosts.launch.run from a plain process (no torchrun environment) = the reference's launch_job: N fresh
    ranks over a TCP rendezvous on 127.0.0.1, every rank sees the whole job."""
    from hosts import launch
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    assert not launch.under_launcher() and launch.rank_env() == (0, 0, 1)
    launch.run(_census_rank, 2, (str(tmp_path),))
    assert "RANK" not in os.environ  # the parent's environment is untouched
    got = [torch.load(os.path.join(str(tmp_path), f"census{r}.pt"), weights_only=True) for r in range(2)]
    for r, g in enumerate(got):
        assert (g["rank"], g["local"], g["world"]) == (r, r, 2)
        assert g["seen"] == {"ranks_seen": 2, "devices": [-1, -1]}
        assert g["counts"].tolist() == [3, 30, 14]
        assert g["addr"] == "127.0.0.1" and g["port"] == got[0]["port"]


def test_launcher_refuses_a_job_of_the_wrong_size(monkeypatch):
    from hosts import launch
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "4")
    assert launch.under_launcher()
    with pytest.raises(SystemExit, match="WORLD_SIZE=4"):
        launch.check_world(2)
    monkeypatch.setenv("RANK", "5")
    with pytest.raises(SystemExit, match="outside"):
        launch.rank_env()
    # bench.py applies the same check before anything else
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    with pytest.raises(SystemExit, match="WORLD_SIZE=4"):
        bench.main()


def test_bench_spawns_when_started_plainly(monkeypatch):
    """`python bench.py --gpus 2` without a launcher environment hands `worker` to hosts.launch.run with 2 ranks
    (and `--gpus 1` runs it in-process) -- checked without a GPU by intercepting the spawn."""
    sys.path.insert(0, ROOT)
    import bench
    from hosts import launch
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    calls = []
    monkeypatch.setattr(launch, "run", lambda fn, n, args=(): calls.append((fn, n, args)))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--backend", "gloo"])
    bench.main()
    assert calls[0][0] is bench.worker and calls[0][1] == 2 and calls[0][2][0].backend == "gloo"


def _world8_rank(out_dir, failing_rank):
    """A rank of the 8-rank job: group over gloo, census, the job's one all-reduce; `failing_rank` raises before the
    collectives (the others must not hang the parent: mp.spawn tears the job down and re-raises)."""
    sys.path.insert(0, os.path.join(ROOT, "video-how-do-your-tokens-merge_amd"))
    from hosts import launch
    from hosts.evalloop import all_reduce_counts
    rank, local, world = launch.check_world(8)
    if rank == failing_rank:
        raise RuntimeError(f"rank {rank} fails on purpose")
    launch.init_process_group("gloo")
    seen = launch.census(torch.device("cpu"))
    launch.check_census(seen, "gloo", world)
    counts = torch.tensor([rank, 1, 2], dtype=torch.int64)
    all_reduce_counts(counts)
    torch.save({"rank": rank, "local": local, "seen": seen, "counts": counts, "port": os.environ["MASTER_PORT"],
                "local_world": os.environ["LOCAL_WORLD_SIZE"]}, os.path.join(out_dir, f"w8_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_launcher_world_8_and_a_failing_rank(tmp_path, monkeypatch):
    """hosts.launch.run with EIGHT ranks (the size of the scaling run) on CPU / gloo: one free port for all, the rank
    environment of every child, the census of 8, the one all-reduce, join -- and a rank that raises makes the PARENT
    raise (no silent partial job, no hang)."""
    from hosts import launch
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    launch.run(_world8_rank, 8, (str(tmp_path), -1))
    got = [torch.load(os.path.join(str(tmp_path), f"w8_{r}.pt"), weights_only=True) for r in range(8)]
    assert [g["rank"] for g in got] == list(range(8)) and [g["local"] for g in got] == list(range(8))
    assert all(g["seen"] == {"ranks_seen": 8, "devices": [-1] * 8} for g in got)
    assert all(g["counts"].tolist() == [28, 8, 16] for g in got)
    assert len({g["port"] for g in got}) == 1 and all(g["local_world"] == "8" for g in got)
    with pytest.raises(Exception, match="fails on purpose"):
        launch.run(_world8_rank, 8, (str(tmp_path), 3))


def test_rccl_refuses_more_ranks_than_gpus_and_stacked_devices(monkeypatch):
    """bench.py / hosts.harness refuse an RCCL job with more ranks on the node than GPUs before the process group
    exists (ranks stacked on one device die inside RCCL instead of saying so), and a finished RCCL job whose census
    shows two ranks on one device is an error; gloo may share a device (dry runs of the N > 1 path)."""
    from hosts import launch
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")
    with pytest.raises(SystemExit, match="one GPU per rank"):
        launch.require_one_gpu_per_rank("nccl", 2)
    launch.require_one_gpu_per_rank("gloo", 2)
    launch.require_one_gpu_per_rank("nccl", 1)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    launch.require_one_gpu_per_rank("nccl", 8)
    launch.check_census({"ranks_seen": 8, "devices": list(range(8))}, "nccl", 8)
    with pytest.raises(SystemExit, match="not one GPU per rank"):
        launch.check_census({"ranks_seen": 8, "devices": [0, 1, 2, 3, 4, 5, 6, 6]}, "nccl", 8)
    launch.check_census({"ranks_seen": 2, "devices": [0, 0]}, "gloo", 2)
    with pytest.raises(SystemExit, match="ranks answered"):
        launch.check_census({"ranks_seen": 7, "devices": list(range(7))}, "nccl", 8)
