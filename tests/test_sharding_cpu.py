"""N > 1 path on CPU: world_size-2 gloo run of the shard plan + score gather, with per-clip values that depend only
on the GLOBAL utterance index (the numpy Philox oracle), so the gathered result must equal the 1-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from audiopure_amd.sharding import all_gather_scores, shard_bounds
from oracle.philox import philox_normal


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 512, 4096, 4099):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _scores(start, stop, K=10):
    # stand-in for per-clip log-probs: a function of the global utterance index only
    return torch.from_numpy(philox_normal(77, 0, start, stop - start, K)) if stop > start else torch.zeros((0, K))


def _worker(rank, world, port, n_total, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s, e = shard_bounds(n_total, rank, world)
    got = all_gather_scores(_scores(s, e), n_total)
    if rank == 0:
        np.save(out, got.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 7])
def test_world2_gather_equals_single_process(tmp_path, n_total):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "g.npy")
    mp.spawn(_worker, args=(2, port, n_total, out), nprocs=2, join=True)
    assert np.array_equal(np.load(out), _scores(0, n_total).numpy())


def test_world8_uneven_gather_equals_single_process(tmp_path):
    """BASELINE configs[2]'s rank count with a batch that does not divide: 4099 clips over 8 ranks (shards of 513 / 512),
    the padded all_gather trimmed back to the global order."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "g8.npy")
    mp.spawn(_worker, args=(8, port, 4099, out), nprocs=8, join=True)
    assert np.array_equal(np.load(out), _scores(0, 4099).numpy())


def test_single_process_passthrough():
    x = torch.arange(12.0).view(4, 3)
    assert all_gather_scores(x, 4) is x


def test_bench_self_launches_its_ranks_as_children_and_prints_one_line():
    """`python bench.py --gpus 2 --dry-run` (the driver's command form, no torchrun around it): the launcher starts two
    gloo ranks as child processes and rank 0 prints exactly one JSON line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1",
                        "--batch", "8"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["config"]["global_batch"] == 16 and out["scaling"] == "weak"
    # the self-evidencing fields of an N > 1 line: group size as the backend formed it, per-rank clocks, per-rank devices
    rk = out["ranks"]
    assert rk["backend"] == "gloo" and rk["ranks"] == 2 and rk["distinct_devices"] == 2
    assert len(rk["rank_elapsed_s"]) == 2 and rk["rank_elapsed_min_s"] <= rk["rank_elapsed_max_s"]
    assert [d["index"] for d in rk["devices"]] == [0, 1]
    assert out["ms_per_step"] * out["steps"] >= rk["rank_elapsed_min_s"] * 1e3 * 0.5


def test_bench_eight_rank_rehearsal():
    """The driver's SCALE command form at N = 8 (`python bench.py --gpus 8 ...`), rehearsed on CPU / gloo: eight child ranks,
    eight device descriptors, one line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--dry-run", "--steps", "2", "--warmup", "1",
                        "--batch", "4"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    rk = out["ranks"]
    assert out["n_gpus"] == 8 and out["config"]["global_batch"] == 32 and out["scaling"] == "weak"
    assert rk["backend"] == "gloo" and rk["ranks"] == 8 and rk["distinct_devices"] == 8 and len(rk["rank_elapsed_s"]) == 8
    assert [d["index"] for d in rk["devices"]] == list(range(8))
    assert len({d["pci_bus_id"] for d in rk["devices"]}) == 8


def test_bench_power_sampler_is_optional_evidence():
    """bench.py's per-leg `power` object (board power / cap / shader clock polled from rocm-smi) must never be a reason for a leg
    to fail: without a GPU driver (this container) or without rocm-smi the sampler yields None."""
    import time
    import bench
    with bench.PowerSampler(0) as ps:
        time.sleep(0.1)
    r = ps.result()
    assert r is None or {"board_W_mean", "board_W_max", "cap_W", "sclk_MHz_mean", "samples"} <= set(r)


def test_config2_split_reproduces_the_single_process_digest():
    """BASELINE configs[2]'s split (4 096 clips as 8 x 512) through bench.py's REAL step function -- global batch from one seed,
    contiguous shards, the scores all_gather, the `scores.sha256` digest of the line -- around a CPU stand-in scorer (gloo): the
    8-rank digest must equal the 1-rank digest of the same global batch.  The two-rank RCCL test on hardware
    (tests/test_gpu_dropin.py) relies on exactly this plumbing."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    digests = {}
    for gpus, batch in ((1, 4096), (8, 512), (2, 2048)):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--dry-run", "--steps", "1", "--warmup", "0",
                            "--batch", str(batch)], capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout
        out = json.loads(lines[0])
        assert out["config"]["global_batch"] == 4096 and out["scores"]["shape"] == [4096, 10]
        digests[gpus] = out["scores"]["sha256"]
    assert digests[8] == digests[1] == digests[2], digests
