"""N > 1 path on CPU: two gloo ranks shard a global batch contiguously and join their "decoded images" with the one
all-gather frido_amd.pipeline uses on RCCL (the same torch.distributed call; only the backend differs)."""
import os
import subprocess
import sys

import pytest

from frido_amd.pipeline import shard_range

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_contiguously():
    for total in (1, 7, 16, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from frido_amd.pipeline import shard_range, all_gather_images
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
for total in (8, 7):
    lo, hi = shard_range(total, rank, world)
    # a rank's "decoded images" depend only on GLOBAL sample indices, like the Philox-keyed sampler
    local = torch.stack([torch.full((3, 4, 4), float(i)) for i in range(lo, hi)])
    full = all_gather_images(local, total=total)
    assert full.shape == (total, 3, 4, 4), full.shape
    assert torch.equal(full[:, 0, 0, 0], torch.arange(total, dtype=torch.float32)), full[:, 0, 0, 0]
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_all_gather_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % REPO)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)],
                         capture_output=True, text=True, env=env, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 2


WORKER2 = r"""
import os, sys, types, torch, torch.distributed as dist
sys.path.insert(0, %r)
import frido_amd.samplers as smp
from frido_amd.pipeline import shard_range, sample_images

class StubSampler:                      # stands in for the HIP sampler: a sample depends only on (seed, GLOBAL index, its cond)
    def __init__(self, model, **kw):
        pass
    def sample(self, S, batch_size, shape, conditioning, num_stage, eta, verbose, unconditional_guidance_scale,
               unconditional_conditioning, noise, seed, sample0, log_every_t):
        assert noise == "philox" and conditioning.shape[0] == batch_size
        idx = torch.arange(sample0, sample0 + batch_size, dtype=torch.float32)
        z = (seed * 1000 + idx).view(-1, 1, 1, 1).expand(batch_size, *shape) + conditioning.mean(dim=(1, 2)).view(-1, 1, 1, 1)
        return z.contiguous(), {}
smp.DDIMSampler = smp.PLMSSampler = StubSampler

class Model:
    def __init__(self):
        self.model = types.SimpleNamespace(diffusion_model=types.SimpleNamespace(in_channels=6, image_size=4, num_stage=2))
    def decode_first_stage(self, z):
        return z[:, :3] * 2.0

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
for total in (8, 7, 3):                  # equal shards, ragged shards, fewer samples than 2 x world
    cond_all = torch.arange(total, dtype=torch.float32).view(total, 1, 1).expand(total, 5, 4) * 0.25
    lo, hi = shard_range(total, rank, world)
    img = sample_images(Model(), cond_all[lo:hi].contiguous(), S=4, seed=3, sample0=lo, total=total)
    ref = sample_images(Model(), cond_all, S=4, seed=3, sample0=0, total=total, gather=False)     # what ONE rank would produce
    assert img.shape == (total, 3, 4, 4) and torch.equal(img, ref), (rank, total)
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_sample_images_shard_logic_world_size_2_gloo(tmp_path):
    """The REAL pipeline.sample_images (contiguous shard, sample0 = global index of the shard's first sample, one all-gather
    with ragged shards padded) on two gloo ranks with a stub sampler / decoder: the joined batch equals the single-rank
    result, i.e. it is invariant to the number of ranks."""
    script = tmp_path / "worker2.py"
    script.write_text(WORKER2 % REPO)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29519", str(script)],
                         capture_output=True, text=True, env=env, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 2


def _bench_line(args, env_extra, timeout=300):
    env = dict(os.environ, FRIDO_BENCH_STUB="1", **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, capture_output=True, text=True, env=env,
                         timeout=timeout, cwd=REPO)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout          # ONE JSON line, printed by rank 0 only
    import json
    return json.loads(lines[0])


def test_bench_gpus_n_self_launches_one_rank_per_gpu():
    """`python bench.py --gpus 2 ...` typed WITHOUT torch.distributed.run (the driver's N = 1 spelling with a larger N) re-execs
    itself as a 2-rank job and prints one whole-job JSON line; FRIDO_BENCH_STUB swaps the HIP sampler for a CPU function of the
    global sample index so the launcher, shard arithmetic, barrier-bracketed timing and max-over-ranks run here over gloo."""
    line = _bench_line(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "5"], {})
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["stub"] is True
    assert line["config"]["global_batch"] == 10 and line["scaling"] == "weak" and len(line["per_rank_ms_per_step"]) == 2
    assert abs(line["value"] - 10 * 3 / (line["ms_per_step"] * 3e-3)) / line["value"] < 1e-3       # whole-job aggregate
    assert abs(line["ms_per_step"] - max(line["per_rank_ms_per_step"])) < 0.02                       # slowest rank sets the time


def test_bench_under_torchrun_as_the_driver_launches_it():
    """The driver's own N > 1 spelling: python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2."""
    env = dict(os.environ, FRIDO_BENCH_STUB="1", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29523", os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, env=env, timeout=300, cwd=REPO)
    assert out.returncode == 0, out.stdout + out.stderr
    assert sum(l.startswith("{") for l in out.stdout.splitlines()) == 1
    bad = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29524", os.path.join(REPO, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, env=env, timeout=300, cwd=REPO)
    assert bad.returncode != 0 and "WORLD_SIZE=2" in bad.stdout + bad.stderr      # a mismatch is an error message, not a traceback


def test_bench_n_gt_1_line_carries_the_config4_pass():
    """(r06) BASELINE config 4 is 256 images over 8 GPUs = 32 per GPU; the driver's scaling spelling runs the default 16 per GPU, so every
    N > 1 job adds one pass at 32 per GPU and reports it as `extra.config4_B32_per_gpu` (named in config.workload).  Stub job over gloo:
    the record's arithmetic and the opt-out."""
    line = _bench_line(["--gpus", "2", "--steps", "1", "--warmup", "0"], {})
    c4 = line["extra"]["config4_B32_per_gpu"]
    assert c4["per_gpu_batch"] == 32 and c4["global_batch"] == 64 and c4["n_gpus"] == 2 and c4["value"] > 0 and c4["unit"] == "images/s"
    assert abs(c4["value"] - 64 / (c4["ms_per_step"] * 1e-3)) / c4["value"] < 1e-2 and "config4_B32_per_gpu" in line["config"]["workload"]
    assert line["config"]["global_batch"] == 32                      # the headline workload is still 16 per GPU
    off = _bench_line(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-config4"], {})
    assert "extra" not in off
    other = _bench_line(["--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "5"], {})       # not the default batch: not config 4's job
    assert "extra" not in other


def test_bench_refuses_cleanly_when_the_node_has_fewer_devices_than_ranks():
    """r05: `python bench.py --gpus 2` on a node with fewer than 2 HIP devices (this container: none) prints ONE message and exits
    with a non-zero code before any process group or launcher exists -- no traceback, no hang at a rendezvous."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "FRIDO_BENCH_STUB"):
        env.pop(k, None)
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this node has two devices")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, env=env, timeout=300, cwd=REPO)
    assert out.returncode == 2, out.stdout + out.stderr
    msg = [l for l in (out.stdout + out.stderr).splitlines() if "--gpus 2 but this node exposes" in l]
    assert len(msg) == 1 and "Traceback" not in out.stderr
