"""`python bench.py --gpus N` must start N ranks by itself (the reference starts one process per GPU,
main_pretrain_mpirun.py:17-49).  `--dry-launch` runs the launcher, the rendezvous, the barrier and the max-over-ranks
timing reduction over gloo on the CPU with an empty timed region, so this runs without a GPU."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _run(argv, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True,
                          timeout=timeout)


@pytest.mark.timeout(300)
def test_gpus_2_starts_two_ranks_and_prints_one_line():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-launch"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]      # gloo prints its own connection notes
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["rccl"]["world"] == 2 and out["rccl"]["allreduce_of_ones"] == 2
    assert out["rccl"]["seeds"] == [1234, 1235]                 # distinct shard seeds
    assert out["rccl"]["launcher"] == "bench.py"
    assert out["elapsed_s"] >= 0.02                             # max over ranks: rank 1 sleeps 20 ms


@pytest.mark.timeout(300)
def test_world_size_mismatch_is_refused():
    r = _run(["--gpus", "2", "--dry-launch"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


@pytest.mark.timeout(300)
def test_a_failing_rank_fails_the_launch():
    # a per-rank crash must surface as a non-zero exit of the parent, not a hang
    r = _run(["--gpus", "2", "--dry-launch"], {"PIXPRO_TEST_FAIL_RANK": "1"})
    assert r.returncode != 0


@pytest.mark.timeout(300)
def test_too_few_gpus_is_refused():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])   # no GPU here and no rehearsal flag
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr      # every rank checks for itself; the parent asks nothing


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_two_rank_rehearsal_runs_the_real_step(dev):
    """The N > 1 control flow of the REAL bench step on the one-GPU box the driver uses: two ranks share
    the GPU and talk over gloo (PIXPRO_REHEARSAL=1) - DDP gradient buckets, the fused SyncBN statistics exchange, shard
    seeds, barrier + max-over-ranks timing, and the SyncBN A/B that a multi-GPU run appends.  Never a measurement."""
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8", "--arch", "resnet18", "--no-cpu-baseline"],
             {"PIXPRO_REHEARSAL": "1"}, timeout=850)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["config"]["global_batch"] == 16
    assert out["rccl"]["world"] == 2 and out["rccl"]["backend"] == "gloo" and out["rccl"]["allreduce_of_ones"] == 2
    assert "rehearsal" in out["config"] and out["value"] > 0
    ab = out["rccl"]["syncbn_ab"]
    assert ab["default"] > 0 and ab["bn_view_batch"] > 0 and "key_bn_group+second_stream" not in ab
    assert out["hot_kernels"]["ppm_loss"]["ms_per_launch"] > 0
