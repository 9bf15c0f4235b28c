"""Row (e) on hardware: the RCCL code path of the N > 1 job -- process-group rendezvous over 127.0.0.1, parameter-block
broadcast from rank 0, barrier, max-reduction of the elapsed time, gather of the episode statistics -- run end to end by
bench.py under torch.distributed backend "nccl" (= RCCL on ROCm).  A one-GPU box can host one RCCL rank only (two ranks
on one device are refused as duplicates), so this is world size 1 with RG_FORCE_PROCESS_GROUP=1; world size 2 is covered
on CPU with gloo (tests/test_host.py) and rehearsed on this GPU with `bench.py --gpus 2 --dist-backend gloo --share-gpu`."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_bench_runs_its_collectives_over_rccl():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               RG_FORCE_PROCESS_GROUP="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "120", "--warmup", "20",
                        "--no-cpu-baseline", "--no-saturated", "--dist-backend", "nccl"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 1e8 and d["scaling"] == "weak"
    assert d["collective"]["backend"] == "nccl" and d["collective"]["world_size"] == 1 and d["collective"]["ranks_seen"] == [0]
    assert "RCCL" in d["collective"]["library"]
    assert d["episodes"]["finished"] > 0 and 0 < d["episodes"]["mean_length"] <= 81      # gathered over RCCL
    assert d["config"]["agents"] == 5                                                    # the broadcast parameter block


def test_two_ranks_share_the_gpu_through_the_launcher_gloo():
    """`python bench.py --gpus 2` starts its own two ranks; with --share-gpu both step their shard on cuda:0 and the
    collectives run on gloo: the whole N = 2 job with real kernels, minus RCCL's device-to-device transport."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--share-gpu",
                        "--steps", "100", "--warmup", "10"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["collective"]["ranks_seen"] == [0, 1] and d["value"] > 1e8
    assert d["config"]["parallelism"] == "env-sharded x2" and d["episodes"]["finished"] > 0


def test_four_ranks_share_the_gpu_with_the_drivers_flags():
    """The driver's multi-GPU command line (`--gpus N --steps 20 --warmup 5`, self-launched ranks, barrier + max over ranks, one
    timing per rank) with real kernels on every rank: four ranks on cuda:0 (with this process that is five on the card; the box
    admits six), gloo for the collectives.  What a one-GPU box cannot show is RCCL's transport and the 8-rank job itself (eight
    ranks with no GPU work: tests/test_host.py)."""
    env = dict(os.environ, RG_BENCH_NO_LIVE_COUNTERS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dist-backend", "gloo", "--share-gpu",
                        "--steps", "20", "--warmup", "5"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["steps"] == 20 and d["warmup"] == 5 and d["collective"]["ranks_seen"] == [0, 1, 2, 3]
    assert len(d["per_rank_ms_per_step"]) == 4 and len(d["per_rank_kernel_ms_per_step"]) == 4 and all(v > 0 for v in d["per_rank_ms_per_step"])
    assert d["ms_per_step"] == pytest.approx(max(d["per_rank_ms_per_step"]), rel=1e-3)      # the value's clock is the slowest rank's
    assert d["config"]["parallelism"] == "env-sharded x4" and "spin-up" in d["config"]["workload"]
    assert d["value"] == pytest.approx(4 * 4096 * 5 / (d["ms_per_step"] * 1e-3), rel=1e-6)
