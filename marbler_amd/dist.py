"""Multi-GPU: one process per GPU, envs sharded in contiguous blocks, no per-step exchange.

The path shards trivially (envs are independent); the only collectives
are a broadcast of the scenario parameter block from rank 0 at init and a gather of
per-env episode statistics per reporting interval -- both through torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for the world_size-2 tests).
"""
import os

import torch
import torch.distributed as dist

from .params import params_from_bytes, params_to_bytes


def init_from_env(backend=None, device_index=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).
    Returns (rank, world_size, local_rank).  world_size 1: no process group is created.
    device_index: the GPU of this rank if it is not LOCAL_RANK (rehearsals on a one-GPU box)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if device_index is not None:
        local = int(device_index)
    # RG_FORCE_PROCESS_GROUP=1: a process group even for one rank, so that a one-GPU box runs the RCCL code path of the
    # N > 1 job (parameter broadcast, statistics gather, barrier, max-reduction) end to end (tests/test_gpu_dist.py)
    force = os.environ.get("RG_FORCE_PROCESS_GROUP") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def collective_device(device):
    """Where tensors handed to a collective must live: the GPU for nccl (RCCL), the CPU for gloo."""
    if dist.is_initialized() and dist.get_backend() == "gloo":
        return torch.device("cpu")
    return torch.device(device)


def shard(total_envs, rank, world):
    """Contiguous block of envs owned by `rank`: (offset, count).  Remainders go to low ranks."""
    base, rem = divmod(total_envs, world)
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def broadcast_params(params, src=0, device=None):
    """Rank `src` sends its rg_scenario_params block (< 1 KB); every rank returns a copy of it."""
    if not dist.is_initialized():
        return params
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    device = collective_device(device)
    if dist.get_rank() == src:
        buf = torch.frombuffer(bytearray(params_to_bytes(params)), dtype=torch.uint8).to(device)
    else:
        import ctypes
        from ._lib import RgScenarioParams
        buf = torch.zeros(ctypes.sizeof(RgScenarioParams), dtype=torch.uint8, device=device)
    dist.broadcast(buf, src=src)
    return params_from_bytes(buf.cpu().numpy().tobytes())


def gather_episode_stats(done_return_sum, done_count, done_steps_sum, dst=0, total_envs=None):
    """Gathers the per-env [E_local] statistics of every rank to `dst`: returns as float32, episode counts and
    step totals as int64 (exact -- never through a float).  ONE fixed-size collective per call (shards are equal up to one
    env; shorter shards are padded): the float32 returns travel as their bit patterns in a third int64 row.
    total_envs: the job's env count when the shards are `shard(total_envs, rank, world)` (every caller in this repo): the
    shard sizes then follow from arithmetic and the call makes no host synchronisation besides the collective itself;
    without it one 8-byte all_gather of the sizes precedes the gather.  A reporting-interval call -- per time step it
    would serialise the ranks; nothing in the step path calls it.
    Returns (returns [E_total] f32, counts i64, steps i64) on `dst`, None elsewhere.  With no process group: the inputs.
    RCCL (nccl backend) has a true gather to one root; gloo on CPU does too."""
    if not dist.is_initialized():
        return done_return_sum, done_count.to(torch.int64), done_steps_sum.to(torch.int64)
    world, rank = dist.get_world_size(), dist.get_rank()
    cdev = collective_device(done_return_sum.device)
    k = done_return_sum.numel()
    if total_envs is not None:
        sizes = [shard(int(total_envs), r, world)[1] for r in range(world)]
        if sizes[rank] != k:
            raise ValueError(f"rank {rank} holds {k} envs, shard({total_envs}, {rank}, {world}) says {sizes[rank]}")
    else:
        n = torch.tensor([k], dtype=torch.int64, device=cdev)
        every = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(every, n)                     # 8 bytes per rank; every rank needs the padded size
        sizes = torch.cat(every).tolist()
    m = max(sizes)
    buf = torch.zeros(3, m, dtype=torch.int64, device=cdev)
    buf[0, :k] = done_return_sum.to(device=cdev, dtype=torch.float32).contiguous().view(torch.int32)
    buf[1, :k] = done_count.to(cdev)
    buf[2, :k] = done_steps_sum.to(cdev)
    got = [torch.zeros_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, got, dst=dst)
    if rank != dst:
        return None
    ret = torch.cat([t[0, :c] for t, c in zip(got, sizes)]).to(torch.int32).view(torch.float32)
    cnt = torch.cat([t[1, :c] for t, c in zip(got, sizes)])
    stp = torch.cat([t[2, :c] for t, c in zip(got, sizes)])
    return ret, cnt, stp
