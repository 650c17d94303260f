"""One process per GPU.  The hot path has no exchange step in the bit-exact mode ("replicas only",
DESIGN.md §6): every rank simulates an independent world; the only collectives are the barrier and the
two reductions of the measurement (max time, summed agent-steps)."""
from __future__ import annotations

import os


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def replica_seed(base_seed: int, rank: int) -> int:
    """Distinct, reproducible seed per replica."""
    return int(base_seed) + 1000 * int(rank)


def init(backend: str, rank: int, world: int, device=None):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    kw = {}
    if device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def aggregate(elapsed_s: float, agent_steps: int, world: int, device="cpu"):
    """(max over ranks of the time, sum over ranks of the agent-steps)."""
    if world == 1:
        return float(elapsed_s), int(agent_steps)
    import torch
    import torch.distributed as dist
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    s = torch.tensor([agent_steps], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return float(t.item()), int(s.item())
