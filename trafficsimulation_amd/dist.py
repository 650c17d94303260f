"""One process per GPU (torch.distributed: backend "nccl" = RCCL over xGMI on the GPU box, "gloo" on CPU).

Two multi-GPU modes, both bit-exact against the sequential reference:
  * replicas: every rank simulates an independent world; the only collectives are the barrier and the two reductions
    of the measurement (max time, summed agent-steps) - `aggregate`;
  * replicated state / sharded replans: every rank holds the SAME world and is stepped with the same calls; each tick's
    replanning searches (99 % of a default-policy tick) are split over the ranks and the results are all-gathered -
    `ShardedReplans` below is the exchange step the engine calls once per tick (`ts_set_replan_sharding`)."""
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


class ShardedReplans:
    """The all-gather the engine calls once per tick in the replicated-state mode.

    `ts_exchange_fn` contract (include/trafficsim.h): variable-size byte buffers in host memory, one per rank, gathered
    into `world` slots of equal stride.  Over torch.distributed that is an all_gather of the sizes followed by an
    all_gather of the buffers padded to the largest; with backend "nccl" the buffers are staged through the rank's GPU
    (RCCL moves them over xGMI), with "gloo" they stay on the host.

        sr = ShardedReplans(group=None)          # after init_process_group
        sr.attach(api)                           # api: _capi.CApi bound to the HIP library, one per rank
        api.step(n)                              # every rank the same calls
    """

    def __init__(self, group=None, device=None, all_gather=None, rank=None, world=None, device_direct=False):
        self.group = group
        self.device = device
        # device_direct: the engine hands over and takes back DEVICE pointers (ts_set_replan_sharding_device): the records
        # stay in HBM, one all_gather_into_tensor between pre-sized device buffers moves them (RCCL over xGMI with backend
        # "nccl"; with "gloo" the collective itself is staged through the host, the engine's side is the same)
        self.device_direct = bool(device_direct)
        self._send_t = None
        self._recv_t = None
        self.bytes_sent = 0
        self.calls = 0
        self._keep = None
        self._sizes = None
        self._cb = None
        if all_gather is not None:      # injected collective (tests): all_gather(list_out, tensor_in)
            self._all_gather = all_gather
            self.rank, self.world = int(rank), int(world)
        else:
            import torch.distributed as dist
            self._all_gather = lambda out, t: dist.all_gather(out, t, group=self.group)
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    # the collective on numpy byte arrays: returns (stacked (world, stride) uint8 array, sizes)
    def gather_bytes(self, payload):
        import numpy as np
        import torch
        dev = self.device if self.device is not None else "cpu"
        n = torch.tensor([len(payload)], dtype=torch.int64, device=dev)
        sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(self.world)]
        self._all_gather(sizes, n)
        sizes = [int(x.item()) for x in sizes]
        stride = max(max(sizes), 1)
        buf = torch.zeros(stride, dtype=torch.uint8)
        if len(payload):
            buf[:len(payload)] = torch.from_numpy(np.frombuffer(payload, dtype=np.uint8).copy())
        buf = buf.to(dev)
        out = [torch.empty(stride, dtype=torch.uint8, device=dev) for _ in range(self.world)]
        self._all_gather(out, buf)
        stacked = torch.stack(out).cpu().contiguous().numpy()
        self.bytes_sent += len(payload)
        self.calls += 1
        return stacked, sizes

    def _callback(self, user, send, nbytes, recv_pp, sizes_pp, stride_p):
        import ctypes as C
        import numpy as np
        try:
            payload = C.string_at(send, nbytes) if nbytes > 0 else b""
            stacked, sizes = self.gather_bytes(payload)
            self._keep = stacked                                # owned by us until the next call
            self._sizes = (C.c_int64 * self.world)(*sizes)
            recv_pp[0] = stacked.ctypes.data
            sizes_pp[0] = C.cast(self._sizes, C.c_void_p).value
            stride_p[0] = stacked.shape[1]
            return 0
        except Exception as ex:      # never let an exception cross the C boundary
            import sys
            print(f"[ShardedReplans] exchange failed on rank {self.rank}: {ex!r}", file=sys.stderr, flush=True)
            return -1

    # ---- device-direct form ------------------------------------------------------------------------------------------
    @staticmethod
    def _wrap_device(ptr, nbytes, device):
        """A uint8 torch tensor over `nbytes` of device memory at `ptr` (no copy; the engine owns the memory)."""
        import torch

        class _Raw:
            __cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
        return torch.as_tensor(_Raw(), device=device)

    def gather_device(self, send_ptr, nbytes):
        """all-gather of this rank's `nbytes` at device pointer `send_ptr`: -> (device tensor of world * stride bytes, sizes, stride)."""
        import torch
        import torch.distributed as dist
        dev = self.device if self.device is not None else torch.device("cuda", torch.cuda.current_device())
        nccl = self._backend() == "nccl"
        cdev = dev if nccl else "cpu"
        n = torch.tensor([int(nbytes)], dtype=torch.int64, device=cdev)
        sizes = [torch.zeros(1, dtype=torch.int64, device=cdev) for _ in range(self.world)]
        self._all_gather(sizes, n)
        sizes = [int(x.item()) for x in sizes]
        stride = (max(max(sizes), 1) + 255) // 256 * 256
        if self._send_t is None or self._send_t.numel() < stride:      # pre-sized, grown geometrically, reused every tick
            cap = max(stride * 2, 1 << 16)
            self._send_t = torch.empty(cap, dtype=torch.uint8, device=dev)
            self._recv_t = torch.empty(cap * self.world, dtype=torch.uint8, device=dev)
        send_t = self._send_t[:stride]
        if nbytes > 0:
            send_t[:nbytes].copy_(self._wrap_device(send_ptr, nbytes, dev))
        recv_t = self._recv_t[:stride * self.world]
        if nccl:
            dist.all_gather_into_tensor(recv_t, send_t, group=self.group)
        else:      # gloo: the collective on host copies, the result back in device memory
            host = send_t.cpu()
            outs = [torch.empty(stride, dtype=torch.uint8) for _ in range(self.world)]
            self._all_gather(outs, host)
            recv_t.copy_(torch.cat(outs))
        torch.cuda.synchronize(dev)          # the engine's stream reads the slots next
        self.bytes_sent += int(nbytes)
        self.calls += 1
        return recv_t, sizes, stride

    def _backend(self):
        try:
            import torch.distributed as dist
            return dist.get_backend(self.group)
        except Exception:
            return "injected"

    def _callback_dev(self, user, send, nbytes, recv_pp, sizes_pp, stride_p):
        import ctypes as C
        try:
            recv_t, sizes, stride = self.gather_device(send, nbytes)
            self._keep = recv_t                                  # owned by us until the next call
            self._sizes = (C.c_int64 * self.world)(*sizes)
            recv_pp[0] = recv_t.data_ptr()
            sizes_pp[0] = C.cast(self._sizes, C.c_void_p).value
            stride_p[0] = stride
            return 0
        except Exception as ex:      # never let an exception cross the C boundary
            import sys
            print(f"[ShardedReplans] device exchange failed on rank {self.rank}: {ex!r}", file=sys.stderr, flush=True)
            return -1

    def attach(self, api):
        """Switch `api`'s engine to sharded replans over this group (world 1 = plain single-GPU stepping)."""
        from . import _capi as capi
        if self.world == 1:
            api.set_replan_sharding(0, 1, None)
            return self
        self._cb = capi.EXCHANGE_FN(self._callback_dev if self.device_direct else self._callback)
        api.set_replan_sharding(self.rank, self.world, self._cb, device_buffers=self.device_direct)
        return self
