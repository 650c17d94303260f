"""The N>1 path on CPU: world_size-2 gloo run of the replica scheme bench.py uses (one process per device,
independent worlds with per-rank seeds, barrier + max-time / sum-steps reductions).  The engine behind each
rank is the CPU oracle here; on the GPU box the same code runs over the HIP engine with backend nccl."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys, time
sys.path.insert(0, %(root)r)
import numpy as np
from trafficsimulation_amd import dist as tdist
import bench
from oracle import pyoracle
rank, local, world = tdist.env_rank()
d = tdist.init("gloo", rank, world)
seed = tdist.replica_seed(7, rank)
tables, routes, _ = bench.make_workload(160, 800, seed)
api = pyoracle.load()
bench.setup(api, tables, routes, seed)
d.barrier()
t0 = time.perf_counter()
api.step(12)
dt = time.perf_counter() - t0
d.barrier()
steps = api.counters().agent_steps
tmax, total = tdist.aggregate(dt, steps, world)
fp = api.rng_fingerprint(0)
print(json.dumps(dict(rank=rank, seed=seed, steps=steps, total=total, tmax=tmax, dt=dt, fp=list(fp))), flush=True)
d.destroy_process_group()
'''


def test_two_rank_replicas_aggregate():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", "-c", WORKER % dict(root=ROOT)]
    # torch.distributed.run needs a script path, not -c: write the worker to a temp file
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(WORKER % dict(root=ROOT))
        path = f.name
    cmd = cmd[:-2] + [path]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    os.unlink(path)
    assert out.returncode == 0, out.stderr[-2000:]
    import re
    rows = [json.loads(m) for m in re.findall(r"\{[^{}]*\}", out.stdout)]
    assert sorted(r["rank"] for r in rows) == [0, 1]
    r0, r1 = sorted(rows, key=lambda r: r["rank"])
    assert r0["seed"] != r1["seed"] and r0["fp"] != r1["fp"]          # independent replicas
    assert r0["total"] == r1["total"] == r0["steps"] + r1["steps"]    # whole-job aggregate
    assert r0["tmax"] == r1["tmax"] >= max(r0["dt"], r1["dt"]) - 1e-9  # max over ranks


EXCHANGE_WORKER = r'''
import ctypes as C, json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
from trafficsimulation_amd import dist as tdist
from trafficsimulation_amd import _capi as capi
rank, local, world = tdist.env_rank()
d = tdist.init("gloo", rank, world)
sr = tdist.ShardedReplans()
out = []
# the engine's side of ts_exchange_fn, called through the same ctypes function type the HIP library is given:
# variable-size buffers (one of them empty), twice in a row (the receive buffer is reused per call)
cb = capi.EXCHANGE_FN(sr._callback)
for rnd in range(3):
    n = (0 if (rank + rnd) %% 3 == 0 else 1000 * (rank + 1) + 17 * rnd)
    payload = (np.arange(n, dtype=np.uint32) * 2654435761 + rank + 7 * rnd).astype(np.uint8).tobytes()
    send = C.create_string_buffer(payload, len(payload)) if payload else None
    recv, sizes, stride = C.c_void_p(), C.c_void_p(), C.c_int64()
    rc = cb(None, C.cast(send, C.c_void_p) if send else None, len(payload), C.byref(recv), C.byref(sizes), C.byref(stride))
    assert rc == 0
    sz = (C.c_int64 * world).from_address(sizes.value)
    got = []
    for r in range(world):
        n_r = int(sz[r])
        data = C.string_at(recv.value + r * stride.value, n_r) if n_r else b""
        want_n = (0 if (r + rnd) %% 3 == 0 else 1000 * (r + 1) + 17 * rnd)
        want = (np.arange(want_n, dtype=np.uint32) * 2654435761 + r + 7 * rnd).astype(np.uint8).tobytes()
        got.append(data == want and n_r == want_n)
    out.append(all(got))
print(json.dumps(dict(rank=rank, ok=out, calls=sr.calls)), flush=True)
d.destroy_process_group()
'''


def test_sharded_replan_exchange_two_ranks_gloo():
    """The exchange step of the replicated-state multi-GPU mode (ts_exchange_fn over torch.distributed): world-size 2 on
    gloo, driven through the ctypes callback type the engine is given - variable sizes, an empty buffer, repeated calls.
    (The engine side of the mode - sharded searches, export / import, state equal to the one-rank run bit for bit - needs
    a GPU: tests/test_gpu_dist.py.)"""
    import re
    import tempfile
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29537")
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(EXCHANGE_WORKER % dict(root=ROOT))
        path = f.name
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29537", path]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    os.unlink(path)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = [json.loads(m) for m in re.findall(r"\{[^{}]*\}", out.stdout)]
    assert sorted(r["rank"] for r in rows) == [0, 1]
    for r in rows:
        assert r["ok"] == [True, True, True] and r["calls"] == 3
