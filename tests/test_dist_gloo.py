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
