"""Multi-GPU mode "replicated state / sharded replans" on the GPU: two ranks (two processes on this box's one GPU,
gloo underneath - RCCL refuses two ranks on one device) each replay a reference trace with the full replanning policy;
every rank plans half of each tick's searches, the results are exchanged, and BOTH ranks must match the reference's
recorded state at every tick - i.e. the N-rank result equals the 1-GPU result bit for bit."""
import json
import os
import re
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
from trafficsimulation_amd import dist as tdist
from trafficsimulation_amd import _capi as capi
from trafficsimulation_amd._lib import new_engine
from trafficsimulation_amd.world import load_trace
from tests.trace_util import setup_from_trace, check_initial, replay_and_compare, trace_path
rank, local, world = tdist.env_rank()
d = tdist.init("gloo", rank, world)
res = {}
for name in %(traces)r:
    tr = load_trace(trace_path(name))
    api = new_engine()
    setup_from_trace(api, tr, explicit_paths=False)
    check_initial(api, tr)
    if %(device_direct)r:
        import torch
        sr = tdist.ShardedReplans(device=torch.device("cuda", 0), device_direct=True).attach(api)
    else:
        sr = tdist.ShardedReplans().attach(api)
    n = replay_and_compare(api, tr)          # raises on the first tick that differs from the reference
    c = api.counters()
    res[name] = dict(ticks=n, astar_calls=int(c.astar_calls), want_calls=int(tr["astar_calls_spawn"]) + int(tr["astar_per_tick"].sum()),
                     exchanges=sr.calls, bytes=sr.bytes_sent, fp=list(api.rng_fingerprint(capi.RNG_GLOBAL)))
    api.close()
print(json.dumps(dict(rank=rank, res=res)), flush=True)
d.destroy_process_group()
'''


@pytest.mark.parametrize("world,device_direct,tight_pool,quads", [(2, False, False, False), (2, True, False, False), (2, True, True, False), (4, True, False, False),
                                                               (5, True, False, False), (2, True, False, True), (4, True, False, True)])
def test_ranks_sharded_replans_match_the_reference(world, device_direct, tight_pool, quads):
    """2, 4 and 5 ranks (with the test runner itself that is the six processes this box lets share its GPU), with the records exchanged between device buffers
    (ts_set_replan_sharding_device; the collective itself is gloo here) and once with the host-staged form."""
    # closed populations, then the agents that step on the host inside the shuffled order (traffic generator spawning and
    # planning mid-tick, service vehicles with their arrival records, rain): every rank runs those redundantly
    traces = ["full_64_s1", "full_96_s8", "faults_64_s9", "carve_96_s10", "dta_64_s12", "rain_96_s14", "config1_64_s11", "despawn_96_s25", "startgoal_96_s27"]
    if world > 2:
        traces = ["full_96_s8", "faults_64_s9", "dta_64_s12", "config1_64_s11", "startgoal_96_s27"]
    port = str(29541 + world + (10 if device_direct else 0) + (20 if tight_pool else 0) + (40 if quads else 0))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if quads:           # (every rank's share of every queue on k_replan_quad, its hand-backs on k_replan beside it)
        env["TS_QUAD"], env["TS_QUAD_MIN"] = "1", "1"
    if tight_pool:      # (almost no room reserved in the path pool: planners find it full, the pool is garbage-collected / grown INSIDE the
        env["TS_DEBUG_POOL_PER_ENTRY"] = "2"      # sharded tick, and the export must not rely on the pool's growth to size its buffer)
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(WORKER % dict(root=ROOT, traces=traces, device_direct=device_direct))
        path = f.name
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", port, path]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    os.unlink(path)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    rows = [json.loads(line) for line in out.stdout.splitlines() if line.startswith('{"rank"')]
    assert sorted(r["rank"] for r in rows) == list(range(world))
    rows = sorted(rows, key=lambda r: r["rank"])
    r0 = rows[0]
    for name in traces:
        a = r0["res"][name]
        for rb in rows[1:]:
            b = rb["res"][name]
            assert a["ticks"] == b["ticks"] > 0
            if name != "config1_64_s11":     # (service vehicles plan inside the generator's step too: not in the fixture's per-tick count)
                assert a["astar_calls"] == b["astar_calls"] == a["want_calls"]     # the searches of all ranks add up to the reference's
            assert a["astar_calls"] == b["astar_calls"]
            assert a["fp"] == b["fp"]
            assert a["exchanges"] == b["exchanges"] > 0
        assert sum(rb["res"][name]["bytes"] for rb in rows) > 0
        if world == 2:
            assert all(rb["res"][name]["bytes"] > 0 for rb in rows)       # both ranks planned something


NCCL_WORKER = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
import torch.distributed as dist
from trafficsimulation_amd import dist as tdist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
sr = tdist.ShardedReplans(device=torch.device("cuda", 0))
ok = True
for n in (0, 1, 4097, 3_000_000):
    payload = np.random.default_rng(n).integers(0, 256, size=n, dtype=np.uint8).tobytes()
    stacked, sizes = sr.gather_bytes(payload)
    ok &= sizes == [n] and stacked.shape == (1, max(n, 1)) and stacked[0, :n].tobytes() == payload
# the ctypes callback the engine calls (ts_exchange_fn), end to end
import ctypes as C
buf = (C.c_uint8 * 5)(1, 2, 3, 4, 5)
recv, sizes_p, stride = C.c_void_p(), C.c_void_p(), C.c_int64()
rc = sr._callback(None, C.addressof(buf), 5, C.pointer(recv), C.pointer(sizes_p), C.pointer(stride))
got = C.string_at(recv.value, 5)
ok &= rc == 0 and got == bytes([1, 2, 3, 4, 5]) and stride.value == 5 and C.cast(sizes_p.value, C.POINTER(C.c_int64))[0] == 5
print(json.dumps({"ok": bool(ok), "backend": dist.get_backend(), "calls": sr.calls}))
dist.destroy_process_group()
'''


def test_exchange_collective_over_rccl():
    """The same exchange object on a real RCCL process group (backend "nccl", buffers staged through the GPU).  One rank
    only - this box has one GPU and RCCL refuses two ranks on a device - so this checks the device staging, the padded
    all_gather and the callback plumbing, not the wire."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29543", HSA_ENABLE_IPC_MODE_LEGACY="0")
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(NCCL_WORKER % dict(root=ROOT))
        path = f.name
    out = subprocess.run([sys.executable, path], env=env, capture_output=True, text=True, timeout=600)
    os.unlink(path)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    row = json.loads([line for line in out.stdout.splitlines() if line.startswith("{")][-1])
    assert row["ok"] and row["backend"] == "nccl" and row["calls"] == 5



def test_bench_two_ranks_over_gloo_smoke():
    """`python bench.py --gpus 2` end to end on this box's one GPU (gloo, sharded replans, device-direct exchange): the line
    reports the process group's size and that both ranks ended in the same state."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--size", "512", "--vehicles", "12000",
           "--steps", "4", "--warmup", "4", "--no-cpu-baseline", "--no-secondary"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    row = json.loads([line for line in out.stdout.splitlines() if line.startswith("{")][-1])
    assert row["n_gpus"] == 2 and row["scaling"] == "strong"
    assert row["config"]["ranks_hold_identical_state"] is True
    assert row["config"]["exchange"]["calls"] > 0 and row["config"]["astar"]["calls"] > 1000
