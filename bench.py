#!/usr/bin/env python3
"""bench.py - agent-steps/s of the per-tick agent-update path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 4096] [--vehicles 1000000] [--policy full]

A "step" is one CityModel.step() tick over the synthetic workload.  At N=1 the workload is the configuration
BASELINE.json's metric is quoted on - a 4096x4096 procedurally generated city with 10^6 vehicles - under the
reference's DEFAULT policy (`--policy full`: QUEUE_ACTUATED light groups, replanning with the GPU A*, contraflow
overtakes / detours, malfunctions and sideswipes; config.py untouched apart from rain and spawning being off so that the
population is closed).  The reduced policies of BASELINE configs 2 / 3-without-searches are kept as
`--policy config2` / `--policy lights` and are reported, for the same world, as `secondary` fields of the N=1 line.

N>1: one process per GPU.  `python bench.py --gpus N` without a torchrun environment launches its own N ranks
(torch.distributed.run, 127.0.0.1) before anything touches a GPU; under the driver's torchrun it reads
RANK / LOCAL_RANK / WORLD_SIZE.  Two bit-exact multi-GPU modes:
  --mode sharded  (default for --policy full)  ONE world, replicated on every rank; each tick's replanning searches are
                  split over the ranks and their results all-gathered over RCCL: total work fixed -> "scaling": "strong";
  --mode replicas (default otherwise)          independent worlds of the same size: "scaling": "weak".
`n_gpus` is the world size the process group reports, and it must equal --gpus.

Prints ONE JSON line on rank 0 with the contract fields plus
  "roofline":     the dominant kernel of THIS run: algorithmic bytes (for k_decide_replan SURVEY.md §8(d)'s
                  50 B per expansion + 25 B per relaxation) / HIP-event time, against 8 TB/s HBM
  "cpu_baseline": the CPU oracle (single-threaded C++ restatement, kind "port") on a bounded sample of the same workload,
                  timed on this box's host cores in a child process while the GPU leg runs (rank 0, N=1 only).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# Algorithmic bytes per agent-step, SURVEY.md §8(d): 80 B = fixed state 20 + path window 4 + map gathers 32 + writes 24,
# split over the kernels that do that work.  k_move_claim is pure parallelisation overhead.
ALGO_BYTES = {"k_decide_pre": 10, "k_decide_main": 44, "k_move_claim": 0, "k_move_resolve": 26,
              "k_compact": 0, "k_apply_event": 0, "k_rng": 0, "k_density": 0, "k_reach_strict": 0}
ASTAR_BYTES_PER_EXPANSION, ASTAR_BYTES_PER_RELAXATION = 50, 25

# "config 2" policy of BASELINE.md: car-following + movement kernels only.  Lights disabled, replans gated off;
# malfunction / sideswipe chances 0 (their draws are still consumed).
POLICY = {
    "TRAFFIC_LIGHT_AGENT_ALGORITHM": "DISABLED", "PATHFINDING_COOLDOWN": 10 ** 9,
    "VEHICLE_STUCK_RECOMPUTE_THRESHOLD": 10 ** 9, "VEHICLE_STUCK_RECOMPUTE_THRESHOLD_INTERSECTION": 10 ** 9,
    "VEHICLE_CONTRAFLOW_OVERTAKE_ACTIVE": False, "VEHICLE_STUCK_CONTRAFLOW_ENABLED": False,
    "VEHICLE_MALFUNCTION_CHANCE": 0.0, "VEHICLE_SIDESWIPE_COLLISION_CHANCE": 0.0, "RAIN_ENABLED": False,
}
FULL_POLICY = {"RAIN_ENABLED": False}   # everything else at the reference's defaults (config.py)
# BASELINE config 3's kernels without the searches: queue-actuated light groups stepping inside the shuffled order
LIGHTS_POLICY = dict(POLICY, TRAFFIC_LIGHT_AGENT_ALGORITHM="QUEUE_ACTUATED")
POLICY_TEXT = {
    "config2": "config-2 policy (reduced): car-following + movement kernels, lights DISABLED, replans gated off, malfunction / "
               "sideswipe chance 0",
    "lights": "lights policy (reduced): config 2 plus QUEUE_ACTUATED light groups (phase change, intersection clearance) "
              "stepping in the shuffled order; replans gated off",
    "full": "full policy = the reference's defaults: QUEUE_ACTUATED lights, replanning (GPU A*, phases 0-4), contraflow, "
            "malfunctions / sideswipes; rain and spawning off (closed population)",
}


def make_workload(size, vehicles, seed, world="synthetic", carves=False):
    from trafficsimulation_amd import citygen
    t0 = time.time()
    if world == "reference":    # the reference's own city for (size, seed): exact but interpreted, minutes at 4096^2
        from trafficsimulation_amd import worldgen
        tables = worldgen.generate_world(size, size, seed=seed, rain_enabled=False, enable_traffic=False,
                                         carve_subblock_roads=bool(carves))
    else:       # (carves: BASELINE config 5's "sub-block roads + L-shaped carves" on the look-alike, citygen._carve_subblock_roads)
        tables = citygen.generate(size, size, seed=seed, carve_subblock_roads=bool(carves))
    t1 = time.time()
    lo, hi = (150, 400) if size >= 1024 else (30, 120)
    routes = citygen.make_routes(tables, vehicles, seed=seed + 1, min_len=lo, max_len=hi)
    return tables, routes, (t1 - t0, time.time() - t1)


def setup(api, tables, routes, seed, extra=None, policy="config2"):
    from trafficsimulation_amd import _capi as capi
    from trafficsimulation_amd.world import build_engine
    d = dict(POLICY if policy == "config2" else LIGHTS_POLICY if policy == "lights" else FULL_POLICY)
    d.update(extra or {})
    p = api.params_from_defaults(d)
    if extra and "eager_density" in extra:
        p.eager_density = int(extra["eager_density"])
    build_engine(api, tables, params=p, global_seed=seed, sched_seed=seed)
    s, g, off, dirs = routes
    api.add_vehicles_dirs(s, g, np.full(len(s), capi.POP["through"], np.int32), off, dirs)
    return api


def cpu_leg(args):
    """Child process: the CPU oracle on the same workload, tick after tick, progress appended to --cpu-out after every
    tick; it stops once --cpu-seconds of stepping are used up (checked between ticks; the parent ends it if one tick
    overruns)."""
    from oracle import pyoracle
    tables, routes, _ = make_workload(args.size, args.vehicles, args.seed, args.world, args.carves)
    cpu = pyoracle.load()
    ts = time.time()
    setup(cpu, tables, routes, args.seed, extra={"eager_density": 1}, policy=args.policy)
    setup_s = time.time() - ts
    v0 = cpu.num_vehicles()
    t0 = time.perf_counter()
    n = 0
    with open(args.cpu_out, "w") as f:
        f.write(json.dumps(dict(ticks=0, seconds=0.0, agent_steps=0, setup_s=setup_s, vehicles=v0, astar_expansions=0, astar_calls=0)) + "\n")
        f.flush()
        while time.perf_counter() - t0 < args.cpu_seconds and n < args.cpu_max_ticks:
            cpu.step(1)
            n += 1
            c = cpu.counters()
            f.write(json.dumps(dict(ticks=n, seconds=time.perf_counter() - t0, agent_steps=int(c.agent_steps), setup_s=setup_s,
                                    vehicles=v0, astar_expansions=int(c.astar_expansions), astar_calls=int(c.astar_calls))) + "\n")
            f.flush()
    cpu.close()


def run_policy(api, tables, routes, seed, policy, steps, warmup, barrier=None):
    """One timed leg on an engine that is already created: W untimed ticks, then exactly K timed ones."""
    import torch
    setup(api, tables, routes, seed, policy=policy)
    v0 = api.num_vehicles()
    t_w = time.perf_counter()
    api.step(warmup)
    warm_s = time.perf_counter() - t_w
    api.profile_enable(True)
    c0 = api.counters()
    if barrier:
        barrier()
    t0 = time.perf_counter()
    api.step(steps)   # ts_step returns only after the engine's stream has drained
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if barrier:
        barrier()
    c1 = api.counters()
    return dict(v0=v0, elapsed=t1 - t0, c0=c0, c1=c1, prof=api.profile(), live_end=api.num_vehicles(), warm_s=warm_s)


def launch_ranks(args):
    """`python bench.py --gpus N` outside torchrun: start the N ranks ourselves, before this process touches a GPU."""
    port = 29500 + (os.getpid() % 400)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--vehicles", type=int, default=1_000_000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--policy", choices=["config2", "lights", "full"], default="full",
                    help="full = the reference's defaults (the headline); config2 / lights = the reduced policies of BASELINE "
                         "configs 2 / 3-without-searches")
    ap.add_argument("--mode", choices=["sharded", "replicas"], default=None,
                    help="multi-GPU mode (N>1): sharded = one world, replans split over the ranks (default for --policy full); "
                         "replicas = independent worlds")
    ap.add_argument("--carves", action="store_true", help="L-shaped one-lane roads inside the large blocks (BASELINE config 5's world)")
    ap.add_argument("--world", choices=["synthetic", "reference"], default="synthetic",
                    help="synthetic = citygen look-alike (fast to build); reference = worldgen, the reference's city for the seed")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend for N>1: nccl = RCCL over xGMI (one GPU per rank); gloo = host collectives, for "
                         "rehearsing the N>1 path with several ranks on one GPU (local rank r uses device r %% visible devices)")
    ap.add_argument("--exchange", choices=["device", "host"], default="device",
                    help="sharded mode: records exchanged between device buffers (default) or staged through host memory")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the reduced-policy legs of the N=1 line")
    ap.add_argument("--cpu-seconds", type=float, default=25.0)
    ap.add_argument("--cpu-max-ticks", type=int, default=1000)
    ap.add_argument("--cpu-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-out", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_only:
        return cpu_leg(args)

    env_world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if args.gpus > 1 and env_world == 0:
        raise SystemExit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = max(env_world, 1)
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    mode = args.mode or ("sharded" if args.policy == "full" else "replicas")

    # the CPU leg runs beside the GPU leg on otherwise idle host cores (N=1 only)
    cpu_proc, cpu_out = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_out = tempfile.NamedTemporaryFile("w", suffix=".jsonl", delete=False).name
        cpu_proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-only", "--cpu-out", cpu_out,
                                     "--size", str(args.size), "--vehicles", str(args.vehicles), "--seed", str(args.seed),
                                     "--policy", args.policy, "--world", args.world, "--cpu-seconds", str(args.cpu_seconds),
                                     "--cpu-max-ticks", str(args.cpu_max_ticks)] + (["--carves"] if args.carves else []),
                                    env=dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES=""))

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    if args.backend == "gloo":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    coll_dev = torch.device("cuda", local_rank) if args.backend == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")
    n_gpus = dist.get_world_size() if world > 1 else 1

    from trafficsimulation_amd._lib import new_engine
    from trafficsimulation_amd import dist as tdist
    from trafficsimulation_amd import _capi as capi_mod
    seed = args.seed + (1000 * rank if (world > 1 and mode == "replicas") else 0)
    tables, routes, gen_t = make_workload(args.size, args.vehicles, seed, args.world, args.carves)
    _rl = np.diff(np.asarray(routes[2], dtype=np.int64))      # cells per initial route (routes = starts, goals, offsets, directions)
    route_stats = ({"min": int(_rl.min()), "median": float(np.median(_rl)), "mean": float(_rl.mean()), "max": int(_rl.max()),
                    "kind": "random walks along the flow; the goal is where the walk ends"} if len(_rl) else None)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    api = new_engine()
    api.set_device(local_rank)
    sharder = None
    setup(api, tables, routes, seed, policy=args.policy)
    if world > 1 and mode == "sharded":
        # (the records stay in device memory: one all_gather_into_tensor over RCCL; --exchange host = the host-staged form)
        sharder = tdist.ShardedReplans(device=torch.device("cuda", local_rank) if (args.exchange == "device" or args.backend == "nccl") else None,
                                       device_direct=args.exchange == "device").attach(api)
    v0 = api.num_vehicles()
    t_w = time.perf_counter()
    api.step(args.warmup)
    warm_s = time.perf_counter() - t_w
    api.profile_enable(True)
    c0 = api.counters()
    barrier()
    t0 = time.perf_counter()
    tick_s, tick_c = [], [c0]
    for _ in range(args.steps):   # ts_step returns only after the engine's stream has drained: one call per tick, to see the spread
        ts = time.perf_counter()
        api.step(1)
        tick_s.append(time.perf_counter() - ts)
        tick_c.append(api.counters())     # (a 300-byte read of the counter block: inside the timed region, it only adds to it)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    barrier()
    c1 = api.counters()
    elapsed = t1 - t0
    steps_done = c1.agent_steps - c0.agent_steps
    rounds = c1.move_rounds - c0.move_rounds
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        if mode == "replicas":   # independent worlds: the job's agent-steps are the sum; sharded: every rank stepped the same world
            ss = torch.tensor([steps_done], dtype=torch.int64, device=coll_dev)
            dist.all_reduce(ss, op=dist.ReduceOp.SUM)
            steps_done = int(ss.item())
    prof = api.profile()
    live_end = api.num_vehicles()
    # sharded mode: every rank must hold the same world after the run (both RNG streams, the counters the searches feed)
    same_state = None
    if world > 1 and mode == "sharded":
        mine = [list(api.rng_fingerprint(capi_mod.RNG_GLOBAL)), list(api.rng_fingerprint(capi_mod.RNG_SCHEDULER)), int(c1.agent_steps),
                int(c1.astar_calls), int(c1.stuck), int(c1.count_completed_through), int(live_end)]
        every = [None] * world
        dist.all_gather_object(every, mine)
        same_state = all(x == every[0] for x in every)
    api.close()

    out = None
    if rank == 0:
        total_ms = {k: v[0] for k, v in prof.items()}
        exp = c1.astar_expansions - c0.astar_expansions
        rel = c1.astar_relaxations - c0.astar_relaxations
        calls = c1.astar_calls - c0.astar_calls
        veh_steps_local = c1.agent_steps - c0.agent_steps
        kernel_keys = [k for k in total_ms if not k.startswith("host_")]
        dom = max(kernel_keys, key=lambda k: total_ms[k])
        ms, launches, items = prof[dom]
        if dom == "k_decide_replan":
            # searches run on every rank's share; the counters of a sharded run are the whole world's, so this rank's share
            # of the algorithmic bytes is 1 / world of them
            algo_bytes_total = (ASTAR_BYTES_PER_EXPANSION * exp + ASTAR_BYTES_PER_RELAXATION * rel) / (world if mode == "sharded" else 1)
            unit_note = {"expansions": exp, "relaxations": rel, "bytes_per_expansion": ASTAR_BYTES_PER_EXPANSION,
                         "bytes_per_relaxation": ASTAR_BYTES_PER_RELAXATION,
                         "expansions_per_s_in_kernel": exp / (world if mode == "sharded" else 1) / max(ms / 1e3, 1e-12)}
        else:
            algo_bytes_total = ALGO_BYTES.get(dom, 0) * veh_steps_local
            unit_note = {"algorithmic_bytes_per_agent_step": ALGO_BYTES.get(dom, 0)}
        bytes_per_launch = algo_bytes_total / max(launches, 1)
        avg_launch_s = ms / 1e3 / max(launches, 1)
        achieved = bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        # HBM traffic per launch from the PMC passes committed under profiles/ (separate rocprofv3 --pmc runs of this very
        # command: FETCH_SIZE, WRITE_SIZE in KB; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950)
        # (only when that profile is of THIS command - size, vehicles, policy, steps and warm-up alike; otherwise null)
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "r03_pmc_summary.json")
        if os.path.exists(pmc_path):
            pj = json.load(open(pmc_path))
            pname = {"k_decide_replan": "k_replan"}.get(dom, dom)     # (the profile slot keeps the round-1 name of the kernel family)
            if pj.get("command_config") == [args.size, args.vehicles, args.policy, args.steps, args.warmup] and pname in pj.get("kernels", {}):
                pk = pj["kernels"][pname]
                traffic = (2.0 * pk["fetch_kb_avg"] + pk["write_kb_avg"]) * 1024.0
                pq = pj["kernels"].get("k_replan_quad")
                if pname == "k_replan" and pq:     # a replanning wave's region = k_replan_quad + one k_replan launch beside it
                    traffic += (2.0 * pq["fetch_kb_avg"] + pq["write_kb_avg"]) * 1024.0 * pq["launches"] / max(launches, 1)
        # Second roofline for the replanning kernel: instruction issue.  Instructions per expansion come from the SQ counter
        # passes committed under profiles/ (rocprofv3 --pmc SQ_INSTS_VALU / SQ_INSTS_SALU ... on this build at this size),
        # the expansion rate is this run's own.  Peaks (MI355X_MICROARCH.md, Execution model): a SIMD issues one wave64 vector
        # instruction every 2 cycles -> 256 CUs x 4 SIMDs x 2.4 GHz / 2; one scalar unit per CU, one instruction per cycle.
        issue = None
        quads = os.environ.get("TS_QUAD", "1") != "0" and world == 1       # (the engine's default: k_replan_quad takes the big queues)
        sq_path = os.path.join(ROOT, "profiles", "r03_sq_replan_quad_4096.json" if quads else "r03_sq_replan_4096.json")
        if dom == "k_decide_replan" and os.path.exists(sq_path):
            sqj = json.load(open(sq_path))
            sq = (sqj.get("both") or sqj.get("k_replan", {})).get("per_expansion")
            if sq:
                rate = unit_note["expansions_per_s_in_kernel"]
                valu_peak, salu_peak = 256 * 4 * 2.4e9 / 2, 256 * 2.4e9
                fv, fs = sq["SQ_INSTS_VALU"] * rate / valu_peak, sq["SQ_INSTS_SALU"] * rate / salu_peak
                issue = {"bound": "salu" if fs >= fv else "valu", "kernel": dom, "unit": "Ginstr/s",
                         "achieved": (sq["SQ_INSTS_SALU"] if fs >= fv else sq["SQ_INSTS_VALU"]) * rate / 1e9,
                         "peak": (salu_peak if fs >= fv else valu_peak) / 1e9, "frac": max(fv, fs),
                         "valu": {"per_expansion": sq["SQ_INSTS_VALU"], "achieved": sq["SQ_INSTS_VALU"] * rate / 1e9, "peak": valu_peak / 1e9, "frac": fv},
                         "salu": {"per_expansion": sq["SQ_INSTS_SALU"], "achieved": sq["SQ_INSTS_SALU"] * rate / 1e9, "peak": salu_peak / 1e9, "frac": fs},
                         "lds_per_expansion": sq.get("SQ_INSTS_LDS"), "vmem_per_expansion": sq.get("SQ_INSTS_VMEM_RD", 0) + sq.get("SQ_INSTS_VMEM_WR", 0),
                         "source": os.path.relpath(sq_path, ROOT) + " (instructions per expansion, all replanning kernels of the same command) x this run's expansions/s; "
                                   "peaks: 1 vector instruction / 2 cycles / SIMD, 1 scalar instruction / cycle / CU at 2.4 GHz"}
        host_keys = [k for k in prof if k.startswith("host_")]
        out = {
            "metric": "agent_steps_per_sec", "value": steps_done / elapsed, "unit": "agent-steps/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if (world > 1 and mode == "sharded") else "weak",
            "vs_baseline": None, "dtype": "i32", "data": "synthetic",
            "config": {
                "workload": f"{args.size}x{args.size} " + ("synthetic city (citygen" if args.world == "synthetic" else "reference-generated city (worldgen") + (", sub-block carves" if args.carves else "")
                            + f" seed {args.seed}), {v0} vehicles" + (" per GPU" if mode == "replicas" and world > 1 else "")
                            + f", {POLICY_TEXT[args.policy]}; random-walk initial routes",
                "policy": args.policy,
                "astar": {"calls": calls, "expansions": exp, "relaxations": rel, "rng_fixups": c1.rng_fixups - c0.rng_fixups,
                          "expansions_per_s": exp / elapsed, "expansions_per_search": exp / max(calls, 1)},
                # (under the default policy the cost of a replan grows with the square of the distance to the target: the
                # trip lengths the bench drives ARE its expansions per tick - SURVEY 8(d)'s exit-block goals would be ~10^6
                # expansions per search)
                "initial_route_cells": route_stats,
                "grid": args.size, "vehicles": v0, "live_vehicles_end": live_end,
                "light_groups": int(len(tables["g_light_off"]) - 1),
                "ranks_hold_identical_state": same_state,
                "multi_gpu_mode": ("single GPU" if world == 1 else
                                   "replicated state, sharded replans (one world on all ranks, bit-exact)" if mode == "sharded"
                                   else "replicas (independent worlds, bit-exact)"),
                "move_rounds_per_tick": rounds / args.steps, "bit_exact_vs_cpu_oracle": "tests/test_gpu_parity.py",
                # (this rank's ticks: under the default policy every sixth one is a replanning wave, the others are bounded
                # by their longest search)
                "tick_ms": {"min": min(tick_s) * 1e3, "median": sorted(tick_s)[len(tick_s) // 2] * 1e3, "max": max(tick_s) * 1e3},
                # per timed tick: [ms, searches, expansions] - which ticks are replanning waves, and what a search-bound tick carries
                "ticks": [[round(t * 1e3, 1), b.astar_calls - a.astar_calls, b.astar_expansions - a.astar_expansions]
                          for t, a, b in zip(tick_s, tick_c[:-1], tick_c[1:])],
                "warmup_seconds": warm_s,
            },
            "roofline": dict({
                "bound": "hbm", "kernel": dom + (" (= k_replan_quad with k_replan beside it on the replanning waves, k_replan alone on the other ticks)" if dom == "k_decide_replan" and quads else ""), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "avg_launch_us": avg_launch_s * 1e6,
                "launches": launches}, **unit_note),
            "roofline_issue": issue,
            "kernels_ms_per_tick": {k: v[0] / args.steps for k, v in prof.items() if not k.startswith("host_")},
            "host_ms_per_tick": {k: prof[k][0] / args.steps for k in host_keys},
            "setup_seconds": {"world": gen_t[0], "routes": gen_t[1]},
        }
        if sharder is not None:
            out["config"]["exchange"] = {"calls": sharder.calls, "bytes_sent_by_rank0": sharder.bytes_sent}

    # the reduced policies on the same world (N=1 only): the car-following / movement / light kernels without the searches
    if rank == 0 and world == 1 and not args.no_secondary and args.policy == "full":
        sec = {}
        for pol in ("config2", "lights"):
            e2 = new_engine()
            e2.set_device(local_rank)
            r = run_policy(e2, tables, routes, seed, pol, steps=20, warmup=5)
            e2.close()
            st = r["c1"].agent_steps - r["c0"].agent_steps
            sec[pol] = {"agent_steps_per_sec": st / r["elapsed"], "ms_per_step": r["elapsed"] / 20 * 1e3, "steps": 20, "warmup": 5,
                        "policy": POLICY_TEXT[pol], "move_rounds_per_tick": (r["c1"].move_rounds - r["c0"].move_rounds) / 20,
                        "kernels_ms_per_tick": {k: v[0] / 20 for k, v in r["prof"].items() if not k.startswith("host_")},
                        "host_ms_per_tick": {k: v[0] / 20 for k, v in r["prof"].items() if k.startswith("host_")}}
        out["secondary"] = sec

    if cpu_proc is not None:
        # let the child use up its budget; a tick that overruns it by far (a replanning wave on one core) ends the sample there
        try:
            cpu_proc.wait(timeout=max(5.0, args.cpu_seconds * 2 + 90))
        except subprocess.TimeoutExpired:
            cpu_proc.kill()
            cpu_proc.wait()
        rows = [json.loads(x) for x in open(cpu_out).read().splitlines() if x.strip()]
        os.unlink(cpu_out)
        last = rows[-1] if rows else None
        if last and last["ticks"] > 0:
            out["cpu_baseline"] = {
                "value": last["agent_steps"] / last["seconds"], "unit": "agent-steps/s", "cores": 1, "kind": "port",
                "sample": f"ticks 1-{last['ticks']} of the same {args.size}x{args.size} / {last['vehicles']}-vehicle {args.policy}-policy workload "
                          f"from its initial state ({last['seconds']:.1f} s of stepping, {last['astar_calls']} searches / "
                          f"{last['astar_expansions']} expansions inside; the GPU figure is ticks {args.warmup + 1}-{args.warmup + args.steps}); "
                          f"density map recomputed every tick as the reference does (city_model.py:1853); "
                          f"host has {os.cpu_count()} cores; oracle setup {last['setup_s']:.1f} s"
                          + ("" if cpu_proc.returncode == 0 else "; the child was ended inside the next tick (a replanning wave on one core)"),
                "ms_per_tick": last["seconds"] / last["ticks"] * 1e3, "ticks": last["ticks"],
                "astar_expansions_per_s": last["astar_expansions"] / last["seconds"],
                # the same unit on both legs (expansions of the same algorithm per second of stepping); still different ticks
                # of the run - the CPU leg never reaches the first replanning wave inside its budget
                "gpu_over_cpu_expansions": (out["config"]["astar"]["expansions_per_s"] / (last["astar_expansions"] / last["seconds"])
                                            if last["astar_expansions"] > 0 else None),
            }
        else:
            out["cpu_baseline"] = {"value": None, "unit": "agent-steps/s", "cores": 1, "kind": "port",
                                   "sample": "the oracle did not finish a tick inside the budget"}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
