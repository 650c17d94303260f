#!/usr/bin/env python3
"""bench.py - agent-steps/s of the per-tick agent-update path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 4096] [--vehicles 1000000]

A "step" is one CityModel.step() tick over the synthetic workload.  At N=1 the workload is the
configuration BASELINE.json's metric is quoted on: a 4096x4096 procedurally generated city with
10^6 vehicles.  For N>1 the driver launches one rank per GPU (torch.distributed / RCCL); every rank
runs an independent replica of the same-size workload with its own seed ("replicas only" mode of
SURVEY.md §8(e): the only mode that is bit-exact against the sequential reference today) and the
value is the whole-job aggregate: agent-steps of all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0 with the contract fields plus
  "roofline":     dominant kernel, algorithmic bytes / HIP-event time vs 8 TB/s HBM
  "cpu_baseline": the CPU oracle (single-threaded C++ restatement, kind "port") on a bounded
                  sample of the same workload, timed on this box's host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# Algorithmic bytes per agent-step, SURVEY.md §8(d): 80 B = fixed state 20 + path window 4 + map
# gathers 32 + writes 24, split over the kernels that do that work.  k_move_claim is
# pure parallelisation overhead: 0 algorithmic bytes.
ALGO_BYTES = {"k_decide_pre": 10, "k_decide_main": 44, "k_move_claim": 0, "k_move_resolve": 26,
              "k_compact": 0, "k_apply_event": 0, "k_rng": 0, "k_decide_replan": 0, "k_density": 0, "k_reach_strict": 0}

# "config 2" policy of BASELINE.md: car-following + movement kernels only.  Lights disabled, replans
# gated off; malfunction / sideswipe chances 0 (their draws are still consumed) because a stranded
# blocker forces a replan, and replanning (the GPU A*) belongs to config 3 (`--policy full`).
POLICY = {
    "TRAFFIC_LIGHT_AGENT_ALGORITHM": "DISABLED", "PATHFINDING_COOLDOWN": 10 ** 9,
    "VEHICLE_STUCK_RECOMPUTE_THRESHOLD": 10 ** 9, "VEHICLE_STUCK_RECOMPUTE_THRESHOLD_INTERSECTION": 10 ** 9,
    "VEHICLE_CONTRAFLOW_OVERTAKE_ACTIVE": False, "VEHICLE_STUCK_CONTRAFLOW_ENABLED": False,
    "VEHICLE_MALFUNCTION_CHANCE": 0.0, "VEHICLE_SIDESWIPE_COLLISION_CHANCE": 0.0, "RAIN_ENABLED": False,
}


def make_workload(size, vehicles, seed, world="synthetic"):
    from trafficsimulation_amd import citygen
    t0 = time.time()
    if world == "reference":    # the reference's own city for (size, seed): exact but interpreted, minutes at 4096^2
        from trafficsimulation_amd import worldgen
        tables = worldgen.generate_world(size, size, seed=seed, rain_enabled=False, enable_traffic=False)
    else:
        tables = citygen.generate(size, size, seed=seed)
    t1 = time.time()
    lo, hi = (150, 400) if size >= 1024 else (30, 120)
    routes = citygen.make_routes(tables, vehicles, seed=seed + 1, min_len=lo, max_len=hi)
    return tables, routes, (t1 - t0, time.time() - t1)


FULL_POLICY = {"RAIN_ENABLED": False}   # everything else at the reference's defaults (config.py)
# BASELINE config 3's kernels without the searches: queue-actuated light groups stepping inside the shuffled order
# (phase + intersection clearance), car-following and movement; replans gated off like config 2
LIGHTS_POLICY = dict(POLICY, TRAFFIC_LIGHT_AGENT_ALGORITHM="QUEUE_ACTUATED")


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--vehicles", type=int, default=1_000_000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--policy", choices=["config2", "lights", "full"], default="config2",
                    help="config2 = car-following + movement only (BASELINE config 2); full = reference defaults: "
                         "QUEUE_ACTUATED lights, replanning (GPU A*), contraflow, malfunctions (BASELINE config 3)")
    ap.add_argument("--world", choices=["synthetic", "reference"], default="synthetic",
                    help="synthetic = citygen look-alike (fast to build); reference = worldgen, the reference's city for the seed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from trafficsimulation_amd._lib import new_engine
    seed = args.seed + 1000 * rank
    tables, routes, gen_t = make_workload(args.size, args.vehicles, seed, args.world)
    api = new_engine()
    api.set_device(local_rank)
    setup(api, tables, routes, seed, policy=args.policy)
    v0 = api.num_vehicles()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    api.step(args.warmup)
    api.profile_enable(True)
    c0 = api.counters()
    barrier()
    t0 = time.perf_counter()
    api.step(args.steps)   # ts_step returns only after the engine's stream has drained
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    barrier()
    c1 = api.counters()
    elapsed = t1 - t0
    steps_done = c1.agent_steps - c0.agent_steps
    rounds = c1.move_rounds - c0.move_rounds
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ss = torch.tensor([steps_done], dtype=torch.int64, device="cuda")
        dist.all_reduce(ss, op=dist.ReduceOp.SUM)
        elapsed, steps_done = float(tt.item()), int(ss.item())
    prof = api.profile()
    live_end = api.num_vehicles()

    out = None
    if rank == 0:
        total_ms = {k: v[0] for k, v in prof.items()}
        dom = max((k for k in total_ms if ALGO_BYTES.get(k, 0) > 0), key=lambda k: total_ms[k])
        host_keys = [k for k in prof if k.startswith("host_")]
        ms, launches, items = prof[dom]
        # algorithmic bytes per launch: per-unit figure x vehicles stepped, spread over this kernel's launches
        veh_steps_local = c1.agent_steps - c0.agent_steps
        bytes_per_launch = ALGO_BYTES[dom] * veh_steps_local / max(launches, 1)
        avg_launch_s = ms / 1e3 / max(launches, 1)
        achieved = bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        # HBM traffic per launch from the PMC passes committed under profiles/ (separate rocprofv3 --pmc runs of
        # this same command: FETCH_SIZE, WRITE_SIZE in KB; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
        # for gfx950 - verified here on k_rank_invert, whose coalesced 4-byte reads report exactly half)
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
        if os.path.exists(pmc_path) and args.size == 4096 and args.vehicles == 1_000_000 and args.policy == "config2":
            pmc = json.load(open(pmc_path)).get(dom)
            if pmc:
                traffic = (2.0 * pmc["fetch_kb_avg"] + pmc["write_kb_avg"]) * 1024.0
        out = {
            "metric": "agent_steps_per_sec", "value": steps_done / elapsed, "unit": "agent-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i32", "data": "synthetic",
            "config": {
                "workload": f"{args.size}x{args.size} " + ("synthetic city (citygen" if args.world == "synthetic" else "reference-generated city (worldgen")
                            + f" seed {args.seed}), {v0} vehicles per GPU, "
                            + ("config-2 policy: car-following + movement kernels, lights DISABLED, replans gated off, "
                               "malfunction/sideswipe chance 0; random-walk routes" if args.policy == "config2" else
                               "lights policy: config 2 plus QUEUE_ACTUATED light groups (phase change, intersection "
                               "clearance) stepping in the shuffled order; replans gated off; random-walk routes"
                               if args.policy == "lights" else
                               "full policy: reference defaults (QUEUE_ACTUATED lights, GPU A* replanning, contraflow, "
                               "malfunctions/sideswipes), rain and spawning off; random-walk initial routes"),
                "policy": args.policy,
                "astar": {"calls": c1.astar_calls - c0.astar_calls, "expansions": c1.astar_expansions - c0.astar_expansions,
                          "relaxations": c1.astar_relaxations - c0.astar_relaxations, "rng_fixups": c1.rng_fixups - c0.rng_fixups},
                "grid": args.size, "vehicles_per_gpu": v0, "live_vehicles_end": live_end,
                "light_groups": int(len(tables["g_light_off"]) - 1), "multi_gpu_mode": "replicas only (bit-exact)",
                "move_rounds_per_tick": rounds / args.steps, "bit_exact_vs_cpu_oracle": "tests/test_gpu_parity.py",
            },
            "roofline": {
                "bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_agent_step": ALGO_BYTES[dom], "avg_launch_us": avg_launch_s * 1e6,
                "launches": launches,
            },
            "kernels_ms_per_tick": {k: v[0] / args.steps for k, v in prof.items() if not k.startswith("host_")},
            "host_ms_per_tick": {k: prof[k][0] / args.steps for k in host_keys},
            "setup_seconds": {"world": gen_t[0], "routes": gen_t[1]},
        }
    api.close()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle
        cpu = pyoracle.load()
        ts = time.time()
        setup(cpu, tables, routes, seed, extra={"eager_density": 1}, policy=args.policy)
        setup_s = time.time() - ts
        cpu.step(1)
        a0 = cpu.counters().agent_steps
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < args.cpu_seconds and n < args.steps:
            cpu.step(1)
            n += 1
        dt = time.perf_counter() - t0
        a1 = cpu.counters().agent_steps
        out["cpu_baseline"] = {
            "value": (a1 - a0) / dt, "unit": "agent-steps/s", "cores": 1, "kind": "port",
            "sample": f"{n} ticks of the same {args.size}x{args.size}/{v0}-vehicle workload after 1 warm-up tick, "
                      f"density map recomputed every tick as the reference does (city_model.py:1853); "
                      f"host has {os.cpu_count()} cores; oracle setup {setup_s:.1f}s",
            "ms_per_tick": dt / max(n, 1) * 1e3,
        }
        cpu.close()
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
