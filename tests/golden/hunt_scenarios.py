"""Differential hunt for the CPU oracle: one random scenario is run by the reference's own sources (make_golden's
harness; importable only in the build container), captured as a trace under /tmp and replayed on the oracle tick by tick,
then the world is rebuilt from (size, seed) alone.  Not a test; DESIGN.md §2 quotes the totals.
usage: python tests/golden/hunt_scenarios.py CASE      (cases >= 100 also move the rarely-touched Defaults;
HUNT_NOBATCH=1: every case with PATHFINDING_BATCHING=False)"""
import sys, os, json, random, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, HERE)
case = int(sys.argv[1])
pr = random.Random(77000 + case)
size = pr.choice([64, 64, 80, 96]) if case < 300 else pr.choice([112, 128])     # cases >= 300: larger maps, denser traffic
d = {}
algo = pr.choice(["QUEUE_ACTUATED", "QUEUE_ACTUATED", "FIXED_TIME", "NEIGHBOR_GREEN_WAVE", "NEIGHBOR_PRESSURE_CONTROL", "DISABLED"])
d["TRAFFIC_LIGHT_AGENT_ALGORITHM"] = algo
if pr.random() < 0.5: d["VEHICLE_MALFUNCTION_CHANCE"] = pr.choice([0.0, 0.001, 0.005]); d["VEHICLE_MALFUNCTION_DURATION"] = pr.choice([10, 25, 60])
if pr.random() < 0.5: d["VEHICLE_SIDESWIPE_COLLISION_CHANCE"] = pr.choice([0.0, 0.05, 0.3]); d["VEHICLE_SIDESWIPE_COLLISION_DURATION"] = pr.choice([10, 30])
if pr.random() < 0.4: d["RAIN_ENABLED"] = False
else: d.update(RAIN_RADIUS_MIN=pr.choice([5, 8, 12]), RAIN_RADIUS_MAX=pr.choice([14, 20, 30]), RAIN_SPAWN_CHANCE=pr.choice([0.05, 0.2, 0.5]))
d["INTERNAL_POPULATION_TRAFFIC_PER_DAY"] = pr.choice([0, 3000, 10000, 30000])
d["PASSING_POPULATION_TRAFFIC_PER_DAY"] = pr.choice([0, 1000, 2400, 10000])
d["TOTAL_SERVICE_VEHICLES_FOOD"] = pr.choice([0, 20, 50, 300]); d["TOTAL_SERVICE_VEHICLES_WASTE"] = pr.choice([0, 20, 50, 300])
if pr.random() < 0.3: d["GRADUAL_CITY_BLOCK_RESOURCES"] = False
if pr.random() < 0.3: d["SERVICE_VEHICLE_LOAD_TIME"] = pr.choice([1, 5, 40])
if pr.random() < 0.3: d["PATHFINDING_COOLDOWN"] = pr.choice([0, 2, 12])
if pr.random() < 0.3: d["VEHICLE_STUCK_RECOMPUTE_THRESHOLD"] = pr.choice([5, 15, 60])
if pr.random() < 0.3: d["TRAFFIC_LIGHT_GREEN_DURATION"] = pr.choice([5, 12, 40])
if case >= 100:   # second batch: the rarely-moved knobs
    if pr.random() < 0.4: d["VEHICLE_CONTRAFLOW_OVERTAKE_ACTIVE"] = False
    if pr.random() < 0.4: d["VEHICLE_STUCK_CONTRAFLOW_ENABLED"] = pr.choice([True, False]); d["VEHICLE_STUCK_CONTRAFLOW_THRESHOLD"] = pr.choice([8, 20, 60]); d["VEHICLE_STUCK_CONTRAFLOW_THRESHOLD_INTERSECTION"] = pr.choice([2, 10])
    if pr.random() < 0.4: d["VEHICLE_MAX_CONTRAFLOW_OVERTAKE_STEPS"] = pr.choice([3, 6, 12]); d["VEHICLE_CONTRAFLOW_OVERTAKE_DURATION"] = pr.choice([5, 30])
    if pr.random() < 0.4: d["VEHICLE_MAX_CONTRAFLOW_STUCK_DETOUR_STEPS"] = pr.choice([8, 20, 40]); d["VEHICLE_CONTRAFLOW_STUCK_DETOUR_DURATION"] = pr.choice([3, 10])
    if pr.random() < 0.3: d["VEHICLE_STUCK_RECOMPUTE_THRESHOLD_INTERSECTION"] = pr.choice([1, 3, 8])
    if pr.random() < 0.3: d["VEHICLE_MAX_SPEED"] = pr.choice([2, 3, 8]); 
    if pr.random() < 0.3: d["VEHICLE_AWARENESS_RANGE"] = pr.choice([4, 7, 14])
    if pr.random() < 0.3: d["VEHICLE_TURN_PENALTY"] = pr.choice([0, 3, 25])
    if pr.random() < 0.3: d["VEHICLE_ROAD_TYPES_PENALTIES_ENABLED"] = False
    if pr.random() < 0.3: d["VEHICLE_DYNAMIC_PENALTIES_ENABLED"] = False
    if pr.random() < 0.3: d["VEHICLE_OBSTACLE_PENALTY_VEHICLE"] = pr.choice([100, 5000]); d["VEHICLE_OBSTACLE_PENALTY_STOP"] = pr.choice([50, 2000])
    if pr.random() < 0.3: d["TRAFFIC_LIGHT_TRANSITION_CLEARANCE_ENABLED"] = False
    if pr.random() < 0.3: d["TRAFFIC_LIGHT_QUEUE_ACTUATED_MIN_GREEN"] = pr.choice([1, 8]); d["TRAFFIC_LIGHT_QUEUE_ACTUATED_MAX_GREEN"] = pr.choice([10, 45]); d["TRAFFIC_LIGHT_QUEUE_ACTUATED_GAP"] = pr.choice([1, 5])
    if pr.random() < 0.3: d["RAIN_SPEED_REDUCTION"] = pr.choice([1, 3]); 
    if pr.random() < 0.3: d["RAIN_OCCURRENCES_MAX"] = pr.choice([1, 3, 8]); d["RAIN_COOLDOWN"] = pr.choice([0, 5, 40])
    if pr.random() < 0.3: d["PATHFINDING_CACHE"] = pr.choice([True, False])
    if pr.random() < 0.3: d["TIME_PER_STEP_IN_SECONDS"] = pr.choice([2, 10, 30])
if os.environ.get("HUNT_NOBATCH") == "1": d["PATHFINDING_BATCHING"] = False      # round 3: the same cases on the non-batched step path
kw = {}
if pr.random() < 0.3: kw["carve_subblock_roads"] = True; kw["subblock_chance"] = 0.8
if pr.random() < 0.3: kw["ring_road_type"] = pr.choice(["R1", "R3", None])
if pr.random() < 0.2: kw["optimized_intersections"] = False
if pr.random() < 0.2: kw["forward_traffic_light_range"] = True; kw["forward_traffic_light_range_intersections"] = pr.choice(["Skip", "Include in Range", "Include as Extra"])
spec = dict(size=size, seed=300 + case, vehicles=pr.choice([20, 60, 120]) if case < 300 else pr.choice([150, 300]),
            ticks=pr.choice([80, 120, 160]) if case < 300 else pr.choice([50, 70]), defaults=d, model_kwargs=kw)
if pr.random() < 0.25: spec["height"] = pr.choice([64, 80, 112])
import make_golden as mg
mg._setup_paths()
os.makedirs('/tmp/hunt_traces', exist_ok=True)
mg.HERE = '/tmp/hunt_traces'
name = f"hunt{case}"
mg.SCENARIOS[name] = spec
t0 = time.time()
try:
    mg.run_scenario(name)
except Exception as e:
    import traceback; traceback.print_exc()
    print(f"CASE {case} REFERENCE-HARNESS-EXC {type(e).__name__}: {e} spec={json.dumps(spec)}", flush=True); sys.exit(0)
gen = time.time() - t0
from trafficsimulation_amd.world import load_trace
from tests.trace_util import setup_from_trace, check_initial, replay_and_compare
from oracle import pyoracle
path = f"/tmp/hunt_traces/trace_{name}.npz"
tr = load_trace(path)
api = pyoracle.load()
try:
    setup_from_trace(api, tr)
    check_initial(api, tr)
    n = replay_and_compare(api, tr)
    # seed-only world check as well
    from tests.test_worldgen import _seed_only
    _seed_only(tr)
    print(f"CASE {case} OK ticks={n} gen={gen:.0f}s spec={json.dumps(spec)}", flush=True)
    os.remove(path)
except BaseException as e:
    msg = str(e).split(chr(10))[0][:300]
    print(f"CASE {case} MISMATCH {type(e).__name__}: {msg} spec={json.dumps(spec)}", flush=True)
