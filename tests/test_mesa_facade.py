"""The Mesa-shaped facade: same names / argument meaning as the reference's CityModel,
VehicleAgent, CellAgent, schedule and grid.  Driven here over the CPU oracle so the host logic is covered
without a GPU; tests/test_gpu_parity.py::test_facade_on_hip runs the same check on the HIP engine."""
import numpy as np
import pytest

from trafficsimulation_amd import _capi as capi
from trafficsimulation_amd.mesa_api import CityModel, VehicleAgent, agent_portrayal
from trafficsimulation_amd.world import load_trace
from tests.trace_util import trace_path


def run_facade_against_trace(engine, name="lights_qa_96_s2", ticks=30):
    tr = load_trace(trace_path(name))
    m = CityModel.from_tables(tr, seed=1, defaults=tr["defaults_json"], engine=engine,
                              global_state=tr["global_rng_after_worldgen"], sched_state=tr["sched_rng_initial"])
    assert (m.width, m.height) == (int(tr["width"]), int(tr["height"]))
    vehicles = []
    n_static = len(m.schedule.agents)            # light groups, the clock agent (and the CityBlocks of a world that has them)
    assert n_static >= len(m.intersection_light_groups) + 1
    for i, (s, g) in enumerate(zip(tr["v_start_xy"], tr["v_goal_xy"])):
        v = VehicleAgent(f"gv_{i}", m, m.cell(int(s[0]), int(s[1])), m.cell(int(g[0]), int(g[1])), population_type="through")
        vehicles.append(v)
    assert len(m.active_vehicle_agents) == len(vehicles)
    assert len(m.schedule.agents) == n_static + len(vehicles)
    H, W = m.height, m.width
    fields = tr["veh_fields"]
    for t in range(ticks):
        m.step()
        want = tr["veh_rows"][tr["veh_off"][t]:tr["veh_off"][t + 1]]
        live = m.active_vehicle_agents
        assert [v._spawn_idx for v in live] == list(want[:, 0])
        for v, r in zip(live[::7], want[::7]):
            assert v.pos == (r[1], r[2])
            assert v.current_speed == r[fields.index("current_speed")]
            assert (v.direction is None) == (r[fields.index("direction")] < 0)
            assert v.get_portrayal()["Position"] == v.pos
        occ = np.unpackbits(tr["occ_t"][t])[:H * W].reshape(H, W)
        assert np.array_equal(m.occupancy_map, occ)
        assert np.array_equal(m.stop_map, np.unpackbits(tr["stop_t"][t])[:H * W].reshape(H, W))
    assert m.step_count == ticks
    # MultiGrid view: the static cell first, then the vehicles standing there
    v = m.active_vehicle_agents[0]
    x, y = v.pos
    contents = m.grid[x, y]
    assert contents[0] is m.cell(x, y) and v in contents[1:]
    assert m.get_cell_contents(-1, 0) == []
    # removed vehicles report pos None like MultiGrid.remove_agent
    gone = [v for v in vehicles if v not in m.active_vehicle_agents]
    assert all(g.pos is None for g in gone)
    # light-group views and the UI's direct writes (cell.py:241-251) reach the engine before the next tick
    g0 = m.intersection_light_groups[0]
    assert g0.current_phase in (0, 1) and g0.pending_phase in (None, 0, 1)
    g0.set_all_stop()
    for tl in g0.traffic_lights:
        assert m.stop_map[tl.position[1], tl.position[0]] == 1
        for cb in tl.controlled_blocks:
            assert m.stop_map[cb.position[1], cb.position[0]] == 1
    m.schedule.step()
    assert agent_portrayal(m.traffic_lights[0])["Shape"] == "rect"
    stats = m.dynamic_traffic_generator.cached_stats
    assert stats["live_through"] == len(m.active_vehicle_agents)
    with pytest.raises(NotImplementedError):
        v.step()
    # duplicate ids are rejected like the Mesa scheduler does
    with pytest.raises(Exception):
        VehicleAgent("gv_0", m, m.cell(*map(int, tr["v_start_xy"][0])), m.cell(*map(int, tr["v_goal_xy"][0])))
    return m


def test_facade_over_oracle(oracle):
    run_facade_against_trace(oracle)


def test_facade_without_batching_over_oracle(oracle):
    """`CityModel(defaults={"PATHFINDING_BATCHING": False})`: the reference's other step path through the same facade, against
    the trace captured from the reference with that switch off."""
    run_facade_against_trace(oracle, name="nobatch_full_96_s28", ticks=40)
    assert oracle.default_params().pathfinding_batching == 1        # (batched is the default, as in config.py:411)


def run_facade_with_generator(engine, name="service_64_s15", ticks=420):
    """The engine's traffic generator behind the facade: vehicles it spawns appear as views, service vehicles carry
    their load / block / phase, CityBlock views read the stock (vehicle_service.py, city_block.py, city_model.py:1738)."""
    import json
    tr = load_trace(trace_path(name))
    dta = json.loads(str(tr["dta_params"]))
    m = CityModel.from_tables(tr, seed=1, defaults=tr["defaults_json"], engine=engine, traffic=dta,
                              global_state=tr["global_rng_before_day0"], sched_state=tr["sched_rng_initial"])
    for i, (s, g) in enumerate(zip(tr["v_start_xy"], tr["v_goal_xy"])):
        VehicleAgent(f"gv_{i}", m, m.cell(int(s[0]), int(s[1])), m.cell(int(g[0]), int(g[1])), population_type="through")
    assert len(m.city_blocks) == len(tr["blk_type"])
    fields = tr["cnt_fields"]
    seen_service = set()
    for t in range(ticks):
        m.step()
        want = tr["veh_rows"][tr["veh_off"][t]:tr["veh_off"][t + 1]]
        live = m.active_vehicle_agents
        assert [v._spawn_idx for v in live] == list(want[:, 0])
        for v in live:
            if v.vehicle_type in ("food", "waste"):
                seen_service.add(v._spawn_idx)
                assert v.population_type == "through" and v.phase in ("to_block", "servicing", "to_exit")
                assert 0.0 <= v.current_load <= v.max_load
                assert (v.phase == "servicing") <= v.is_parked
                assert v.get_portrayal()["Type"].endswith("ServiceVehicle")
        got = np.asarray([[b.get_food_units(), b.get_waste_units()] for b in m.city_blocks.values()])
        assert np.array_equal(got, tr["blk_rows"][t])
    # the generator's live attributes (dynamic_traffic_generator.py:102-131) follow the tick; its cached_stats is the dict the
    # generator refreshed at its last update (every STATISTICS_UPDATE_INTERVAL ticks), as in the reference
    dta = m.dynamic_traffic_generator
    want_c = dict(zip(fields, tr["cnt_rows"][ticks - 1]))
    for k in ("created_service_food", "created_service_waste", "live_service_food", "live_service_waste", "live_through", "parked"):
        assert getattr(dta, k) == want_c[k], k
    stats = dta.cached_stats
    assert (stats == {}) == (ticks < 20) and (ticks < 20 or "daily_total_service_food" in stats)
    assert seen_service, "the scenario is expected to spawn service vehicles"
    # a vehicle added through the facade after engine-side spawns gets the next free spawn index
    v = VehicleAgent("late_one", m, m.cell(*map(int, tr["v_start_xy"][0])), m.cell(*map(int, tr["v_goal_xy"][0])))
    assert v._spawn_idx == m.engine.num_spawned() - 1 and v in m.active_vehicle_agents
    assert len(m.schedule.agents) == len(m.intersection_light_groups) + len(m.city_blocks) + 1 + len(m.active_vehicle_agents)
    return m


def test_facade_with_generator_over_oracle(oracle):
    run_facade_with_generator(oracle)


def test_facade_rain_control_over_oracle(oracle):
    """What the RainControl card and the /spawn_rain handler do (rain_control.py:22-73): len(model.rains),
    rain_manager.cooldown, rain_manager.add_random_rain() between ticks; rain_map then follows the clouds."""
    tr = load_trace(trace_path("rain_96_s14"))
    m = CityModel.from_tables(tr, seed=1, defaults=tr["defaults_json"], engine=oracle,
                              global_state=tr["global_rng_after_worldgen"], sched_state=tr["sched_rng_initial"])
    assert m.rain_manager.cooldown == 0 and len(m.rains) == 0
    m.rain_manager.add_random_rain()
    assert len(m.rains) == 1 and m.rain_manager.counter == 1
    n_sched = m.schedule.get_agent_count()
    for _ in range(30):
        m.step()
    assert m.rain_map.sum() > 0 or len(m.rains) >= 1
    assert m.schedule.get_agent_count() >= n_sched - 1   # the cloud is a scheduled agent until it leaves the map


def test_facade_ui_vehicle_handlers_over_oracle(oracle):
    """CreateVehicleHandler / CreateServiceVehicleHandler (vehicle_control.py:182-252): get_start_blocks,
    get_valid_exits (empty for a highway entrance, like the reference), VehicleAgent(vid, model, start, target) and
    ServiceVehicleAgent(vid, model, entrance, sv_type) between ticks."""
    import json
    from trafficsimulation_amd.mesa_api import ServiceVehicleAgent
    tr = load_trace(trace_path("service_64_s15"))
    dta = json.loads(str(tr["dta_params"]))
    m = CityModel.from_tables(tr, seed=1, defaults=tr["defaults_json"], engine=oracle, traffic=dict(dta, P_int=0, P_thr=0,
                              service_food=0, service_waste=0),
                              global_state=tr["global_rng_before_day0"], sched_state=tr["sched_rng_initial"])
    starts = m.get_start_blocks()
    be = m.block_entrances[0]
    exits = m.get_valid_exits(be)
    assert be not in exits and len(exits) == len(m.block_entrances) - 1 + len(m.highway_exits)
    assert m.get_valid_exits(m.highway_entrances[0]) == []
    assert m.get_valid_exits(m.cell(0, 0)) == [] and be in starts
    v = VehicleAgent("V1", m, be, exits[0])
    sv = ServiceVehicleAgent("SV1", m, m.highway_entrances[0], "Food")
    sw = ServiceVehicleAgent("SV2", m, m.highway_entrances[1], "Waste")
    assert sv.phase == "to_block" and sv.current_load == sv.max_load == 50.0 and sw.current_load == 0.0
    assert sv.get_vehicle_type_name() == "FoodServiceVehicle" and sv.population_type == "through"
    for _ in range(150):
        m.step()
    live = m.active_vehicle_agents
    assert all(x in (v, sv, sw) for x in live)
    stats = m.dynamic_traffic_generator.cached_stats
    assert stats["live_service_food"] + stats["live_service_waste"] == sum(1 for x in live if x is not v)
    with pytest.raises(Exception):
        ServiceVehicleAgent("SV1", m, m.highway_entrances[0], "Food")


def test_facade_light_group_links_over_oracle(oracle):
    """What the UI's group handlers call (traffic_light_control.py:300-400): get_opposite_traffic_lights() with its
    populate_links() side effect, get_neighbor_groups(), set_all_*_with_neighbors()."""
    tr = load_trace(trace_path("lights_npress_96_s6"))
    m = CityModel.from_tables(tr, seed=1, defaults=tr["defaults_json"], engine=oracle,
                              global_state=tr["global_rng_after_worldgen"], sched_state=tr["sched_rng_initial"])
    g = m.intersection_light_groups[3]
    ctor = np.asarray(tr["g_neighbors_ctor"]).reshape(-1, 4, 2)[3]
    full = np.asarray(tr["g_neighbors"]).reshape(-1, 4, 2)[3]
    assert not m.engine.group_links(3)
    assert len(g.get_neighbor_groups()) == int(np.sum((ctor[:, 0] >= 0) & (ctor[:, 1] >= 0)))
    pairs = g.get_opposite_traffic_lights()
    assert set(pairs) == {"N-S", "W-E"} and all(tl in g.traffic_lights for ax in pairs.values() for tl in ax)
    assert m.engine.group_links(3)            # the call re-populated the group's links
    nb = g.get_neighbor_groups()
    assert len(nb) == int(np.sum((full[:, 0] >= 0) & (full[:, 1] >= 0)))
    g.set_all_stop_with_neighbors()
    for grp in [g] + list(nb.values()):
        for tl in grp.traffic_lights:
            assert m.stop_map[tl.position[1], tl.position[0]] == 1
    m.step()
    with pytest.raises(NotImplementedError):
        g.get_intermediate_groups()


def test_facade_builds_reference_world_from_seed(oracle):
    """CityModel(width, height, seed=...) with nothing else: world-gen, the state it leaves the global stream in, day 0
    of the traffic generator and the first ticks are those of the reference's run with random.seed(11)."""
    import json
    tr = load_trace(trace_path("config1_64_s11"))
    dta = json.loads(str(tr["dta_params"]))
    m = CityModel(64, 64, seed=11, defaults=tr["defaults_json"], engine=oracle, traffic=dta)
    for k in ("allowed_dirs_map", "is_road_map", "road_type_map", "intersection_map"):
        assert np.array_equal(getattr(m, k), tr[k])
    assert [c.position for c in m.block_entrances] == [tuple(p) for p in tr["block_entrances_xy"]]
    for i, (s, g) in enumerate(zip(tr["v_start_xy"], tr["v_goal_xy"])):
        VehicleAgent(f"gv_{i}", m, m.cell(int(s[0]), int(s[1])), m.cell(int(g[0]), int(g[1])), population_type="through")
    H, W = m.height, m.width
    for t in range(60):
        m.step()
        want = tr["veh_rows"][tr["veh_off"][t]:tr["veh_off"][t + 1]]
        assert [v._spawn_idx for v in m.active_vehicle_agents] == list(want[:, 0])
        assert np.array_equal(m.occupancy_map, np.unpackbits(tr["occ_t"][t])[:H * W].reshape(H, W))
        assert np.array_equal(m.rain_map, np.unpackbits(tr["rain_t"][t])[:H * W].reshape(H, W))


def test_facade_intermediate_group_handlers_over_oracle(oracle):
    """/set_group_neighbors_intermediate_stop|go (traffic_light_control.py:389-399) on a worldgen city whose light
    groups have intermediate groups (ring road R3, seed 103: the reference's tables are in worlds.npz)."""
    m = CityModel(96, 96, seed=103, ring_road_type="R3", defaults={"RAIN_ENABLED": False}, engine=oracle)
    with_mid = [g for g in m.intersection_light_groups if g.get_intermediate_groups()]
    assert with_mid, "this world is expected to have intermediate groups in its constructor-time links"
    g = with_mid[0]
    involved = [g] + list(g.get_neighbor_groups().values()) + g.get_intermediate_groups()
    g.set_all_stop_with_neighbors_and_intermediate()
    for grp in involved:
        for tl in grp.traffic_lights:
            assert m.stop_map[tl.position[1], tl.position[0]] == 1
    g.set_all_go_with_neighbors_and_intermediate()
    for grp in involved:
        for tl in grp.traffic_lights:
            assert m.stop_map[tl.position[1], tl.position[0]] == 0
    m.step()


def test_facade_cell_types_blocks_and_block_queries_over_oracle(oracle):
    """Worlds from worldgen carry the reference's cell_type / block_id per cell (pinned by worlds.npz), so the facade's
    cells report them like CellAgent does, `city_blocks` is keyed by block_id, and the block pickers of
    city_model.py:2017-2087 work on the engine's block stock."""
    import json
    tr = load_trace(trace_path("config1_64_s11"))
    m = CityModel(64, 64, seed=11, defaults=tr["defaults_json"], engine=oracle, traffic=json.loads(str(tr["dta_params"])))
    assert m.cell(0, 0).cell_type == "Wall" and m.cell(0, 0).block_id is None
    for be in m.block_entrances:
        assert be.cell_type == "BlockEntrance" and be.block_id in m.city_blocks
        assert be.block_type == m.city_blocks[be.block_id].block_type
        assert be.get_portrayal()["Block ID"] == be.block_id and be.get_portrayal()["Color"] == "magenta"
        assert be in m.city_blocks[be.block_id].get_entrances()
    assert {c.cell_type for c in m.highway_entrances} == {"HighwayEntrance"}
    assert {c.cell_type for c in m.highway_exits} == {"HighwayExit"}
    assert {c.cell_type for c in m.controlled_roads} == {"ControlledRoad"}
    kinds = {m.cell(x, y).cell_type for y in range(64) for x in range(64)}
    assert {"Sidewalk", "R2", "Intersection", "TrafficLight"} <= kinds
    assert m.is_type(0, 0, "Wall") and not m.is_type(-1, 0, "Wall") and m.next_cell_in_direction(3, 4, "N") == (3, 5)
    assert list(m.city_blocks) == [b.block_id for b in m.get_all_city_blocks()]
    for _ in range(40):
        m.step()
    by_food = m.get_blocks_needing_food(sort_by="food")
    assert [b.get_food_units() for b in by_food] == sorted(b.get_food_units() for b in by_food)
    assert all(b.block_type in ("Market", "Leisure") for b in by_food)
    by_waste = m.get_all_city_blocks(sort_by="waste")
    assert [b.get_waste_units() for b in by_waste] == sorted((b.get_waste_units() for b in by_waste), reverse=True)
    assert m.get_block_most_in_need_of_waste_pickup() is by_waste[0]
    assert m.get_block_most_in_need_of_food() is (by_food[0] if by_food else None)
    typed = m.get_residential_city_blocks() + m.get_office_city_blocks() + m.get_market_city_blocks() \
        + m.get_leisure_city_blocks() + m.get_other_city_blocks()
    assert sorted(b.block_id for b in typed) == sorted(m.city_blocks)
    assert m.get_city_blocks_by_type("Office") == m.get_city_blocks_by_types(["Office"])
    assert m.get_city_blocks_by_types(None) == m.get_all_city_blocks()


def test_example_runner_over_oracle(oracle):
    """examples/run_city.py: its traffic constants are config.py's (as recorded in the config-1 fixture), and a short run
    from (size, seed) alone spawns traffic and reports statistics."""
    import json
    from examples import run_city
    tr = load_trace(trace_path("config1_64_s11"))
    want = json.loads(str(tr["dta_params"]))
    want.pop("pending_day0")
    assert run_city.TRAFFIC == want
    lines = []
    m = run_city.run(64, 11, 40, every=20, engine=oracle, out=lines.append)
    assert len(lines) == 3 and lines[0].startswith("city 64x64 seed 11: 4 light groups, 4 blocks")
    assert m.step_count == 40 and len(m.active_vehicle_agents) > 0
    assert np.array_equal(m.allowed_dirs_map, tr["allowed_dirs_map"])


@pytest.mark.parametrize("tag", ["rect", "ring_r1", "carve_dense", "seed204"])
def test_facade_display_names_match_reference(oracle, tag):
    """CellAgent.get_display_name() - the labels of the UI's light / start / target / entrance drop-downs
    (ui_modules/*.py) - for every light, block entrance, highway entrance and exit of worlds the reference built."""
    import json
    import os
    from tests.trace_util import GOLDEN
    z = np.load(os.path.join(GOLDEN, "worlds.npz"), allow_pickle=False)
    e = next(x for x in json.loads(str(z["index"])) if x["tag"] == tag)
    want = json.loads(str(z[f"{tag}/display_names"]))
    m = CityModel(e["width"], e["height"], seed=e["seed"], defaults={"RAIN_ENABLED": False}, engine=oracle, **e["kwargs"])
    assert [str(c.get_display_name()) for c in m.traffic_lights] == want["lights"]
    assert [str(c.get_display_name()) for c in m.block_entrances] == want["block_entrances"]
    assert [str(c.get_display_name()) for c in m.highway_entrances] == want["highway_entrances"]
    assert [str(c.get_display_name()) for c in m.highway_exits] == want["highway_exits"]
    inter = next(c for g in m.intersection_light_groups for c in g.intersection_cells if c.cell_type == "Intersection")
    assert inter.get_display_name() == f"Intersection_{inter.position[0]}_{inter.position[1]}"
    assert m.block_entrances[0].is_block_entrance() and m.highway_exits[0].is_highway_exit()
    assert m.highway_entrances[0].is_highway_entrance() and m.controlled_roads[0].is_controlled_road()


def test_remove_vehicle_counts_the_callers_population_type(oracle):
    """CityModel.remove_vehicle (city_model.py:1920-1941) decrements live_internal / live_through by the `population_type`
    the CALLER passes - 'undefined' by default, i.e. neither - not by the vehicle's own population."""
    import json
    tr = load_trace(trace_path("dta_64_s12"))
    m = CityModel(64, 64, seed=11, defaults=tr["defaults_json"], engine=oracle, traffic=json.loads(str(tr["dta_params"])))
    for _ in range(40):
        m.step()
    vs = [v for v in m.active_vehicle_agents if not getattr(v, "is_service", False)][:3]
    assert len(vs) == 3
    c0 = m.engine.counters()
    m.remove_vehicle(vs[0])                                   # defaults: 'undefined'
    c1 = m.engine.counters()
    assert (c1.live_internal, c1.live_through) == (c0.live_internal, c0.live_through)
    m.remove_vehicle(vs[1], population_type="internal")
    c2 = m.engine.counters()
    assert (c2.live_internal, c2.live_through) == (c0.live_internal - 1, c0.live_through)
    m.remove_vehicle(vs[2], "through", "undefined")
    c3 = m.engine.counters()
    assert (c3.live_internal, c3.live_through) == (c0.live_internal - 1, c0.live_through - 1)
    assert len(m.active_vehicle_agents) == len([v for v in m.active_vehicle_agents])


def test_facade_cached_stats_keys_over_oracle(oracle):
    """model.dynamic_traffic_generator.cached_stats carries what ui_modules/traffic_statistics.py:122-150 reads."""
    import json
    tr = load_trace(trace_path("dta_64_s12"))
    m = CityModel(64, 64, seed=11, defaults=tr["defaults_json"], engine=oracle, traffic=json.loads(str(tr["dta_params"])))
    assert m.dynamic_traffic_generator.cached_stats == {}
    for _ in range(25):
        m.step()
    cs = m.dynamic_traffic_generator.cached_stats
    for k in ("daily_total_internal", "remaining_through", "percentage_created_internal", "eta_service_food", "avg_duration_through_live",
              "avg_time_per_unit_internal_total", "live_average_stuck_duration", "live_max_stuck_duration", "avg_daily_difference"):
        assert k in cs
    assert cs["daily_total_internal"] == json.loads(str(tr["dta_params"]))["P_int"] and cs["eta_service_food"] is None
