/*
 * trafficsim.h - C-ABI of the MI355X-native per-timestep agent-update engine.
 *
 * This is the drop-in boundary for the hot path of kurisu-n/TrafficSimulation
 * (SURVEY.md §8(b)): everything at or below `CityModel.step()`.  The reference has no
 * C-ABI of its own (it is pure Python + one unused pybind11 file), so every entry point
 * cites the reference *Python* interface it replaces (paths relative to
 * /root/reference/Simulation).  The reference-side binding (ctypes) is shown in
 * INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch/numpy types.
 *   - maps are C-contiguous (H, W), indexed [y*W + x], exactly like the reference's numpy
 *     maps (city_model.py:109-115).  Coordinates are (x, y) pairs like the reference's tuples.
 *   - every call returns 0 on success or a negative TS_E_* code; `ts_last_error` gives the
 *     text.  (The reference raises Python exceptions; the facade re-raises from the code.)
 *   - the caller owns every buffer it passes in; the engine copies in/out and owns all
 *     device memory.  No pointer returned by the engine outlives the handle.
 *   - one caller thread per handle (the reference serialises model.step() with the UI on the
 *     Tornado IOLoop, SURVEY.md §8(b) "Threading").
 *
 * The same function set is implemented twice:
 *   ts_*   libtrafficsim_hip.so   hand-written HIP kernels for gfx950 (the product)
 *   tso_*  oracle/libtso.so       single-threaded CPU restatement (test infrastructure only)
 */
#ifndef TRAFFICSIM_H
#define TRAFFICSIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ts_engine* ts_handle;

enum {
  TS_OK = 0,
  TS_E_INVALID = -1,     /* bad argument (shape, range, null) */
  TS_E_STATE = -2,       /* call order (e.g. step before seeding)  */
  TS_E_DEVICE = -3,      /* HIP runtime error / no gfx950 device */
  TS_E_UNSUPPORTED = -4, /* a Defaults combination the engine does not implement */
  TS_E_CAPACITY = -5     /* a fixed-size device pool overflowed */
};

/* Defaults.TRAFFIC_LIGHT_AGENT_ALGORITHM (config.py:341-347); RL variants are out of scope. */
enum {
  TS_LIGHTS_DISABLED = 0,
  TS_LIGHTS_FIXED_TIME = 1,
  TS_LIGHTS_QUEUE_ACTUATED = 2,
  TS_LIGHTS_PRESSURE_CONTROL = 3,
  TS_LIGHTS_NEIGHBOR_PRESSURE_CONTROL = 4,
  TS_LIGHTS_NEIGHBOR_GREEN_WAVE = 5
};

/* The subset of config.py `Defaults` that the hot path reads.  Field names follow the
 * reference's attribute names (lower-cased).  Filled by ts_default_params() with the values
 * of config.py. */
typedef struct TsParams {
  /* vehicles (config.py:275-326) */
  int32_t vehicle_min_speed;                         /* VEHICLE_MIN_SPEED = 1 */
  int32_t vehicle_max_speed;                         /* VEHICLE_MAX_SPEED = 5 */
  int32_t vehicle_awareness_range;                   /* VEHICLE_AWARENESS_RANGE = 10 */
  int32_t rain_enabled;                              /* RAIN_ENABLED */
  int32_t rain_speed_reduction;                      /* RAIN_SPEED_REDUCTION = 2 */
  int32_t pathfinding_cooldown;                      /* PATHFINDING_COOLDOWN = 5 */
  int32_t pathfinding_cache;                         /* PATHFINDING_CACHE = True */
  int32_t stuck_recompute_threshold;                 /* VEHICLE_STUCK_RECOMPUTE_THRESHOLD = 30 */
  int32_t stuck_recompute_threshold_intersection;    /* ..._INTERSECTION = 1 */
  int32_t contraflow_overtake_active;                /* VEHICLE_CONTRAFLOW_OVERTAKE_ACTIVE */
  int32_t max_contraflow_overtake_steps;             /* VEHICLE_MAX_CONTRAFLOW_OVERTAKE_STEPS = 6 */
  int32_t contraflow_overtake_duration;              /* VEHICLE_CONTRAFLOW_OVERTAKE_DURATION = 30 */
  int32_t stuck_contraflow_enabled;                  /* VEHICLE_STUCK_CONTRAFLOW_ENABLED */
  int32_t stuck_contraflow_threshold;                /* = 60 */
  int32_t stuck_contraflow_threshold_intersection;   /* = 10 */
  int32_t max_contraflow_stuck_detour_steps;         /* = 20 */
  int32_t contraflow_stuck_detour_duration;          /* = 10 */
  int32_t malfunction_active;                        /* VEHICLE_MALFUNCTION_ACTIVE */
  int32_t malfunction_duration;                      /* = 400 */
  int32_t sideswipe_active;                          /* VEHICLE_SIDESWIPE_COLLISION_ACTIVE */
  int32_t sideswipe_duration;                        /* = 600 */
  double malfunction_chance;                         /* = 1e-7 */
  double sideswipe_chance;                           /* = 1e-9 */
  /* A* penalties, captured at import time by astar_numba.py:11-24 */
  int32_t contraflow_penalty;                        /* VEHICLE_CONTRAFLOW_PENALTY = 5000 */
  int32_t obstacle_penalty_vehicle;                  /* = 1000 */
  int32_t obstacle_penalty_stop;                     /* = 500 */
  int32_t road_type_penalties_enabled;               /* True */
  int32_t turn_penalty_enabled;                      /* True */
  int32_t turn_penalty;                              /* = 10 */
  int32_t dynamic_penalties_enabled;                 /* True */
  int32_t _pad0;
  double road_type_penalty_r1;                       /* 0.5 */
  double road_type_penalty_r2;                       /* 5 */
  double road_type_penalty_r3;                       /* 50 */
  double dynamic_penalty_scale;                      /* 4.0 */
  /* light groups (config.py:338-362) */
  int32_t light_algorithm;                           /* TS_LIGHTS_* */
  int32_t transition_duration_enabled;               /* False */
  int32_t transition_clearance_enabled;              /* True */
  int32_t all_red_duration;                          /* 2 */
  int32_t green_duration;                            /* TRAFFIC_LIGHT_GREEN_DURATION = 20 */
  int32_t qa_min_green;                              /* 5 */
  int32_t qa_max_green;                              /* 30 */
  int32_t qa_gap;                                    /* 3 */
  /* model */
  int32_t enable_traffic;                            /* ENABLE_TRAFFIC (trip statistics) */
  int32_t time_per_step_seconds;                     /* TIME_PER_STEP_IN_SECONDS = 6 */
  int32_t eager_density;                             /* 1: recompute density_map every tick like
                                                        city_model.py:1853; 0: only when a soft A*
                                                        needs it (results are identical) */
  /* rain clouds (config.py:262-271), used once a TS_AGENT_RAIN_MANAGER is scheduled */
  int32_t rain_radius_min;                           /* RAIN_RADIUS_MIN = 50 */
  int32_t rain_radius_max;                           /* RAIN_RADIUS_MAX = 100 */
  int32_t rain_occurrences_max;                      /* RAIN_OCCURRENCES_MAX = 3 */
  int32_t rain_cooldown;                             /* RAIN_COOLDOWN = 86400 (seconds; dead code in the reference) */
  int32_t rain_spawn_offset;                         /* RAIN_SPAWN_OFFSET = 10 */
  double rain_spawn_chance;                          /* RAIN_SPAWN_CHANCE = 0.1 */
  /* _despawn_check (vehicle_base.py:695-706): a vehicle whose stuck_ticks reach the threshold (the smaller one on an
   * intersection cell) leaves the model at the end of its step and counts as errored_internal / errored_through */
  int32_t stuck_despawn_enabled;                     /* VEHICLE_STUCK_DESPAWN_ENABLED = False */
  int32_t stuck_despawn_threshold;                   /* VEHICLE_STUCK_DESPAWN_THRESHOLD = 3600 */
  int32_t stuck_despawn_threshold_intersection;      /* VEHICLE_STUCK_DESPAWN_THRESHOLD_INTERSECTION = 20 */
  /* VEHICLE_RESPECT_AWARENESS (config.py:278, astar_numba.py:29-50): occupied / red cells only count as obstacles of a
   * search inside the field of view cast from its start cell (straight road runs, vehicle_awareness_range wide) */
  int32_t respect_awareness;                         /* VEHICLE_RESPECT_AWARENESS = False */
  /* PATHFINDING_BATCHING (config.py:411, city_model.py:1855, vehicle_base.py:669): 1 = every active vehicle runs step_decide
   * before the schedule is shuffled (run_parallel_decide); 0 = each vehicle runs it at the top of its own step(), in the
   * shuffled order, on the maps and the global stream as the agents stepped before it left them */
  int32_t pathfinding_batching;                      /* PATHFINDING_BATCHING = True */
} TsParams;

/* Static maps produced by world-gen (`_build_simple_maps`, city_model.py:2151-2199). */
typedef struct TsWorld {
  int32_t width, height;
  const uint8_t* allowed_dirs_map; /* bitmask N=1,E=2,S=4,W=8 (city_model.py:2190-2197) */
  const int8_t* is_road_map;
  const int8_t* road_type_map;     /* {0,1,2,3} */
  const int8_t* intersection_map;
} TsWorld;

/* Light-group topology (`_create_intersection_light_groups`, city_model.py:1587-1650 and
 * IntersectionLightGroup.initialize_cached_lane_coords / populate_links,
 * intersection_light_group.py:118-171, 175-279).  Ragged tables as (offsets[n+1], values). */
typedef struct TsLightTables {
  int32_t n_groups, n_lights;
  const int32_t* g_light_off;    /* [G+1] lights of group g = [off[g], off[g+1]) (traffic_lights order) */
  const int32_t* light_xy;       /* [L*2] light cell */
  const int32_t* light_ctrl_off; /* [L+1] controlled_blocks of each light */
  const int32_t* light_ctrl_xy;
  const int32_t* g_ns_off;       /* opposite_pairs["N-S"]: global light indices */
  const int32_t* g_ns;
  const int32_t* g_ew_off;       /* opposite_pairs["W-E"] */
  const int32_t* g_ew;
  const int32_t* g_icell_off;    /* intersection_cells */
  const int32_t* g_icell_xy;
  const int32_t* g_ns_in_off;    /* ns_in_coords ... ew_out_coords (118-171) */
  const int32_t* g_ns_in_xy;
  const int32_t* g_ns_out_off;
  const int32_t* g_ns_out_xy;
  const int32_t* g_ew_in_off;
  const int32_t* g_ew_in_xy;
  const int32_t* g_ew_out_off;
  const int32_t* g_ew_out_xy;
  const int32_t* g_neighbors;    /* [G*4*2] (dir code N0 E1 S2 W3 or -1, group index): neighbor_groups
                                    after populate_links() has been re-run on the finished model */
  const int32_t* g_neighbors_ctor; /* same layout, as left by the constructor: populate_links() runs
                                    before the model assigns cell.intersection_group
                                    (city_model.py:1639-1650), so opposite_pairs is empty and only
                                    earlier groups are visible until the group's first
                                    _execute_phase_change reaches get_opposite_traffic_lights()
                                    (intersection_light_group.py:303-307, 369), which re-populates.
                                    NULL = same as g_neighbors. */
} TsLightTables;

/* Kinds of non-vehicle entries in the Mesa schedule, appended in insertion order
 * (city_model.py:1642, 1738, 200, 204).  Their presence matters: RandomActivation shuffles
 * ALL scheduled agents each tick. */
enum {
  TS_AGENT_LIGHT_GROUP = 0, /* next IntersectionLightGroup, in table order */
  TS_AGENT_NOOP = 1,        /* CityBlock: occupies a shuffle slot only */
  TS_AGENT_RAIN_MANAGER = 2,/* RainManager (rain.py:86-184): spawns RainAgents (each a schedule entry of its own,
                               added and removed by the engine) and writes rain_map at its place in the order */
  TS_AGENT_CLOCK = 3,       /* DynamicTrafficAgent: elapsed += dt (dynamic_traffic_generator.py:153-155); spawns once
                               ts_set_traffic_generator armed it */
  TS_AGENT_CITY_BLOCK = 6   /* next CityBlock in city_blocks order: food / waste bookkeeping (city_block.py:148-150) */
};

/* DynamicTrafficAgent (dynamic_traffic_generator.py:71-150, 307-430): the daily trip schedule and the mid-tick
 * spawning of internal / through vehicles.  Service vehicles are not covered (their quotas must be 0). */
typedef struct TsTrafficZone {      /* one entry of Defaults.TIME_ZONES (config.py:155-236) */
  int32_t start_hour, end_hour;
  double through_distribution;
  int32_t n_internal;               /* entries of internal_distribution, in dict order */
  int32_t origin_type[8], dest_type[8]; /* index into Defaults.AVAILABLE_CITY_BLOCKS (Res 0, Off 1, Mar 2, Lei 3, Oth 4) */
  double fraction[8];
} TsTrafficZone;
typedef struct TsTrafficTables {
  int32_t n_blocks;                 /* city_blocks in dict order (what get_city_blocks_by_type iterates) */
  const int32_t* blk_type;          /* [n_blocks] */
  const int32_t* blk_entr_off;      /* [n_blocks+1] CityBlock.get_entrances() */
  const int32_t* blk_entr_xy;
  int32_t n_highway_entrances;      /* city.get_highway_entrances() / get_highway_exits() */
  const int32_t* highway_entrances_xy;
  int32_t n_highway_exits;
  const int32_t* highway_exits_xy;
  int32_t internal_population_per_day;  /* INTERNAL_POPULATION_TRAFFIC_PER_DAY */
  int32_t passing_population_per_day;   /* PASSING_POPULATION_TRAFFIC_PER_DAY */
  int32_t start_offset_seconds;         /* SIMULATION_STARTING_TIME_OF_DAY_* in seconds */
  int32_t n_zones;
  TsTrafficZone zones[8];
  /* service vehicles + CityBlock resources (vehicle_service.py, city_block.py); all zero = not used */
  int32_t total_service_vehicles_food;  /* TOTAL_SERVICE_VEHICLES_FOOD = 50 */
  int32_t total_service_vehicles_waste; /* TOTAL_SERVICE_VEHICLES_WASTE = 50 */
  int32_t service_load_time;            /* SERVICE_VEHICLE_LOAD_TIME = 20 */
  int32_t gradual_city_block_resources; /* GRADUAL_CITY_BLOCK_RESOURCES = True */
  int32_t food_consumption_ticks;       /* FOOD_CONSUMPTION_TICKS = 50 */
  int32_t waste_production_ticks;       /* WASTE_PRODUCTION_TICKS = 100 */
  int32_t needs_food_type_mask;         /* bit t set: block type t is in CITY_BLOCK_THAT_NEED_FOOD (Market, Leisure) */
  int32_t produces_waste_type_mask;     /* bit t set: block type t is in CITY_BLOCK_THAT_PRODUCE_WASTE (all five) */
  double service_max_load_food;         /* SERVICE_VEHICLE_MAX_LOAD_FOOD = 50 */
  double service_max_load_waste;        /* SERVICE_VEHICLE_MAX_LOAD_WASTE = 250 */
  double food_capacity_per_cell;        /* FOOD_CAPACITY_PER_CELL = 2 */
  double waste_capacity_per_cell;       /* WASTE_CAPACITY_PER_CELL = 1.5 */
  const int32_t* blk_inner_cells;       /* [n_blocks] len(CityBlock._inner_blocks) */
  const int32_t* blk_service_off;       /* [n_blocks+1] CityBlock.get_service_road_cell's ranked candidates */
  const int32_t* blk_service_xy;        /*   (city_block.py:152-190: static; ties in CPython set order, so the
                                             list is an input recorded from the interpreter, not recomputed) */
  int32_t statistics_update_interval;   /* STATISTICS_UPDATE_INTERVAL = 20 (config.py:428); 0 = 20 */
} TsTrafficTables;

/* dynamic_traffic_generator.py:102-131 counters that the hot path writes. */
typedef struct TsCounters {
  int64_t stuck, collisions, malfunctions, overtaking, in_stuck_detour, parked;
  int64_t live_internal, live_through;
  int64_t count_completed_internal, count_completed_through;
  int64_t total_distance_internal, total_distance_through;
  int64_t errored_internal, errored_through;
  double total_duration_internal, total_duration_through;
  double elapsed;
  /* engine statistics (not in the reference) */
  int64_t step_count;          /* CityModel.step_count */
  int64_t agent_steps;         /* sum over ticks of vehicles stepped (the benchmark unit) */
  int64_t astar_calls, astar_expansions, astar_relaxations;
  int64_t move_rounds;         /* dependency-resolution rounds executed by the move phase */
  int64_t rng_fixups;          /* decide-phase re-scans caused by malfunction/collision events */
  int64_t created_internal, created_through; /* DynamicTrafficAgent.created_* (reset at day rollover) */
  int64_t created_service_food, created_service_waste, live_service_food, live_service_waste;
} TsCounters;

/* DynamicTrafficAgent._update_cached_stats (dynamic_traffic_generator.py:525-648): what the generator gathers every
 * STATISTICS_UPDATE_INTERVAL ticks INSIDE its own step - at its place in the shuffled order, after that tick's spawns - and
 * the statistics panel reads between updates (ui_modules/traffic_statistics.py:58-239).  The sums over the scheduled vehicles
 * are raw here (a device reduction at the generator's sync point of the move phase); the facade forms the reference's
 * quotients and keys from them (mesa_api: cached_stats).  index 0..3 = internal, through, service_food, service_waste. */
typedef struct TsCachedStats {
  int32_t valid;                 /* 0 until the first update: the reference's dict is empty until then */
  int32_t pad_;
  int64_t update_step;           /* CityModel.step_count of the tick whose generator step took the snapshot (0-based) */
  double dur_live[2];            /* sum of (elapsed - depart_time) over scheduled vehicles of population internal / through */
  int64_t dist_live[2], n_live[2];   /* sum of steps_traveled, number of them */
  int64_t stuck_ticks_sum, stuck_ticks_max;   /* over the scheduled vehicles with is_stuck */
  int64_t stuck, collisions, malfunctions, parked, overtaking, in_stuck_detour;   /* the generator's counters at that moment */
  int64_t live_internal, live_through, live_service_food, live_service_waste;
  int64_t count_completed[2], total_distance[2];
  double total_duration[2];
  int64_t daily_total[4], created[4], errored[4];
  double eta[4];                 /* next_service_eta(kind), NaN = None */
  double avg_daily_difference;
} TsCachedStats;
int ts_cached_stats(ts_handle h, TsCachedStats* out);

/* One row per live vehicle, in `active_vehicle_agents` order (city_model.py:1903). */
enum {
  TS_V_SPAWN_IDX = 0, TS_V_X, TS_V_Y, TS_V_BASE_SPEED, TS_V_CURRENT_SPEED, TS_V_MAX_STEPS,
  TS_V_DIRECTION,      /* N0 E1 S2 W3, -1 = None */
  TS_V_STUCK_TICKS, TS_V_COOLDOWN, TS_V_FLAGS, TS_V_STRANDED_LEFT, TS_V_STEPS_TRAVELED,
  TS_V_PATH_LEN, TS_V_PATH_CRC, /* crc32 of the remaining path as int32 (x,y) pairs, 0 if empty */
  TS_V_OVERTAKE_DUR, TS_V_DETOUR_DUR,
  TS_V_NFIELDS
};
enum {
  TS_F_EARLY_EXIT = 1, TS_F_STUCK = 2, TS_F_PARKED = 4, TS_F_COLLISION = 8, TS_F_MALFUNCTION = 16,
  TS_F_OVERTAKING = 32, TS_F_DETOUR = 64, TS_F_BLOCKED = 128,
  TS_F_HAS_PREV = 256 /* previous_pos == pos (vehicle_base.py:688) */
};
/* One row per light group. */
enum {
  TS_G_CURRENT_PHASE = 0, TS_G_PENDING_PHASE, /* -1 = None */
  TS_G_QUEUE_TIMER, TS_G_GAP_TIMER, TS_G_LAST_ARRIVAL, TS_G_FIXED_TIME_TIMER, TS_G_FT_PHASE,
  TS_G_NS_PRESSURE, TS_G_EW_PRESSURE,
  TS_G_NFIELDS
};
enum { TS_MAP_OCCUPANCY = 0, TS_MAP_STOP = 1, TS_MAP_STUCK = 2, TS_MAP_RAIN = 3 };
enum { TS_RNG_GLOBAL = 0, TS_RNG_SCHEDULER = 1 };
enum { TS_POP_UNDEFINED = 0, TS_POP_INTERNAL = 1, TS_POP_THROUGH = 2 };
/* Trip kinds of the traffic generator beyond the two population types */
enum { TS_TRIP_SERVICE_FOOD = 3, TS_TRIP_SERVICE_WASTE = 4 };

/* config.py defaults. */
void ts_default_params(TsParams* p);

/* CityModel.__init__ after world-gen (city_model.py:109-115, 147-148): allocate the dynamic
 * maps and take a copy of the static ones. */
int ts_create(const TsWorld* world, const TsParams* params, ts_handle* out);
int ts_destroy(ts_handle h);
const char* ts_last_error(ts_handle h);

/* IntersectionLightGroup construction (city_model.py:1639-1650): registers the groups.  With
 * an algorithm other than DISABLED every group starts with pending_phase = 0
 * (intersection_light_group.py:115-116).  Does NOT add them to the schedule. */
int ts_set_lights(ts_handle h, const TsLightTables* t);

/* schedule.add() for non-vehicle agents, in insertion order (city_model.py:1642, 1738, 200, 204). */
int ts_schedule_add(ts_handle h, int32_t kind, int32_t count);

/* DynamicTrafficAgent.__init__ (dynamic_traffic_generator.py:71-150): arms the spawner behind the TS_AGENT_CLOCK
 * schedule entry and generates day 0, which DRAWS FROM THE GLOBAL STREAM (random(), choice()) - call it after
 * seeding TS_RNG_GLOBAL, at the point where the reference constructs the agent (end of CityModel.__init__). */
int ts_set_traffic_generator(ts_handle h, const TsTrafficTables* t);

/* random.setstate() for the two MT19937 streams: TS_RNG_GLOBAL = module-level `random`
 * (vehicle_base.py:112, 600, 609), TS_RNG_SCHEDULER = model.random (city_model.py:55, 1858).
 * `mt` is random.getstate()[1][:624], `index` is [624]. */
int ts_seed(ts_handle h, int32_t stream, const uint32_t* mt, uint32_t index);
/* random.seed(int) / random.Random(int) for a non-negative integer seed. */
int ts_seed_int(ts_handle h, int32_t stream, uint64_t seed);
int ts_rng_state(ts_handle h, int32_t stream, uint32_t* mt_out, uint32_t* index_out);

/* VehicleAgent(custom_id, model, start_cell, target_cell, population_type) for n vehicles, in
 * order (vehicle_base.py:29-89 -> city_model.place_vehicle 1897-1918 -> _compute_path 143-167).
 * path_off == NULL: the initial path is computed as the reference does (cache, then the
 * phase 1-4 planner) on the maps as they are at that moment.  Otherwise vehicle i's initial
 * path is path_xy[2*path_off[i] .. 2*path_off[i+1]) (4-adjacent chain starting next to the
 * start cell).  The density map must exist first in the reference; here it is
 * produced on demand. */
int ts_add_vehicles(ts_handle h, int32_t n, const int32_t* start_xy, const int32_t* goal_xy,
                    const int32_t* population_type, const int32_t* path_off, const int32_t* path_xy);

/* Same as ts_add_vehicles with the initial paths given as direction codes (N0 E1 S2 W3), one byte
 * per step: vehicle i's path is path_dirs[path_off[i] .. path_off[i+1]).  An engine-side extension
 * (the reference has no such entry) so that 10^6 synthetic routes do not travel as 8-byte (x, y) pairs. */
int ts_add_vehicles_dirs(ts_handle h, int32_t n, const int32_t* start_xy, const int32_t* goal_xy,
                         const int32_t* population_type, const int64_t* path_off, const uint8_t* path_dirs);

/* CityModel.remove_vehicle (city_model.py:1920-1941) called by the host between ticks: vehicle `spawn_idx` (the creation
 * index that ts_download_vehicles reports in column 0) leaves the maps (occupancy and stuck_map of its cell are cleared,
 * even if another vehicle shares the cell), the cell's MultiGrid list, the schedule and active_vehicle_agents.  The live
 * counters follow the CALLER's `population_type` argument like the reference's do (1935-1941: 'internal' -> live_internal,
 * 'through' -> live_through, the default 'undefined' -> neither), not the vehicle's own population: pass TS_POP_INTERNAL /
 * TS_POP_THROUGH / TS_POP_UNDEFINED.  As in the reference nothing else is touched (a stuck or parked vehicle stays counted
 * in `stuck` / `parked`).  TS_E_INVALID: no such live vehicle; TS_E_UNSUPPORTED: a service vehicle (its block / load
 * bookkeeping lives in ServiceVehicleAgent, which has no such entry in the reference). */
int ts_remove_vehicle(ts_handle h, int32_t spawn_idx, int32_t population_type);

/* Host writes between ticks (UI handlers / RainManager): whole-map upload of stop_map or
 * rain_map (cell.py:241-251, rain.py:156-184). */
int ts_upload_map(ts_handle h, int32_t which, const int8_t* src);

/* CityModel.step() x n_ticks (city_model.py:1831-1860). */
int ts_step(ts_handle h, int32_t n_ticks);

int ts_num_vehicles(ts_handle h);  /* len(active_vehicle_agents) */
int ts_num_groups(ts_handle h);
int ts_num_scheduled(ts_handle h); /* len(schedule._agents) */
int ts_download_map(ts_handle h, int32_t which, int8_t* dst);
int ts_download_density(ts_handle h, float* dst); /* _update_density_map (1764-1778), recomputed now */
int ts_download_vehicles(ts_handle h, int32_t* rows, int32_t cap_rows); /* [n][TS_V_NFIELDS] */
/* What a vehicle *is* (constant or slowly changing; vehicle_base.py:29-40, vehicle_service.py:27-41), one row per live
 * vehicle in the order of ts_download_vehicles: what the UI needs for vehicles the traffic generator created. */
enum {
  TS_M_SPAWN_IDX = 0, TS_M_POPULATION /* TS_POP_* */, TS_M_TARGET_X, TS_M_TARGET_Y,
  TS_M_VEHICLE_TYPE /* 0 plain, TS_TRIP_SERVICE_FOOD, TS_TRIP_SERVICE_WASTE */,
  TS_M_SERVICE_PHASE /* -1 none, 0 "to_block", 1 "servicing", 2 "to_exit" */,
  TS_M_NFIELDS
};
int ts_num_spawned(ts_handle h); /* vehicles ever placed = the next spawn index (ts_add_vehicles and the generator share it) */
int ts_download_vehicle_meta(ts_handle h, int32_t* rows, int32_t cap_rows); /* [n][TS_M_NFIELDS], returns n */
/* ServiceVehicleAgent.current_load / max_load / current_block of every live service vehicle; returns their number.
 * spawn_idx [n], loads [n][2], block [n] (index into city_blocks, -1 = None) */
int ts_download_service_vehicles(ts_handle h, int32_t* spawn_idx, double* loads, int32_t* block, int32_t cap);
/* remaining path of the vehicle at position `active_pos`; returns its length (cells). */
int ts_download_path(ts_handle h, int32_t active_pos, int32_t* xy, int32_t cap_cells);
int ts_download_groups(ts_handle h, int32_t* rows); /* [G][TS_G_NFIELDS] */
/* CityBlock.get_food_units() / get_waste_units() per block, city_blocks order (city_block.py:100-101); rows [n][2] */
int ts_num_blocks(ts_handle h);
int ts_download_blocks(ts_handle h, double* rows);
int ts_counters(ts_handle h, TsCounters* out);
/* IntersectionLightGroup.get_opposite_traffic_lights() (intersection_light_group.py:303-307), which the UI's
 * SetOppGo / SetOppStop handlers call (ui_modules/traffic_light_control.py:336-362): while the group's opposite_pairs
 * are still empty it re-runs populate_links(), after which the NEIGHBOR_* controllers read the re-populated
 * neighbour table (TsLightTables::g_neighbors instead of g_neighbors_ctor).  repopulate != 0 performs that side
 * effect; the return value is 1 if the group's links have been re-populated (by this call, an earlier one or the
 * group's first phase change), else 0. */
int ts_group_links(ts_handle h, int32_t group, int32_t repopulate);
/* CreateServiceVehicleHandler (visualization/ui_modules/vehicle_control.py:182-206): ServiceVehicleAgent(vid, model,
 * entrance, sv_type) created from the UI between ticks - vehicle_service.py:19-41 as it is, without the generator's id
 * draw (the handler numbers its vehicles itself).  service_type: TS_TRIP_SERVICE_FOOD / TS_TRIP_SERVICE_WASTE.
 * Needs the block tables of ts_set_traffic_generator. */
int ts_add_service_vehicle(ts_handle h, int32_t x, int32_t y, int32_t service_type);
/* RainControl card and the /spawn_rain handler (visualization/ui_modules/rain_control.py:22-73):
 * len(model.rains), RainManager.cooldown / .counter, and RainManager.add_random_rain() called between ticks
 * (rain.py:100-148: draws from the global stream, appends the cloud to city_model.rains and to the schedule). */
typedef struct TsRainInfo { int32_t has_manager, n_rains, cooldown, counter; } TsRainInfo;
int ts_rain_info(ts_handle h, TsRainInfo* out);
int ts_rain_spawn(ts_handle h);

/* The pathfinder operator seam: astar(width, height, sx, sy, gx, gy, occupancy_map, stop_map,
 * is_road_map, road_type_map, allowed_dirs_map, respect_awareness=False, awareness_range,
 * density_map, soft_obstacles, ignore_flow, maximum_steps) -> [(x, y)]  (astar_numba.py:243-281),
 * evaluated on the engine's current maps.  Returns the path length (0 = no path) or <0. */
int ts_astar(ts_handle h, int32_t sx, int32_t sy, int32_t gx, int32_t gy, int32_t soft_obstacles,
             int32_t ignore_flow, int32_t maximum_steps, int32_t* out_xy, int32_t cap_cells);

/* Select the HIP device used by subsequent ts_create calls in this process (one process per GPU:
 * rank r of a multi-GPU launch calls ts_set_device(LOCAL_RANK)). */
int ts_set_device(int32_t device);

/* Multi-GPU, replicated state / sharded replans (the all-gather variant): every rank of a
 * one-process-per-GPU job holds the whole world and is stepped with the same calls and seeds; the replanning searches
 * of a tick - 99 % of a default-policy tick - are split by decide-order index (index % world == rank), and the ranks
 * trade what those step_decides changed (paths, timers, flags, counters) through `exchange` once per tick, so that
 * every rank ends the tick with the state a single GPU would have: bit for bit (tests/test_gpu_dist.py).  The
 * reference has no distributed code; the contract is this build's.
 * `exchange(user, send, send_bytes, &recv, &sizes, &stride)` is an all-gather of variable-size byte buffers in host
 * memory: on return recv points at world slots of `stride` bytes (slot r = rank r's buffer, sizes[r] bytes of it
 * valid), owned by the callee until the next call.  It returns 0 or a negative error.  Over torch.distributed it
 * is an all_gather on RCCL (backend "nccl") or gloo: trafficsimulation_amd/dist.py.  world == 1 switches it off. */
typedef int (*ts_exchange_fn)(void* user, const void* send, int64_t send_bytes, void** recv, int64_t** sizes, int64_t* stride);
int ts_set_replan_sharding(ts_handle h, int32_t rank, int32_t world, ts_exchange_fn exchange, void* user);
/* The same with the buffers in DEVICE memory (the form for RCCL): `send` is a device pointer to this rank's packed records
 * (valid until the call returns), `recv` must come back as a device pointer to world slots of `stride` bytes each, readable
 * by the engine's stream when the call returns and owned by the callee until the next call; `sizes` stays a host array.
 * The payload never visits the host: one all_gather_into_tensor between pre-sized device buffers
 * (trafficsimulation_amd/dist.py: ShardedReplans(device_direct=True)). */
int ts_set_replan_sharding_device(ts_handle h, int32_t rank, int32_t world, ts_exchange_fn exchange, void* user);

/* Per-kernel timing with HIP events recorded on the engine's own stream (bench.py's roofline leg).
 * ts_profile_enable(h, 1) starts collecting; ts_profile_get returns, for kernel class `kernel_id`
 * (0 <= id < ts_profile_count()), the summed device time in ms, the number of launches and the
 * number of work items (agents / cells) those launches covered. */
int ts_profile_enable(ts_handle h, int32_t on);
int ts_profile_count(void);
const char* ts_profile_name(int32_t kernel_id);
int ts_profile_get(ts_handle h, int32_t kernel_id, double* total_ms, int64_t* launches, int64_t* items);

/* Test hook: overwrite occupancy_map without placing vehicles (A* / density known-answer tests). */
int ts_debug_set_occupancy(ts_handle h, const int8_t* src);

#ifdef __cplusplus
}
#endif
#endif /* TRAFFICSIM_H */
