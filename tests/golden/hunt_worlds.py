"""Differential hunt for trafficsimulation_amd/worldgen.py: random CityModel constructor configurations built by the
reference itself (importable only in the build container: /root/reference + tests/golden/standins) and by worldgen,
every table and the stream state compared.  Not a test (nothing here travels to the GPU box); DESIGN.md §7 quotes the
totals.  usage: python tests/golden/hunt_worlds.py FIRST_CASE N_CASES"""
import sys, os, json, random, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, HERE)
import make_golden as mg
mg._setup_paths()
from Simulation.config import Defaults
Defaults.SAVE_TOTAL_RESULTS=False; Defaults.SAVE_INDIVIDUAL_RESULTS=False; Defaults.RAIN_ENABLED=False; Defaults.ENABLE_TRAFFIC=False
from Simulation.city_model import CityModel
from trafficsimulation_amd.worldgen import generate_world
first=int(sys.argv[1]); n=int(sys.argv[2])
bad=0
for case in range(first, first+n):
    pr=random.Random(9000+case)
    w=pr.choice([40,56,64,80,96,128]); h=w if pr.random()<0.6 else pr.choice([48,64,90,120])
    kw={}
    if pr.random()<0.5: kw['ring_road_type']=pr.choice(['R1','R2','R3',None])
    if pr.random()<0.3: kw['optimized_intersections']=False
    if pr.random()<0.4:
        kw['carve_subblock_roads']=True; kw['subblock_chance']=pr.choice([0.3,0.7,1.0])
        if pr.random()<0.4: kw['subblock_roads_have_intersections']=False
        if pr.random()<0.4: kw['min_subblock_spacing']=pr.choice([2,3,4])
        if pr.random()<0.15: kw['subblock_road_type']='R2'
    if pr.random()<0.3:
        kw['forward_traffic_light_range']=True; kw['forward_traffic_light_range_intersections']=pr.choice(["Skip","Include in Range","Include as Extra"])
    if pr.random()<0.3: kw['traffic_light_range']=pr.choice([0,1,3,6,20])
    if pr.random()<0.3: kw['wall_thickness']=pr.choice([3,6,10,15]); kw['sidewalk_ring_width']=pr.choice([1,2,3])
    if pr.random()<0.3: kw['highway_offset_from_edges']=pr.choice([0,2,5,10])
    if pr.random()<0.3: kw['min_r1_bands']=pr.choice([0,1,2,3])
    if pr.random()<0.3: kw['min_block_spacing']=pr.choice([3,4,6]); kw['max_block_spacing']=pr.choice([6,9,14,24])
    if pr.random()<0.3: kw['r1_chance_mean']=pr.choice([0.0,0.1,0.4]); kw['r2_chance_mean']=pr.choice([0.1,0.5,0.9])
    lvl=pr.choice([0,0,1,2]); Defaults.BLOCK_ENTRANCE_ROAD_LEVEL=lvl
    seed=5000+case
    random.seed(seed)
    ref_exc=None
    try:
        m=CityModel(width=w,height=h,seed=seed,**kw); t=mg.world_tables(m); t['global_rng_state']=np.asarray(random.getstate()[1],dtype=np.uint32)
    except Exception as e:
        ref_exc=type(e).__name__
    my_exc=None
    try:
        g=generate_world(w,h,seed=seed,rain_enabled=False,enable_traffic=False,block_entrance_road_level=lvl,**kw)
    except Exception as e:
        my_exc=type(e).__name__
    if ref_exc or my_exc:
        st='OK' if ref_exc==my_exc else 'EXC-MISMATCH'
        if st!='OK': bad+=1
        print(case,w,h,kw,lvl,'ref',ref_exc,'mine',my_exc,st, flush=True); continue
    diff=[k for k in g if not (np.asarray(g[k]).shape==np.asarray(t[k]).shape and np.array_equal(g[k],t[k]))]
    if diff: bad+=1
    print(case,w,h,kw,lvl,'OK' if not diff else diff, flush=True)
print('bad',bad)
