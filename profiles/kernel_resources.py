#!/usr/bin/env python3
"""Register / scratch / LDS budget of the replanning kernels as hipcc reports them (-Rpass-analysis=kernel-resource-usage,
device code only; no GPU needed):

    python profiles/kernel_resources.py > profiles/r03_kernel_resources.csv
"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "trafficsimulation_amd", "csrc")
WANT = ["k_replan", "k_replan_quad", "quad_policy", "replan_turn", "k_astar_single", "k_spawn_plan", "k_decide_main", "k_decide_pre",
        "k_move_claim", "k_move_resolve", "k_amap_build"]
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-pthread", "--offload-arch=gfx950", "--cuda-device-only", "-c",
                    "-o", "/dev/null", "engine.hip", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:], cwd=CSRC, capture_output=True, text=True)
cur, rows = None, {}
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        # (mangled names of the anonymous namespace: _ZN12_GLOBAL__N_1<len><name>...)
        n = re.match(r"_ZN12_GLOBAL__N_1(\d+)", m.group(1))
        cur = m.group(1)[len(n.group(0)):len(n.group(0)) + int(n.group(1))] if n else m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
print("function,VGPRs,AGPRs,SGPRs,scratch_bytes_per_lane,occupancy_waves_per_SIMD,LDS_bytes_per_block,SGPR_spills,VGPR_spills")
for k in WANT:
    if k in rows:
        v = rows[k]
        print(",".join([k, v.get("VGPRs", ""), v.get("AGPRs", ""), v.get("TotalSGPRs", ""), v.get("ScratchSize", ""), v.get("Occupancy", ""),
                        v.get("LDS Size", ""), v.get("SGPRs Spill", ""), v.get("VGPRs Spill", "")]))
