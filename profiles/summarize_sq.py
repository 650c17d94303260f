#!/usr/bin/env python3
"""Condense rocprofv3 SQ counter passes of one bench command into profiles/r03_sq_*.json:

    summarize_sq.py <out.json> <bench line .json> "<command text>" <pass1 results.db> [<pass2 results.db> ...]

Per kernel family (k_replan, k_replan_quad): every counter summed over the process' launches, and divided by the A*
expansions the bench line reports for the same launches (run the bench with --warmup 0 so that its counters cover them all)."""
import collections, json, re, sqlite3, sys

out_path, bench_path, command = sys.argv[1:4]
line = [l for l in open(bench_path).read().splitlines() if l.startswith("{")][-1]
b = json.loads(line)
exp = b["config"]["astar"]["expansions"]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(int))
for db in sys.argv[4:]:
    c = sqlite3.connect(db)
    for name, counter, val in c.execute("select kernel_name, counter_name, value from counters_collection"):
        m = re.search(r"\b(k_[a-z_0-9]+)\(", name) or re.search(r"\b(k_[a-z_0-9]+)", name)
        k = m.group(1) if m else name
        if not k.startswith("k_replan") or k in ("k_replan_keys", "k_replan_export", "k_replan_import"):
            continue
        tot[k][counter] += float(val)
        launches[k][counter] += 1
res = {"command": command, "astar_expansions": exp, "astar_searches": b["config"]["astar"]["calls"],
       "expansions_per_s_in_kernel": b["roofline"].get("expansions_per_s_in_kernel"), "ms_per_step": b["ms_per_step"],
       "note": "k_replan and k_replan_quad share the expansions of a run with TS_QUAD=1 (the quads hand some vehicles to k_replan): "
               "the per-expansion figures of such a run are for the two kernels together"}
both = collections.defaultdict(float)
for k in tot:
    for c_, v in tot[k].items():
        both[c_] += v
for k in list(tot) + (["both"] if len(tot) > 1 else []):
    src = both if k == "both" else tot[k]
    res[k] = {"counters": dict(src), "launches": (dict(launches[k]) if k != "both" else None),
              "per_expansion": {c_: v / max(exp, 1) for c_, v in src.items()}}
json.dump(res, open(out_path, "w"), indent=1)
for k in res:
    if isinstance(res[k], dict) and "per_expansion" in res[k]:
        print(k, {c_: round(v, 2) for c_, v in sorted(res[k]["per_expansion"].items())})
