#!/usr/bin/env python3
"""Condense two rocprofv3 --pmc counter_collection.csv files (FETCH_SIZE run, WRITE_SIZE run) into
profiles/r01_pmc_summary.json: per kernel, launches and average / max KB per launch."""
import collections, csv, json, re, sys

def agg(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(r"\b(k_[a-z_0-9]+)", r["Kernel_Name"])
        d[m.group(1) if m else r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return d

f, w = agg(sys.argv[1], "FETCH_SIZE"), agg(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(f) | set(w)):
    fv, wv = f.get(k, []), w.get(k, [])
    out[k] = dict(launches=len(fv), fetch_kb_avg=sum(fv) / max(len(fv), 1), fetch_kb_max=max(fv) if fv else 0.0,
                  write_kb_avg=sum(wv) / max(len(wv), 1), write_kb_max=max(wv) if wv else 0.0)
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k.startswith("k_move") or k.startswith("k_decide")}, indent=1))
