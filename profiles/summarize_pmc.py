#!/usr/bin/env python3
"""Condense rocprofv3 output (rocpd .db files, or the older counter_collection.csv) into the summaries committed
under profiles/:

  summarize_pmc.py stats  <kernel-trace results.db>  <out.csv>          per-kernel calls / total / avg / min / max (ns)
  summarize_pmc.py pmc    <FETCH_SIZE run .db|.csv> <WRITE_SIZE run .db|.csv> <out.json>
                                                                          per kernel: launches, avg / max KB per launch

Collected with (separate runs, as MI355X_MICROARCH.md prescribes; `cd /tmp && export TMPDIR=/tmp` first):
  rocprofv3 --kernel-trace --stats -d <dir> -o bench -- python3 bench.py --steps 50 --warmup 5
  rocprofv3 --pmc FETCH_SIZE -d <dir> -o pmc -- python3 bench.py --steps 12 --warmup 3
  rocprofv3 --pmc WRITE_SIZE -d <dir> -o pmc -- python3 bench.py --steps 12 --warmup 3
"""
import collections, csv, json, re, sqlite3, sys


def short(name):
    m = re.search(r"\b(k_[a-z_0-9]+)", name)
    return m.group(1) if m else name


def agg(path, counter):
    d = collections.defaultdict(list)
    if path.endswith(".db"):
        c = sqlite3.connect(path)
        for name, val in c.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
            d[short(name)].append(float(val))
        return d
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return d


def main():
    if sys.argv[1] == "stats":
        c = sqlite3.connect(sys.argv[2])
        rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                         "group by name order by sum(duration) desc").fetchall()
        total = sum(r[2] for r in rows)
        with open(sys.argv[3], "w", newline="") as f:
            w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows:
                w.writerow([r[0], r[1], r[2], round(r[3], 3), round(100.0 * r[2] / total, 4), r[4], r[5]])
        for r in rows[:8]:
            print(f"{short(r[0]):24s} calls {r[1]:5d} avg {r[3] / 1e3:9.2f} us  total {r[2] / 1e6:8.2f} ms")
        # "k_name:N first_leg_calls": the bench line's HIP-event figure covers the timed region only - the last N of the
        # kernel's launches in the first (headline) leg of the traced process; the trace also holds its warm-up launches
        # "replan_regions:N": the replanning phase of a tick is one k_replan launch, or - on a replanning wave - k_replan_quad
        # with a k_replan launch running beside it: one region per tick = the union of the launches that overlap; the last N
        # regions are the bench's timed ticks (its HIP events bracket exactly these regions)
        for spec in [x for x in sys.argv[4:] if x.startswith("replan_regions:")]:
            n = int(spec.split(":")[1])
            rows2 = c.execute("select name, start, end from kernels where name like '%k_replan(%' or name like '%k_replan_quad(%' order by start").fetchall()
            regions = []
            for name, st_, en_ in rows2:
                if regions and st_ < regions[-1][1]:
                    regions[-1][1] = max(regions[-1][1], en_); regions[-1][2] += 1
                else:
                    regions.append([st_, en_, 1])
            last = [r[1] - r[0] for r in regions[-n:]]
            with open(sys.argv[3], "a", newline="") as f:
                csv.writer(f, quoting=csv.QUOTE_NONNUMERIC).writerow(
                    [f"replanning regions (k_replan_quad + k_replan beside it, or k_replan alone) [last {len(last)} = the timed ticks]", len(last), sum(last),
                     round(sum(last) / max(len(last), 1), 3), "", min(last), max(last)])
            print(f"replanning regions: last {len(last)} avg {sum(last) / max(len(last), 1) / 1e3:.2f} us (of {len(regions)} regions, {sum(1 for r in regions if r[2] > 1)} with two kernels)")
        for spec in [x for x in sys.argv[4:] if not x.startswith("replan_regions:")]:
            kname, n = spec.split(":")
            # (exactly this kernel: `k_replan(`, not k_replan_keys / k_replan_export)
            durs = [r[0] for r in c.execute("select duration from kernels where name like ? order by start", (f"%{kname}(%",)).fetchall()]
            last = durs[-int(n):]
            with open(sys.argv[3], "a", newline="") as f:
                csv.writer(f, quoting=csv.QUOTE_NONNUMERIC).writerow(
                    [f"{kname} [last {len(last)} launches = the timed region]", len(last), sum(last), round(sum(last) / max(len(last), 1), 3), "", min(last), max(last)])
            print(f"{kname}: last {len(last)} launches avg {sum(last) / max(len(last), 1) / 1e3:.2f} us")
        return
    f, w = agg(sys.argv[2], "FETCH_SIZE"), agg(sys.argv[3], "WRITE_SIZE")
    out = {}
    for k in sorted(set(f) | set(w)):
        fv, wv = f.get(k, []), w.get(k, [])
        out[k] = dict(launches=len(fv), fetch_kb_avg=sum(fv) / max(len(fv), 1), fetch_kb_max=max(fv) if fv else 0.0,
                      write_kb_avg=sum(wv) / max(len(wv), 1), write_kb_max=max(wv) if wv else 0.0)
    if len(sys.argv) > 5:      # bench.py reads `kernels` when `command_config` = [size, vehicles, policy, steps, warmup] matches its run
        size, vehicles, policy, steps, warmup = sys.argv[5].split(",")
        out = {"command": f"python3 bench.py --steps {steps} --warmup {warmup} --no-cpu-baseline (full-policy leg + the config2 / lights secondary legs, one process)",
               "command_config": [int(size), int(vehicles), policy, int(steps), int(warmup)],
               "note": "KB per launch as rocprofv3 reports them; bench.py doubles FETCH_SIZE (gfx950: 128-B requests tallied at 64 B)",
               "kernels": out}
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    out = out.get("kernels", out)
    # k_decide_replan = the replanning phase as bench.py names it: k_replan + k_replan_quad launches (per LAUNCH REGION the
    # bench divides by the ticks; here the per-launch averages of each kernel are kept, and their sum over the process)
    print(json.dumps({k: v for k, v in out.items() if k.startswith("k_move") or k.startswith("k_decide") or k.startswith("k_replan")}, indent=1))


if __name__ == "__main__":
    main()
