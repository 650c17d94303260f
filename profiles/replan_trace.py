#!/usr/bin/env python3
"""Per-search timeline of k_replan's work queue, from a -DTS_TRACE_REPLAN build (profiles/run_replan_trace.sh).

The engine of such a build writes gpurun_out/rtrace_tick<N>.bin for every k_replan-only replanning pass of 1000 entries
and more: one int4 per queue entry = (start, end: low words of the 100 MHz wall clock; expansions of the vehicle's
searches; predicted cost bits | searcher slot << 8).  This script turns them into one JSON summary: for every traced
tick the span of the pass, when the queue ran empty, the expansion rate per tenth of the span, and the searches that
finished last (start, duration, expansions, microseconds per expansion, predicted vs actual cost bits, queue position).

    python3 profiles/replan_trace.py gpurun_out/rtrace_tick*.bin > profiles/r03_replan_trace_4096.json
"""
import json
import sys

import numpy as np


def summarize(path):
    a = np.fromfile(path, dtype=np.int32).reshape(-1, 4)
    t0 = a[:, 0].astype(np.uint32).astype(np.int64)
    t1 = a[:, 1].astype(np.uint32).astype(np.int64)
    ok = (t0 != 0) | (t1 != 0)
    a, t0, t1 = a[ok], t0[ok], t1[ok]
    pos = np.nonzero(ok)[0]
    base = t0.min()
    s, e = (t0 - base) / 100.0, (t1 - base) / 100.0          # microseconds
    exp = a[:, 2].astype(np.int64)
    bits = a[:, 3] & 255
    span = float(e.max())
    deciles = np.zeros(10)
    for k in np.nonzero(exp > 0)[0]:
        lo, hi = s[k], max(e[k], s[k] + 1e-3)
        for b in range(int(lo / span * 10), min(int(hi / span * 10), 9) + 1):
            l, h = max(lo, b * span / 10), min(hi, (b + 1) * span / 10)
            if h > l:
                deciles[b] += exp[k] * (h - l) / (hi - lo)
    last = np.argsort(-e)[:6]
    top = np.argsort(-exp)[:256]
    return {
        "entries": int(len(a)), "span_ms": span / 1e3, "queue_empty_at_ms": float(s.max()) / 1e3,
        "expansions": int(exp.sum()),
        "gexp_per_s_by_tenth_of_span": [round(float(x) / (span / 10 * 1e-6) / 1e9, 2) for x in deciles],
        "searches_over": {str(n): int((exp > n).sum()) for n in (16384, 32768, 65536, 100000, 150000)},
        "of_the_256_most_expensive_in_the_first_n_queue_positions": {str(n): int((pos[top] < n).sum()) for n in (256, 1024, 2048)},
        "last_to_finish": [{"start_ms": round(float(s[k]) / 1e3, 1), "duration_ms": round(float(e[k] - s[k]) / 1e3, 1),
                            "expansions": int(exp[k]), "us_per_expansion": round(float(e[k] - s[k]) / max(int(exp[k]), 1), 2),
                            "predicted_bits": int(bits[k]), "actual_bits": int(exp[k]).bit_length(), "queue_position": int(pos[k])}
                           for k in last],
    }


def main():
    out = {}
    for p in sorted(sys.argv[1:], key=lambda q: int(q.split("tick")[1].split(".")[0])):
        out["tick_" + p.split("tick")[1].split(".")[0]] = summarize(p)
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
