#!/usr/bin/env python3
"""Steady-state window of a rocprofv3 kernel trace (bench.py under --kernel-trace): how much of the time the GPU runs 0 / 1 / 2 / ... kernels at
once, and where the idle gaps are.  usage: tools/trace_concurrency.py <..._kernel_trace.csv>  -> JSON on stdout"""
import collections, csv, json, sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
anchor = "stem_down_kernel"
starts = [e[0] for e in ev if anchor in e[2]]
runs, cur = [], [starts[0]]
for a, b in zip(starts, starts[1:]):              # the timed loop: the longest run of stem launches less than 6 ms apart
    if b - a < 6e6:
        cur.append(b)
    else:
        runs.append(cur)
        cur = [b]
runs.append(cur)
run = max(runs, key=len)
t0, t1 = run[2], run[-3]
win = [e for e in ev if t0 <= e[0] < t1]
busy, cs, ce, gaps = 0, win[0][0], win[0][1], []
for s, e, n in win[1:]:
    if s > ce:
        busy += ce - cs
        gaps.append((s - ce, n.split("(")[0][-40:]))
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
pts = sorted([(s, 1) for s, _, _ in win] + [(e, -1) for _, e, _ in win])
c, last, hist = 0, pts[0][0], collections.Counter()
for t, d in pts:
    hist[c] += t - last
    last, c = t, c + d
tot = sum(hist.values())
print(json.dumps({"window_ms": round((t1 - t0) / 1e6, 3), "stem_launches_in_window": len(run) - 5, "busy_fraction": round(busy / (t1 - t0), 4),
                  "idle_gaps": len(gaps), "idle_us_total": round(sum(g for g, _ in gaps) / 1e3, 1),
                  "largest_gaps_us_before_kernel": [(round(g / 1e3, 1), n) for g, n in sorted(gaps, reverse=True)[:6]],
                  "time_share_by_kernels_in_flight": {str(k): round(v / tot, 4) for k, v in sorted(hist.items())}}, indent=1))
