#!/usr/bin/env python3
"""Concurrency of the frame's kernels from a rocprofv3 --kernel-trace CSV: tools/timeline.py <kernel_trace.csv> [skip_fraction]
Prints per-kernel mean duration and, over the steady-state part of the run, how much of the wall time had 0, 1, 2, ... kernels
(and traversal kernels) executing at once."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = []
for r in rows:
    n = r["Kernel_Name"].replace("rt::", "")
    if not any(k in n for k in ("k_trace", "k_shade", "k_raygen", "k_tail", "k_resolve")):
        continue
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0][:40]))
ev.sort()
t_lo = ev[0][0] + (ev[-1][1] - ev[0][0]) * skip
ev = [e for e in ev if e[0] >= t_lo]
dur = collections.defaultdict(list)
for a, b, n in ev:
    dur[n].append(b - a)
print("steady-state window: %.3f ms, %d dispatches" % ((ev[-1][1] - ev[0][0]) / 1e6, len(ev)))
for n, v in sorted(dur.items()):
    print("  %-42s n=%4d mean %.1f us  max %.1f us" % (n, len(v), sum(v) / len(v) / 1e3, max(v) / 1e3))
pts = []
for a, b, n in ev:
    tr = 1 if "k_trace" in n else 0
    pts.append((a, 1, tr)); pts.append((b, -1, -tr))
pts.sort()
hist, hist_tr = collections.Counter(), collections.Counter()
cur = cur_tr = 0
last = pts[0][0]
for t, d, dt in pts:
    hist[cur] += t - last; hist_tr[cur_tr] += t - last
    cur += d; cur_tr += dt; last = t
tot = sum(hist.values())
print("kernels executing at once:   " + "  ".join("%d: %.1f%%" % (k, 100.0 * v / tot) for k, v in sorted(hist.items())))
print("traversal kernels at once:   " + "  ".join("%d: %.1f%%" % (k, 100.0 * v / tot) for k, v in sorted(hist_tr.items())))
