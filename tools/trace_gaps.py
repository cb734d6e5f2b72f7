#!/usr/bin/env python3
"""Reads a rocprofv3 kernel_trace.csv and prints, for the last full training steps: per stream (queue) busy time, the idle gaps between
consecutive kernels of the chip as a whole (no kernel of any stream running), and the kernels in time order with their overlap."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ks = []
for r in rows:
    ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70], r.get("Queue_Id", "?"), r.get("Stream_Id", r.get("Queue_Id", "?"))))
ks.sort()
# a step starts at each k_adam launch's end
adam = [i for i, k in enumerate(ks) if "k_adam" in k[2]]
if len(adam) < 3:
    print("fewer than 3 steps in the trace")
    sys.exit(0)
lo, hi = adam[-3], adam[-1]
seg = ks[lo + 1:hi + 1]
t0, t1 = seg[0][0], seg[-1][1]
nsteps = 2
print("steps analysed: %d, span %.3f ms per step" % (nsteps, (t1 - t0) / nsteps / 1e6))
busy = defaultdict(float)
for s, e, n, q, st in seg:
    busy[q] += (e - s)
for q, b in sorted(busy.items()):
    print("queue %s: busy %.3f ms per step" % (q, b / nsteps / 1e6))
# union of busy intervals -> idle time of the whole chip
ev = sorted((s, e) for s, e, *_ in seg)
cur_s, cur_e = ev[0]
union = 0
gaps = []
for s, e in ev[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        gaps.append((s - cur_e, cur_e))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print("some kernel running: %.3f ms per step; nothing running: %.3f ms per step in %d gaps (largest %.1f us)" % (
    union / nsteps / 1e6, (t1 - t0 - union) / nsteps / 1e6, len(gaps), max(g[0] for g in gaps) / 1e3 if gaps else 0))
fam = defaultdict(float)
for s, e, n, q, st in seg:
    fam[n[:40]] += e - s
print("per kernel name (sum of durations per step, ms):")
for n, t in sorted(fam.items(), key=lambda kv: -kv[1])[:25]:
    print("  %-42s %.3f" % (n, t / nsteps / 1e6))
print("timeline of the last step (start offset us, duration us, queue, kernel):")
last = ks[adam[-2] + 1:adam[-1] + 1]
b0 = last[0][0]
for s, e, n, q, st in last:
    print("  %9.1f %8.1f  q%-3s %s" % ((s - b0) / 1e3, (e - s) / 1e3, q, n))
