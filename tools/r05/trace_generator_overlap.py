#!/usr/bin/env python3
"""kernel_trace.csv of tools/bench_fit.py's device-generator leg -> what the generator's launches cost the training step: time per step, the
generator's own kernel time per step, and for every step kernel its mean duration when a generator kernel ran during it vs when none did."""
import csv
import sys
from collections import defaultdict

GEN = ("k_affine_sample", "k_elastic", "k_minmax_ws", "k_rescale_ws", "k_shot_", "k_noise_rng", "k_coarse_dropout")
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60])
            for r in rows)
adam = [k for k in ks if "k_adam" in k[2]]
n = len(adam) - 1
print("steps %d, %.3f ms per step" % (n, (adam[-1][1] - adam[0][1]) / n / 1e6))
lo, hi = adam[0][1], adam[-1][1]
seg = [k for k in ks if lo <= k[0] < hi]
gen = [k for k in seg if any(g in k[2] for g in GEN)]
print("generator kernels: %.3f ms per step in %.1f launches" % (sum(e - s for s, e, _ in gen) / n / 1e6, len(gen) / n))
step = [k for k in seg if not any(g in k[2] for g in GEN)]
print("step kernels: %.3f ms per step (sum of durations, two streams)" % (sum(e - s for s, e, _ in step) / n / 1e6))
# overlap
gi = 0
acc = defaultdict(lambda: [0, 0.0, 0, 0.0])
for s, e, name in step:
    while gi < len(gen) and gen[gi][1] <= s:
        gi += 1
    j, ov = gi, 0
    while j < len(gen) and gen[j][0] < e:
        ov += min(e, gen[j][1]) - max(s, gen[j][0])
        j += 1
    a = acc[name]
    if ov > 0:
        a[0] += 1; a[1] += e - s
    else:
        a[2] += 1; a[3] += e - s
print("%-60s %8s %10s %8s %10s %8s" % ("kernel", "n_with", "us_with", "n_clear", "us_clear", "extra_us/step"))
tot = 0.0
for name, (n1, t1, n0, t0) in sorted(acc.items(), key=lambda kv: -(kv[1][1] + kv[1][3])):
    if n1 and n0:
        extra = (t1 / n1 - t0 / n0) * n1 / n / 1e3
        tot += extra
        print("%-60s %8d %10.1f %8d %10.1f %8.1f" % (name, n1, t1 / n1 / 1e3, n0, t0 / n0 / 1e3, extra))
print("sum of (duration with - duration clear) x launches with: %.1f us per step" % tot)
