#!/usr/bin/env python3
"""Per-kernel HBM traffic per training step from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, one counter per pass).

usage: pmc_aggregate.py <dir with pmc_fetch/ and pmc_write/> <steps profiled (fallback when the trace holds no k_adam pair)> [git head] [kernel source hash]
The two optional stamps land under "_meta"; bench.py reports roofline.traffic from the file only while the hash matches its tree.
Both counters are in KiB-like units of 1 KB per the guide's table; on gfx950 FETCH_SIZE under-reports wide streaming reads by 2x
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section), so hbm_mb_corrected = (2 * fetch_kb + write_kb) * 1024 / 1e6."""
import csv
import glob
import json
import os
import re
import sys


def short_name(kernel_name):
    name = re.sub(r"^void ", "", kernel_name)
    return re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0][:48]


def load(d, counter):
    """-> ({kernel: [launches, counter sum]}, optimizer steps covered): only the dispatches BETWEEN the first and the last k_adam launch of
    the run (the first step's own kernels up to its k_adam excluded, every later step whole) - what lies in front of the first optimizer
    step is the benchmark building its pool of batches (k_correlate1d_sym, aten element-wise / reduce / cat kernels: 2.6 GB of the 33.2 GB
    round 4's summary quoted per step) and behind the last one nothing of the step."""
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") == counter:
                    rows.append((int(row["Dispatch_Id"]), short_name(row["Kernel_Name"]), float(row["Counter_Value"])))
    rows.sort()
    adam = [i for i, (_, name, _) in enumerate(rows) if name.startswith("k_adam")]
    steps = len(adam) - 1
    if steps >= 1:
        rows = rows[adam[0] + 1:adam[-1] + 1]
    else:
        steps = 0                                  # no optimizer step pair in the trace: everything, the caller's step count
    out = {}
    for _, name, val in rows:
        e = out.setdefault(name, [0, 0.0])
        e[0] += 1
        e[1] += val
    return out, steps


def main():
    root, steps_arg = sys.argv[1], int(sys.argv[2])
    (fe, sf), (wr, sw) = load(os.path.join(root, "pmc_fetch"), "FETCH_SIZE"), load(os.path.join(root, "pmc_write"), "WRITE_SIZE")
    steps_f, steps_w = sf or steps_arg, sw or steps_arg
    res = {"_meta": {"git_head": sys.argv[3] if len(sys.argv) > 3 else None,
                     "kernel_source_hash": sys.argv[4] if len(sys.argv) > 4 else None, "steps_profiled": steps_f,
                     "window": "dispatches between the first and the last k_adam of the run" if sf else "whole run (no k_adam pair found)"}}
    total = 0.0
    for k in fe:
        calls = fe[k][0] / steps_f
        f, w = fe[k][1] / steps_f, wr.get(k, [0, 0.0])[1] / steps_w
        res[k] = {"calls": calls, "fetch_kb": f, "write_kb": w, "hbm_mb_corrected": (2 * f + w) * 1024 / 1e6}
        total += res[k]["hbm_mb_corrected"]
    res["_meta"]["all_kernels_hbm_mb_corrected"] = total
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
