#!/usr/bin/env python3
"""Per-kernel HBM traffic per training step from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, one counter per pass).

usage: pmc_aggregate.py <dir with pmc_fetch/ and pmc_write/> <steps profiled (warmup + timed)> [git head] [kernel source hash]
The two optional stamps land under "_meta"; bench.py reports roofline.traffic from the file only while the hash matches its tree.
Both counters are in KiB-like units of 1 KB per the guide's table; on gfx950 FETCH_SIZE under-reports wide streaming reads by 2x
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section), so hbm_mb_corrected = (2 * fetch_kb + write_kb) * 1024 / 1e6."""
import csv
import glob
import json
import os
import re
import sys


def load(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = re.sub(r"^void ", "", row["Kernel_Name"])
                name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0][:48]
                e = out.setdefault(name, [0, 0.0])
                e[0] += 1
                e[1] += float(row["Counter_Value"])
    return out


def main():
    root, steps = sys.argv[1], int(sys.argv[2])
    fe, wr = load(os.path.join(root, "pmc_fetch"), "FETCH_SIZE"), load(os.path.join(root, "pmc_write"), "WRITE_SIZE")
    res = {"_meta": {"git_head": sys.argv[3] if len(sys.argv) > 3 else None,
                     "kernel_source_hash": sys.argv[4] if len(sys.argv) > 4 else None, "steps_profiled": steps}}
    for k in fe:
        calls = fe[k][0] / steps
        f, w = fe[k][1] / steps, wr.get(k, [0, 0.0])[1] / steps
        res[k] = {"calls": calls, "fetch_kb": f, "write_kb": w, "hbm_mb_corrected": (2 * f + w) * 1024 / 1e6}
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
