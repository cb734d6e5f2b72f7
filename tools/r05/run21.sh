#!/bin/bash
# round 5: FETCH_SIZE calibrated on the forward kernel's own halo access pattern (tools/probe/probe_fetch_calib.hip)
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/fc_pmc -o f -- $ROOT/tools/probe/probe_fetch_calib > /tmp/fc1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fc_st -o s -- $ROOT/tools/probe/probe_fetch_calib > /tmp/fc2.log 2>&1
tail -1 /tmp/fc2.log
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/fc_pmc/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
acc = collections.defaultdict(list)
for r in rows:
    if r.get("Counter_Name") == "FETCH_SIZE":
        acc[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
st = glob.glob("/tmp/fc_st/**/*kernel_stats.csv", recursive=True)[0]
dur = {r["Name"][:40]: float(r["AverageNs"]) for r in csv.DictReader(open(st))}
req = {"0": 2048, "1": 1024, "2": 2048, "3": 512}
print("%-34s %10s %14s %10s %12s" % ("kernel", "req MiB", "FETCH_SIZE KiB", "FETCH/req", "us per launch"))
for k, v in sorted(acc.items()):
    mode = k.split("<")[1][0] if "<" in k else "?"
    m = sum(v) / len(v)
    rq = req.get(mode, 0)
    print("%-34s %10d %14.0f %10.3f %12.1f   (%.2f TB/s requested)" % (k, rq, m, m / 1024.0 / rq if rq else 0, dur.get(k, 0) / 1e3, rq * 1.048576e6 / max(dur.get(k, 1), 1) / 1e3))
PY
