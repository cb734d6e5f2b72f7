#!/bin/bash
# Per-layer A/B on ONE box: bench.py --per-layer once per "VAR=value" argument (and once plain), then the exclusive ms of every conv x pass
# side by side.   usage: bash tools/ab_layers.sh FMRI_FWD_ASYNC=0
mkdir -p gpurun_out/ab
REPS=${REPS:-2}
for rep in $(seq 1 $REPS); do
  i=0
  for cfg in "" "$@"; do
    env $cfg python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --val-dice-steps 0 --no-secondary --per-layer gpurun_out/ab/pl_${i}_$rep.json > gpurun_out/ab/bench_${i}_$rep.json 2>/dev/null
    i=$((i+1))
  done
done
export REPS
python3 - "$@" <<'PY'
import json, sys
import os
cfgs = ["default"] + sys.argv[1:]
reps = int(os.environ.get("REPS", "2"))
tabs = [[json.load(open("gpurun_out/ab/pl_%d_%d.json" % (i, r))) for r in range(1, reps + 1)] for i in range(len(cfgs))]
lines = [[json.loads(open("gpurun_out/ab/bench_%d_%d.json" % (i, r)).read().strip().splitlines()[-1]) for r in range(1, reps + 1)] for i in range(len(cfgs))]
fmt = lambda vs: "%22s" % (" ".join("%.4g" % v for v in vs))
print("(each cell: one value per interleaved repetition)")
print("%-22s" % "" + "".join("%22s" % c[-22:] for c in cfgs))
print("%-22s" % "patches/s" + "".join(fmt([l["value"] for l in ls]) for ls in lines))
print("%-22s" % "ms/step" + "".join(fmt([l["ms_per_step"] for l in ls]) for ls in lines))
for k in ("conv_fwd_mfma", "conv_wgrad_mfma"):
    print("%-22s" % ("excl " + k) + "".join(fmt([l["roofline_exclusive"]["kernel_ms_per_step"].get(k, 0) for l in ls]) for ls in lines))
for r0 in tabs[0][0]["rows"]:
    key = (r0["layer"], r0["pass"])
    cells = []
    for ts in tabs:
        vals = []
        for t in ts:
            m = [r for r in t["rows"] if (r["layer"], r["pass"]) == key]
            vals.append(m[0]["ms"] if m else float("nan"))
        cells.append(fmt(vals))
    print("%-10s %4d->%-4d %-5s" % (r0["layer"], r0["cin"], r0["cout"], r0["pass"]) + "".join(cells))
PY
