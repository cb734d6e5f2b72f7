#!/bin/bash
# Runs ON the GPU box (via gpurun) from the repo root: rocprofv3 kernel-trace stats of the default bench plus the two PMC passes
# (FETCH_SIZE, WRITE_SIZE - separate runs, counters only, as the MI355X guide prescribes), then tools/pmc_aggregate.py turns the
# counter CSVs into per-kernel HBM bytes per step.  Results land under gpurun_out/prof_<tag>/; copy what is to be judged to profiles/.
# Every file it produces is stamped with the commit it was collected on (the box has no .git: the caller passes `git rev-parse HEAD`,
# expanded in the container) and with bench.py's kernel_source_hash(), which bench.py checks at run time before it reports
# roofline.traffic from the profile.
#   usage (from the container): gpurun -- "bash tools/collect_profiles.sh r02 $(git rev-parse --short HEAD)"
set -u
TAG=${1:-r02}
HEAD=${2:-unknown}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
# default command: the weight-gradient kernels run on a second stream, concurrently with the input-gradient chain (durations overlap)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-exclusive-pass > "$OUT/stats.log" 2>&1
# one stream: exclusive kernel durations (what bench.py reports as roofline_exclusive)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_serial" -o stats -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --serialize-streams --per-layer "$OUT/per_layer.json" > "$OUT/stats_serial.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o fetch -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --serialize-streams > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o write -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --serialize-streams > "$OUT/pmc_write.log" 2>&1
cd "$ROOT"
SRC_HASH=$(python3 -c "import bench; print(bench.kernel_source_hash())")
python3 tools/pmc_aggregate.py "$OUT" 2 "$HEAD" "$SRC_HASH" > "$OUT/pmc_traffic_per_step.json"
STAMP="# git_head=$HEAD kernel_source_hash=$SRC_HASH tag=$TAG"
for pair in stats:kernel_stats.csv stats_serial:kernel_stats_serial.csv; do
    src=$(find "$OUT/${pair%%:*}" -name "*kernel_stats.csv" | head -1)
    [ -n "$src" ] && { echo "$STAMP"; cat "$src"; } > "$OUT/${pair##*:}"
done
python3 - "$OUT/per_layer.json" "$HEAD" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    d["git_head"] = sys.argv[2]
    json.dump(d, open(sys.argv[1], "w"), indent=1)
except Exception as e:
    print("per_layer.json not stamped:", e)
PY
ls -la "$OUT"
