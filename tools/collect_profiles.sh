#!/bin/bash
# Runs ON the GPU box (via gpurun) from the repo root: rocprofv3 kernel-trace stats of the default bench plus the two PMC passes
# (FETCH_SIZE, WRITE_SIZE - separate runs, counters only, as the MI355X guide prescribes), then tools/pmc_aggregate.py turns the
# counter CSVs into per-kernel HBM bytes per step.  Results land under gpurun_out/prof_<tag>/; copy what is to be judged to profiles/.
#   usage: bash tools/collect_profiles.sh r01
set -u
TAG=${1:-r01}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
# default command: the weight-gradient kernels run on a second stream, concurrently with the input-gradient chain (durations overlap)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-exclusive-pass > "$OUT/stats.log" 2>&1
# one stream: exclusive kernel durations (what bench.py reports as roofline_exclusive)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_serial" -o stats -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --serialize-streams > "$OUT/stats_serial.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o fetch -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --serialize-streams > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o write -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --serialize-streams > "$OUT/pmc_write.log" 2>&1
cd "$ROOT"
python3 tools/pmc_aggregate.py "$OUT" 2 > "$OUT/pmc_traffic_per_step.json"
find "$OUT/stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
find "$OUT/stats_serial" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_serial.csv"
ls -la "$OUT"
