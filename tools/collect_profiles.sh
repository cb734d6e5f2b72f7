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
# PASSES="stats traffic mfma" (default: all) selects which collections run
PASSES=${PASSES:-stats traffic mfma}
want() { [[ " $PASSES " == *" $1 "* ]]; }
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1
cd /tmp
# default command: the weight-gradient kernels run on a second stream, concurrently with the input-gradient chain (durations overlap)
want stats && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --val-dice-steps 0 --no-secondary --no-exclusive-pass > "$OUT/stats.log" 2>&1
# one stream: exclusive kernel durations (what bench.py reports as roofline_exclusive)
want stats && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_serial" -o stats -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --val-dice-steps 0 --no-secondary --serialize-streams --per-layer "$OUT/per_layer.json" > "$OUT/stats_serial.log" 2>&1
want traffic && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o fetch -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --prewarm-seconds 0 --no-cpu-baseline --val-dice-steps 0 --no-secondary --serialize-streams > "$OUT/pmc_fetch.log" 2>&1
want traffic && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o write -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --prewarm-seconds 0 --no-cpu-baseline --val-dice-steps 0 --no-secondary --serialize-streams > "$OUT/pmc_write.log" 2>&1
# MFMA utilisation from the counters (north_star / SURVEY 8d "SQ_VALU_MFMA_BUSY_CYCLES"): SQ + GRBM counters only, their own pass, the
# program directly behind `--`.  SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles in which a SIMD's matrix pipe is busy (32 per
# v_mfma_f32_32x32x16_bf16), GRBM_GUI_ACTIVE is the sum over the 8 XCDs of their active cycles (MI355X_MICROARCH.md, DVFS give-back).
want mfma && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_mfma" -o mfma -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --prewarm-seconds 0 --no-cpu-baseline --val-dice-steps 0 --no-secondary --serialize-streams > "$OUT/pmc_mfma.log" 2>&1
want mfma && rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d "$OUT/pmc_sq" -o sq -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --prewarm-seconds 0 --no-cpu-baseline --val-dice-steps 0 --no-secondary --serialize-streams > "$OUT/pmc_sq.log" 2>&1
cd "$ROOT"
want mfma && python3 tools/pmc_mfma.py "$OUT" 2 "$HEAD" "$(python3 -c "import bench; print(bench.kernel_source_hash())")" > "$OUT/pmc_mfma.json"
SRC_HASH=$(python3 -c "import bench; print(bench.kernel_source_hash())")
want traffic && python3 tools/pmc_aggregate.py "$OUT" 2 "$HEAD" "$SRC_HASH" > "$OUT/pmc_traffic_per_step.json"
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
