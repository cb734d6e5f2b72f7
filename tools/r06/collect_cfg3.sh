#!/bin/bash
# configs[3] (2-D U-Net, batch 64 x 256x256x5) rocprof evidence: kernel stats of the timed region, the MFMA counter
# pass, and the exclusive per-op table.  Runs ON the GPU box from the repo root:  bash tools/r06/collect_cfg3.sh <tag> <git head>
set -u
TAG=${1:-r06}
HEAD=${2:-unknown}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_${TAG}_cfg3
mkdir -p "$OUT"
export TMPDIR=/tmp
SRC_HASH=$(python3 -c "import bench; print(bench.kernel_source_hash())")
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 "$ROOT/bench.py" --config cfg3 --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/stats.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_mfma" -o mfma -- python3 "$ROOT/tools/step_2d.py" 2 > "$OUT/pmc_mfma.log" 2>&1
cd "$ROOT"
python3 tools/pmc_mfma.py "$OUT" 2 "$HEAD" "$SRC_HASH" > "$OUT/cfg3_pmc_mfma.json"
python3 tools/per_layer_2d.py "$OUT/cfg3_per_layer.json" "$HEAD" > "$OUT/cfg3_per_layer.txt" 2>&1
src=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1)
[ -n "$src" ] && { echo "# git_head=$HEAD kernel_source_hash=$SRC_HASH tag=$TAG command: bench.py --config cfg3 --steps 10 --warmup 3 (two streams: durations overlap)"; cat "$src"; } > "$OUT/cfg3_kernel_stats.csv"
tail -3 "$OUT/stats.log"; tail -25 "$OUT/cfg3_per_layer.txt"
