#!/bin/bash
# Kernel timeline of a few training steps (rocprofv3 --kernel-trace, per-dispatch start / end / stream): where the two streams idle.
#   gpurun -- 'bash tools/trace_step.sh'   ->  gpurun_out/trace_step/kernel_trace.csv (analysed by tools/trace_gaps.py)
set -u
ROOT=$PWD
OUT=$ROOT/gpurun_out/trace_step
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/raw" -o t -- python3 "$ROOT/bench.py" --steps 6 --warmup 2 --prewarm-seconds 0.5 --no-cpu-baseline --val-dice-steps 0 --no-secondary --no-launch-timing > "$OUT/run.log" 2>&1
cd "$ROOT"
f=$(find "$OUT/raw" -name "*kernel_trace.csv" | head -1)
python3 tools/trace_gaps.py "$f" > "$OUT/summary.txt"
# keep only the last ~3 steps of the trace (the merge limit of gpurun_out is 64 MiB)
python3 - "$f" "$OUT/kernel_trace_tail.csv" <<'PY'
import sys
rows = open(sys.argv[1]).read().splitlines()
open(sys.argv[2], "w").write("\n".join([rows[0]] + rows[-700:]) + "\n")
PY
rm -rf "$OUT/raw"
tail -40 "$OUT/summary.txt"
