set -u
ROOT=$PWD
OUT=$ROOT/gpurun_out/trace_fit
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
FMRI_BENCH_FIT_ONLY=device rocprofv3 --kernel-trace --output-format csv -d "$OUT/raw" -o t -- python3 "$ROOT/tools/bench_fit.py" > "$OUT/run.log" 2>&1
cd "$ROOT"
f=$(find "$OUT/raw" -name "*kernel_trace.csv" | head -1)
python3 tools/trace_gaps.py "$f" > "$OUT/summary.txt"
rm -rf "$OUT/raw"
head -60 "$OUT/summary.txt"; tail -3 "$OUT/run.log"
