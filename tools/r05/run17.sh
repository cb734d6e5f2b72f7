set -u
ROOT=$PWD
OUT=$ROOT/gpurun_out/trace_gen
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
FMRI_BENCH_FIT_STEPS=20 FMRI_BENCH_FIT_ONLY=device rocprofv3 --kernel-trace --output-format csv -d "$OUT/raw" -o t -- python3 "$ROOT/tools/bench_fit.py" > "$OUT/run.log" 2>&1
cd "$ROOT"
f=$(find "$OUT/raw" -name "*kernel_trace.csv" | head -1)
python3 tools/r05/trace_generator_overlap.py "$f" > "$OUT/summary.txt"
FMRI_BENCH_FIT_STEPS=20 FMRI_BENCH_FIT_ONLY=resident rocprofv3 --kernel-trace --output-format csv -d "$OUT/raw2" -o t -- python3 "$ROOT/tools/bench_fit.py" > "$OUT/run2.log" 2>&1
f=$(find "$OUT/raw2" -name "*kernel_trace.csv" | head -1)
python3 tools/r05/trace_generator_overlap.py "$f" > "$OUT/summary_resident.txt"
rm -rf "$OUT/raw" "$OUT/raw2"
cat "$OUT/summary.txt"; tail -2 "$OUT/run.log"; head -4 "$OUT/summary_resident.txt"
