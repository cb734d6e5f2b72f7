#!/bin/bash
# round 6, final evidence collection on the kernels of this tree (tools/collect_profiles.sh + cfg3 + fp32 mode + W1 counters + variants + timeline)
HEAD=$1
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/collect_profiles.sh r06 $HEAD > gpurun_out/r06_collect.log 2>&1; tail -3 gpurun_out/r06_collect.log
bash tools/r06/collect_cfg3.sh r06 $HEAD > gpurun_out/r06_collect_cfg3.log 2>&1; tail -2 gpurun_out/r06_collect_cfg3.log
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_r06
SRC_HASH=$(python3 -c "import bench; print(bench.kernel_source_hash())")
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/f32_stats" -o stats -- python3 "$ROOT/tools/step_f32.py" 3 > "$OUT/f32_stats.log" 2>&1
FMRI_WGRAD_KD_BLK=64 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/w1/pmc_mfma" -o mfma -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --prewarm-seconds 0 --no-cpu-baseline --val-dice-steps 0 --no-secondary --serialize-streams > "$OUT/w1_pmc_mfma.log" 2>&1
cd "$ROOT"
python3 tools/pmc_mfma.py "$OUT/w1" 2 "$HEAD" "$SRC_HASH" > "$OUT/pmc_mfma_w1.json"
src=$(find "$OUT/f32_stats" -name "*kernel_stats.csv" | head -1)
[ -n "$src" ] && { echo "# git_head=$HEAD kernel_source_hash=$SRC_HASH tag=r06 command: tools/step_f32.py 3 (fp32 parity mode, configs[1], one untimed + 3 steps)"; cat "$src"; } > "$OUT/f32_kernel_stats.csv"
tail -1 "$OUT/f32_stats.log"
bash tools/trace_step.sh > gpurun_out/r06_trace_step.log 2>&1; tail -12 gpurun_out/r06_trace_step.log
bash tools/bench_all.sh 2> gpurun_out/r06_bench_all.err | cut -c1-300
python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
tail -c 600 gpurun_out/r06_bench_default.json
