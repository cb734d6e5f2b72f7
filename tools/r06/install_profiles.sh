#!/bin/bash
# container side: copy what tools/r06/run6.sh + tools/gputest_stamp.sh r06 left under gpurun_out/ into profiles/ and regenerate the summary + DESIGN block
set -e
cd "$(dirname "$0")/../.."
cp gpurun_out/r06_gputest_final.log profiles/
P=gpurun_out/prof_r06
cp $P/kernel_stats.csv profiles/r06_kernel_stats.csv
cp $P/kernel_stats_serial.csv profiles/r06_kernel_stats_one_stream.csv
cp $P/per_layer.json profiles/r06_per_layer.json
cp $P/pmc_traffic_per_step.json profiles/r06_pmc_traffic_per_step.json
cp $P/pmc_mfma.json profiles/r06_pmc_mfma.json
cp $P/pmc_mfma_w1.json profiles/r06_pmc_mfma_w1.json
cp $P/f32_kernel_stats.csv profiles/r06_f32_kernel_stats.csv
cp gpurun_out/bench_all.jsonl profiles/r06_bench_all.jsonl
cp gpurun_out/trace_step/summary.txt profiles/r06_trace_step_summary.txt
cp gpurun_out/r06_bench_default.json profiles/r06_bench_default.json
C=gpurun_out/prof_r06_cfg3
cp $C/cfg3_kernel_stats.csv profiles/r06_cfg3_kernel_stats.csv
cp $C/cfg3_pmc_mfma.json profiles/r06_cfg3_pmc_mfma.json
cp $C/cfg3_per_layer.json profiles/r06_cfg3_per_layer.json
python3 tools/summarize_profiles.py r06 | tail -1
python3 - <<'PY'
import sys
sys.path.insert(0, 'tools')
import summarize_profiles as SP
fresh = SP.summary('r06')
s = open('DESIGN.md').read()
b = s.index("<!-- generated: profiles/r06_summary.md -->"); e = s.index("<!-- end generated -->")
open('DESIGN.md', 'w').write(s[:b] + "<!-- generated: profiles/r06_summary.md -->\n" + SP.demote(fresh) + s[e:])
PY
