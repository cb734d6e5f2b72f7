#!/bin/bash
# round 5, final: profile collection on the committed kernels (7a695bd) + the default bench line + every measured configuration
bash tools/collect_profiles.sh r05 7a695bd > gpurun_out/collect_r05.log 2>&1
tail -5 gpurun_out/collect_r05.log
python3 bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
tail -c 1500 gpurun_out/r05_bench_default.json
bash tools/bench_all.sh > /dev/null 2> gpurun_out/bench_all.err
cp gpurun_out/bench_all.jsonl gpurun_out/r05_bench_all.jsonl
