#!/bin/bash
# round 5, eighth GPU call: the first conv's weight gradient on the main stream (A/B)
mkdir -p gpurun_out/r05
REPS=3 bash tools/ab_bench.sh FMRI_FIRST_WGRAD_MAIN=1 > gpurun_out/r05/ab_first_wgrad.log 2>&1
for cfg in "" FMRI_FIRST_WGRAD_MAIN=1; do for r in 1 2 3; do env $cfg python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --val-dice-steps 0 --no-secondary --no-launch-timing 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$cfg] %.1f patches/s %.3f ms clock %.3f' % (d['value'], d['ms_per_step'], d['clock_ghz']))"; done; done | tee -a gpurun_out/r05/ab_first_wgrad.log
cat gpurun_out/r05/ab_first_wgrad.log
