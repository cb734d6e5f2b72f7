#!/bin/bash
mkdir -p gpurun_out/r05
for r in 1 2 3; do for cfg in FMRI_FWD_PRIO=0 FMRI_FWD_PRIO=2; do env $cfg python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --val-dice-steps 0 --no-secondary --no-exclusive-pass 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$cfg] rep $r  %.1f patches/s %.3f ms clock %.3f' % (d['value'], d['ms_per_step'], d['clock_ghz']))"; done; done | tee gpurun_out/r05/ab_xcd_numbering.log
