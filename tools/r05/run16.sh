#!/bin/bash
# round 5: forward pass as two half-batch chains on two streams (FMRI_FWD_SPLIT=1): engine tests, then A/B
mkdir -p gpurun_out/r05
FMRI_FWD_SPLIT=1 timeout 1500 python3 -m pytest tests/test_gpu_engine.py tests/test_gpu_model.py tests/test_gpu_fullsize_parity.py -x -q > gpurun_out/r05/t_split.log 2>&1; echo "split tests rc=$?" | tee -a gpurun_out/r05/summary.txt
tail -n 4 gpurun_out/r05/t_split.log
for r in 1 2 3; do for cfg in FMRI_FWD_SPLIT=0 FMRI_FWD_SPLIT=1; do env $cfg python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --val-dice-steps 0 --no-secondary --no-launch-timing 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$cfg] rep $r  %.1f patches/s %.3f ms clock %.3f dice %.4f' % (d['value'], d['ms_per_step'], d['clock_ghz'], d['train_dice_last_step']))"; done; done | tee gpurun_out/r05/ab_fwd_split.log
