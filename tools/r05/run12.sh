#!/bin/bash
# round 5: buffer-descriptor halo DMA in the symmetric (2-D) forward kernel, every single-source launch: tests + configs[3] A/B against the previous library
mkdir -p gpurun_out/r05
timeout 1500 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py tests/test_gpu_fullsize.py tests/test_gpu_fullsize_parity.py -x -q > gpurun_out/r05/t_sym.log 2>&1; echo "sym tests rc=$?" | tee -a gpurun_out/r05/summary.txt
tail -n 4 gpurun_out/r05/t_sym.log
for r in 1 2 3; do for cfg in FMRI_FH_MAXCH=0 FMRI_FH_MAXCH=2 ""; do env $cfg python3 bench.py --config cfg3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$cfg] rep $r  %.0f slices/s %.3f ms' % (d['value'], d['ms_per_step']))"; done; done | tee gpurun_out/r05/ab_cfg3_buf.log
