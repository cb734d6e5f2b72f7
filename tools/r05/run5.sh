#!/bin/bash
# round 5, fifth GPU call: halo pieces through a buffer descriptor (FMRI_FH_MAXCH=99: every single-source 3-D launch), with both MFMA shapes
mkdir -p gpurun_out/r05
./tools/probe/probe_bufdma > gpurun_out/r05/probe_bufdma.log 2>&1; cat gpurun_out/r05/probe_bufdma.log
for cfg in "FMRI_FH_MAXCH=99" "FMRI_FH_MAXCH=99 FMRI_MFMA16=1 FMRI_WGRAD_MFMA16=1" ""; do
  env $cfg timeout 900 python3 -m pytest tests/test_gpu_ops.py -x -q -k "conv3d or upcat or dgrad or wgrad" > gpurun_out/r05/t_ops_buf.log 2>&1; echo "ops [$cfg] rc=$?" | tee -a gpurun_out/r05/summary.txt
  env $cfg timeout 900 python3 -m pytest tests/test_gpu_fullsize_parity.py -x -q -k "dyadic or kd_sharing or n1_full" > gpurun_out/r05/t_full_buf.log 2>&1; echo "full [$cfg] rc=$?" | tee -a gpurun_out/r05/summary.txt
  tail -n 3 gpurun_out/r05/t_ops_buf.log gpurun_out/r05/t_full_buf.log
done
REPS=3 bash tools/ab_layers.sh FMRI_FH_MAXCH=99 "FMRI_FH_MAXCH=99 FMRI_MFMA16=1 FMRI_WGRAD_MFMA16=1" "FMRI_MFMA16=1 FMRI_WGRAD_MFMA16=1" > gpurun_out/r05/ab_buf.log 2>&1
cat gpurun_out/r05/ab_buf.log
for i in 0 1 2 3; do for r in 1 2 3; do python3 -c "
import json
l=json.loads(open('gpurun_out/ab/bench_${i}_${r}.json').read().strip().splitlines()[-1])
print($i,$r,'%.1f patches/s  clock %.3f GHz' % (l['value'], l['clock_ghz']))"; done; done | tee gpurun_out/r05/ab_buf_clock.log
