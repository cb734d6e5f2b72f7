#!/bin/bash
# round 5, fourth GPU call: the buffer-descriptor LDS-DMA probe; the zero-page address hoisted out of the DMA issue paths (A/B against the round-4 library)
mkdir -p gpurun_out/r05
./tools/probe/probe_bufdma > gpurun_out/r05/probe_bufdma.log 2>&1; cat gpurun_out/r05/probe_bufdma.log
timeout 900 python3 -m pytest tests/test_gpu_ops.py -x -q -k "conv3d or upcat or dgrad or wgrad" > gpurun_out/r05/t_ops_zp.log 2>&1; echo "ops_zp rc=$?" | tee -a gpurun_out/r05/summary.txt
REPS=3 bash tools/ab_layers.sh FMRI_LIB=$PWD/fetal-mri-segmentation_amd/lib/libfmri_hip_r04.so "FMRI_MFMA16=1 FMRI_WGRAD_MFMA16=1" > gpurun_out/r05/ab_zero_page.log 2>&1
cat gpurun_out/r05/ab_zero_page.log
for i in 0 1 2; do for r in 1 2 3; do python3 -c "
import json
l=json.loads(open('gpurun_out/ab/bench_${i}_${r}.json').read().strip().splitlines()[-1])
print($i,$r,'%.1f patches/s  clock %.3f GHz' % (l['value'], l['clock_ghz']))"; done; done | tee gpurun_out/r05/ab_zero_page_clock.log
