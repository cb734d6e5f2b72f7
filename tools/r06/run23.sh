#!/bin/bash
# round 6: planar tails (DPP logits reduction) + dice kernel: per-layer costs, tests, configs[3] A/B, configs[1] step A/B against the library before
mkdir -p gpurun_out
python tools/r06/bench_ptail.py 2>&1 | grep -v amdgpu.ids > gpurun_out/ptail_layers2.log
python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py tests/test_gpu_losses.py -x -q 2>&1 | tail -4 > gpurun_out/ptail_tests2.log
for rep in 1 2 3; do
  for v in 1 0; do
    FMRI_TAIL_FUSE_2D=$v python tools/bench_2d.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('FMRI_TAIL_FUSE_2D=$v rep$rep  %.1f slices/s  %.3f ms  mfma_frac %.4f' % (d['slices_per_s'], d['ms_per_step'], d['mfma_frac']))"
  done
done > gpurun_out/ptail_ab2.log 2>&1
bash tools/ab_bench.sh FMRI_LIB=$PWD/build/ab/libfmri_hip_base.so > gpurun_out/dice_step_ab.log 2>&1
cat gpurun_out/ptail_layers2.log gpurun_out/ptail_tests2.log gpurun_out/ptail_ab2.log; cut -c1-120 gpurun_out/dice_step_ab.log
