#!/bin/bash
# round 6, planar pool / logits tails: the tests that run them, then configs[3] with and without (FMRI_TAIL_FUSE_2D), interleaved on one box
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py -x -q -k "planar or 2d or unet2d or tail" 2>&1 | tail -6 > gpurun_out/ptail_tests.log
for rep in 1 2 3; do
  for v in 1 0; do
    FMRI_TAIL_FUSE_2D=$v python tools/bench_2d.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('FMRI_TAIL_FUSE_2D=$v rep$rep  %.1f slices/s  %.3f ms  mfma_frac %.4f' % (d['slices_per_s'], d['ms_per_step'], d['mfma_frac']))"
  done
done > gpurun_out/ptail_ab.log 2>&1
cat gpurun_out/ptail_tests.log gpurun_out/ptail_ab.log
