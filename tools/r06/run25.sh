#!/bin/bash
# round 6, batched weight repack (fmri_pack_weights_batched): tests, then the step with FMRI_PACK_BATCHED=1 (default) / 0, interleaved, configs[1] and configs[3]
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -x -q -k "pack or batched" 2>&1 | tail -3 > gpurun_out/batched_tests.log
bash tools/ab_bench.sh FMRI_PACK_BATCHED=0 > gpurun_out/batched_ab.log 2>&1
for rep in 1 2; do
  for v in 1 0; do
    FMRI_PACK_BATCHED=$v python tools/bench_2d.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('FMRI_PACK_BATCHED=$v rep$rep  %.1f slices/s  %.3f ms' % (d['slices_per_s'], d['ms_per_step']))"
  done
done > gpurun_out/batched_cfg3_ab.log 2>&1
cat gpurun_out/batched_tests.log; cut -c1-150 gpurun_out/batched_ab.log; cat gpurun_out/batched_cfg3_ab.log
