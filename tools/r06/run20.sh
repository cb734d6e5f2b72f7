#!/bin/bash
# round 6: configs[3] with the decoder 'a' convs as 9-tap fused-upsample kernels instead of the 2-D parity form (FMRI_UPCAT=0)
for rep in 1 2 3; do
  for v in 1 0; do
    FMRI_UPCAT=$v python bench.py --config cfg3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('FMRI_UPCAT=$v rep$rep  %.0f slices/s  %.3f ms  mfma_frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))"
  done
done | tee gpurun_out/r06_cfg3_upcat_ab.log
