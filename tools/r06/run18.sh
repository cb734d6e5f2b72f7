#!/bin/bash
# round 6: ReLU-mask lines of the symmetric (2-D / cube) kernel's epilogue requested in front of the barrier: correctness + configs[3] A/B
AB=$PWD/build/ab
FMRI_LIB=$AB/libfmri_hip_mk2.so timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py tests/test_gpu_fullsize_parity.py -x -q -p no:cacheprovider -k "planar or 2d or cube or noact_mask or pack_and_dgrad" 2>&1 | tail -4
for rep in 1 2 3; do
  for lib in "" "$AB/libfmri_hip_mk.so" "$AB/libfmri_hip_mk2.so"; do
    FMRI_LIB=$lib python bench.py --config cfg3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-10s rep$rep  %.0f slices/s  %.3f ms  mfma_frac %.3f' % ('$(basename "${lib:-product}")', d['value'], d['ms_per_step'], d['roofline']['frac']))"
  done
done | tee gpurun_out/r06_cfg3_mask_res_hoist_ab.log
