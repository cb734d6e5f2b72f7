#!/bin/bash
# round 6: parity-form weight gradient with both pw classes per workgroup (FMRI_UPW_KD=3): exactness, per layer, whole steps
FMRI_UPW_KD=3 timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize_parity.py -x -q -p no:cacheprovider -k "upcat or parity_form or weight_gradient_is_exact" 2>&1 | tail -5
for v in 1 3; do echo "== FMRI_UPW_KD=$v"; FMRI_UPW_KD=$v timeout 600 python tools/bench_upcat.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r06_upw_kd2_layers.log
bash tools/ab_bench.sh FMRI_UPW_KD=3 2>&1 | tee gpurun_out/r06_upw_kd2_ab.log
