#!/bin/bash
# round 6: the kd'-sharing parity-form weight-gradient kernel (FMRI_UPW_KD=1): exactness, per-layer, step A/B
mkdir -p gpurun_out
FMRI_UPW_KD=1 timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize_parity.py -x -q -p no:cacheprovider -k "upcat or parity_form or weight_gradient_is_exact" 2>&1 | tail -12
for v in 0 1; do echo "== FMRI_UPW_KD=$v"; FMRI_UPW_KD=$v timeout 600 python tools/bench_upcat.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r06_upw_kd_layers.log
bash tools/ab_bench.sh FMRI_UPW_KD=1 2>&1 | tee gpurun_out/r06_upw_kd_ab.log
