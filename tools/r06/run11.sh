#!/bin/bash
# round 6: the 8-wave form of k_conv_wgrad_up_kd (default), against the 4-wave form (FMRI_UPW_KD=2) and the per-kd' kernel (FMRI_UPW_KD=0)
AB=$PWD/build/ab
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize_parity.py -x -q -p no:cacheprovider -k "upcat or parity_form or weight_gradient_is_exact or kd_sharing" 2>&1 | tail -4
echo "== prof (8-wave form; MFMAs per wave and unit: 32)"; FMRI_LIB=$AB/libfmri_hip_wuprof.so timeout 600 python tools/prof_wgrad.py --upcat 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_upw_kd8_prof.log
for v in 0 2 1; do echo "== FMRI_UPW_KD=$v"; FMRI_UPW_KD=$v timeout 600 python tools/bench_upcat.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r06_upw_kd8_layers.log
bash tools/ab_bench.sh FMRI_UPW_KD=0 FMRI_UPW_KD=2 2>&1 | tee gpurun_out/r06_upw_kd8_ab.log
