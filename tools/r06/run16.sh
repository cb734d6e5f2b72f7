#!/bin/bash
# round 6 experiment: dynamic tile hand-out in k_conv_fwd_ws (FMRI_FWD_DYN = % of a workgroup's tiles that stay static)
AB=$PWD/build/ab
export FMRI_LIB=$AB/libfmri_hip_dyn.so
FMRI_FWD_DYN=75 timeout 900 python -m pytest tests/test_gpu_fullsize_parity.py tests/test_gpu_fullsize.py -x -q -p no:cacheprovider -k "forward_is_exact or input_gradient_is_exact or n1_full_size_bf16 or bit_identical or tail_fusion" 2>&1 | tail -5
FMRI_FWD_DYN=50 timeout 900 python -m pytest tests/test_gpu_fullsize_parity.py -x -q -p no:cacheprovider -k "forward_is_exact or input_gradient_is_exact or n4_live" 2>&1 | tail -3
unset FMRI_LIB
bash tools/ab_bench.sh "FMRI_LIB=$AB/libfmri_hip_dyn.so FMRI_FWD_DYN=0" "FMRI_LIB=$AB/libfmri_hip_dyn.so FMRI_FWD_DYN=50" "FMRI_LIB=$AB/libfmri_hip_dyn.so FMRI_FWD_DYN=75" "FMRI_LIB=$AB/libfmri_hip_dyn.so FMRI_FWD_DYN=90" 2>&1 | tee gpurun_out/r06_fwd_dyn_ab.log
