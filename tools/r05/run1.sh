#!/bin/bash
# round 5, first GPU call: correctness of the 16x16x32 instantiations, then the per-layer A/B on live data
mkdir -p gpurun_out/r05
export FMRI_MFMA16=1 FMRI_WGRAD_MFMA16=1
timeout 900 python3 -m pytest tests/test_gpu_ops.py -x -q -k "conv3d or upcat or dgrad or wgrad" > gpurun_out/r05/t_ops16.log 2>&1; echo "ops16 rc=$?" | tee -a gpurun_out/r05/summary.txt
timeout 900 python3 -m pytest tests/test_gpu_fullsize_parity.py -x -q -k "dyadic or kd_sharing or n1_full" > gpurun_out/r05/t_full16.log 2>&1; echo "full16 rc=$?" | tee -a gpurun_out/r05/summary.txt
tail -5 gpurun_out/r05/t_ops16.log gpurun_out/r05/t_full16.log
unset FMRI_MFMA16 FMRI_WGRAD_MFMA16
REPS=2 bash tools/ab_layers.sh FMRI_MFMA16=1 "FMRI_MFMA16=1 FMRI_WGRAD_MFMA16=1" FMRI_LIB=$PWD/fetal-mri-segmentation_amd/lib/libfmri_hip_r04.so > gpurun_out/r05/ab_mfma16.log 2>&1
cat gpurun_out/r05/ab_mfma16.log
