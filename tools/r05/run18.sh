#!/bin/bash
# round 5: accumulators re-initialised by loads from the bias instead of register moves (make exp EXP=-DFMRI_EXP_ACC_RELOAD): dyadic tests, then A/B
mkdir -p gpurun_out/r05
EXP=$PWD/fetal-mri-segmentation_amd/lib/libfmri_hip_exp.so
FMRI_LIB=$EXP timeout 1500 python3 -m pytest tests/test_gpu_fullsize_parity.py tests/test_gpu_fullsize.py tests/test_gpu_ops.py -x -q -m gpu -k "not stamp" > gpurun_out/r05/t_reload.log 2>&1; echo "reload tests rc=$?"
tail -n 3 gpurun_out/r05/t_reload.log
REPS=3 bash tools/ab_layers.sh FMRI_LIB=$EXP 2>&1 | tee gpurun_out/r05/ab_acc_reload.log | head -60
