#!/bin/bash
# round 5, third GPU call: 16x16x32 forward form with the 4-deep halo-fragment ring (no spills), with and without raised MFMA-wave priority
mkdir -p gpurun_out/r05
export FMRI_MFMA16=1 FMRI_WGRAD_MFMA16=1
timeout 900 python3 -m pytest tests/test_gpu_ops.py -x -q -k "conv3d or upcat or dgrad or wgrad" > gpurun_out/r05/t_ops16b.log 2>&1; echo "ops16b rc=$?" | tee -a gpurun_out/r05/summary.txt
timeout 900 python3 -m pytest tests/test_gpu_fullsize_parity.py -x -q -k "dyadic or kd_sharing or n1_full" > gpurun_out/r05/t_full16b.log 2>&1; echo "full16b rc=$?" | tee -a gpurun_out/r05/summary.txt
unset FMRI_MFMA16 FMRI_WGRAD_MFMA16
P=$PWD/fetal-mri-segmentation_amd/lib/libfmri_hip_prof.so
FMRI_LIB=$P FMRI_MFMA16=1 python3 tools/prof_phases.py --more > gpurun_out/r05/prof_phases_16b.log 2>&1
FMRI_LIB=$P FMRI_MFMA16=1 FMRI_FWD_PRIO=1 python3 tools/prof_phases.py --more > gpurun_out/r05/prof_phases_16b_prio.log 2>&1
FMRI_LIB=$P FMRI_MFMA16=0 FMRI_FWD_PRIO=1 python3 tools/prof_phases.py --more > gpurun_out/r05/prof_phases_32_prio.log 2>&1
REPS=2 bash tools/ab_layers.sh FMRI_MFMA16=1 "FMRI_MFMA16=1 FMRI_FWD_PRIO=1" FMRI_FWD_PRIO=1 > gpurun_out/r05/ab_mfma16b.log 2>&1
grep "cyc/phase" gpurun_out/r05/prof_phases_16b.log gpurun_out/r05/prof_phases_16b_prio.log gpurun_out/r05/prof_phases_32_prio.log | cut -c1-200
cat gpurun_out/r05/ab_mfma16b.log
