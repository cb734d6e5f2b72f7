#!/bin/bash
# round 5, second GPU call: where the 16x16x32 forward kernel loses its cycles (phase profile of both forms), and the A/B of the
# weight-gradient form alone + the round-4 library as reference
mkdir -p gpurun_out/r05
P=$PWD/fetal-mri-segmentation_amd/lib/libfmri_hip_prof.so
FMRI_LIB=$P FMRI_MFMA16=0 python3 tools/prof_phases.py --more > gpurun_out/r05/prof_phases_32.log 2>&1
FMRI_LIB=$P FMRI_MFMA16=1 python3 tools/prof_phases.py --more > gpurun_out/r05/prof_phases_16.log 2>&1
REPS=2 bash tools/ab_layers.sh FMRI_WGRAD_MFMA16=1 FMRI_LIB=$PWD/fetal-mri-segmentation_amd/lib/libfmri_hip_r04.so > gpurun_out/r05/ab_wgrad16.log 2>&1
cat gpurun_out/r05/prof_phases_32.log gpurun_out/r05/prof_phases_16.log gpurun_out/r05/ab_wgrad16.log
