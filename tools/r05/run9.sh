#!/bin/bash
# round 5, ninth GPU call: phase profile of the 16x16x32 forward form now that every launch's producers issue without vector arithmetic
mkdir -p gpurun_out/r05
P=$PWD/fetal-mri-segmentation_amd/lib/libfmri_hip_prof.so
FMRI_LIB=$P FMRI_MFMA16=1 python3 tools/prof_phases.py --more > gpurun_out/r05/prof_phases_16_buf.log 2>&1
grep "cyc/phase" gpurun_out/r05/prof_phases_16_buf.log | cut -c1-210
