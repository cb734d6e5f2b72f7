#!/bin/bash
# round 5, seventh GPU call: kernel timeline of the step + the weight-gradient kernels' per-unit cycle split after the buffer-DMA change
mkdir -p gpurun_out/r05
bash tools/trace_step.sh > gpurun_out/r05/trace_step.log 2>&1
cp gpurun_out/trace_step/summary.txt gpurun_out/r05/trace_step_summary.txt
FMRI_LIB=$PWD/fetal-mri-segmentation_amd/lib/libfmri_hip_prof.so python3 tools/prof_wgrad.py > gpurun_out/r05/prof_wgrad.log 2>&1
FMRI_LIB=$PWD/fetal-mri-segmentation_amd/lib/libfmri_hip_prof.so FMRI_WGRAD_MFMA16=1 python3 tools/prof_wgrad.py > gpurun_out/r05/prof_wgrad16.log 2>&1
FMRI_LIB=$PWD/fetal-mri-segmentation_amd/lib/libfmri_hip_prof.so python3 tools/prof_phases.py --more > gpurun_out/r05/prof_phases_buf.log 2>&1
tail -50 gpurun_out/r05/trace_step_summary.txt; grep -v amdgpu gpurun_out/r05/prof_wgrad.log; grep -v amdgpu gpurun_out/r05/prof_wgrad16.log; grep "cyc/phase" gpurun_out/r05/prof_phases_buf.log | cut -c1-210
