#!/bin/bash
# round 6, experiment: more workgroups than CUs in the forward family (the dispatcher hands the later ones to whichever CU falls free: balances the XCD clock spread)
AB=$PWD/build/ab
bash tools/ab_bench.sh "FMRI_LIB=$AB/libfmri_hip_gm.so FMRI_EXP_GRID_MULT=1" "FMRI_LIB=$AB/libfmri_hip_gm.so FMRI_EXP_GRID_MULT=2" "FMRI_LIB=$AB/libfmri_hip_gm.so FMRI_EXP_GRID_MULT=4" 2>&1 | tee gpurun_out/r06_grid_mult_ab.log
