#!/bin/bash
mkdir -p gpurun_out/r05
REPS=3 bash tools/ab_layers.sh FMRI_WGRAD_WS=1 FMRI_WGRAD_WS=2 > gpurun_out/r05/ab_wgrad_ws.log 2>&1
grep -v amdgpu gpurun_out/r05/ab_wgrad_ws.log | head -6; grep -v amdgpu gpurun_out/r05/ab_wgrad_ws.log | grep wgrad
