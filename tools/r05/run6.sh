#!/bin/bash
# round 5, sixth GPU call: buffer-descriptor DMA in all weight-gradient kernels + buffer halo DMA for every single-source launch (now the
# defaults): correctness, then A/B against the round-4 library and of the weight-gradient stream's priority
mkdir -p gpurun_out/r05
python3 -c "import torch; print('stream priority range', torch.cuda.Stream.priority_range())" 2>&1 | tail -1
timeout 1200 python3 -m pytest tests/test_gpu_ops.py -x -q > gpurun_out/r05/t_ops_d.log 2>&1; echo "ops rc=$?" | tee -a gpurun_out/r05/summary.txt
timeout 1200 python3 -m pytest tests/test_gpu_fullsize_parity.py tests/test_gpu_engine.py -x -q > gpurun_out/r05/t_full_d.log 2>&1; echo "full+engine rc=$?" | tee -a gpurun_out/r05/summary.txt
tail -n 3 gpurun_out/r05/t_ops_d.log gpurun_out/r05/t_full_d.log
REPS=3 bash tools/ab_layers.sh FMRI_LIB=$PWD/fetal-mri-segmentation_amd/lib/libfmri_hip_r04.so FMRI_WGRAD_PRIO=1 FMRI_WGRAD_PRIO=-1 > gpurun_out/r05/ab_prio.log 2>&1
grep -v amdgpu.ids gpurun_out/r05/ab_prio.log
