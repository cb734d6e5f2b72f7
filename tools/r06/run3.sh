#!/bin/bash
# round 6, GPU call 3: fp32 on the MFMA kernels (new tests + the fp32 engine tests), W1 weight-gradient kernel v2
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_f32_mfma.py -x -q -p no:cacheprovider 2>&1 | tail -25 | tee gpurun_out/r06_f32_tests.log
timeout 900 python -m pytest tests/test_gpu_fullsize_parity.py -x -q -p no:cacheprovider -s -k "fp32_on_the_mfma" 2>&1 | tail -30 | tee -a gpurun_out/r06_f32_tests.log
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_model.py -x -q -p no:cacheprovider 2>&1 | tail -8 | tee -a gpurun_out/r06_f32_tests.log
FMRI_WGRAD_KD_BLK=64 timeout 900 python -m pytest tests/test_gpu_fullsize_parity.py -x -q -p no:cacheprovider -k "kd_sharing or weight_gradient_is_exact" 2>&1 | tail -5
LIB=$PWD/fetal-mri-segmentation_amd/lib/libfmri_hip.so
timeout 900 python tools/bench_conv.py --libs $LIB,$LIB@FMRI_WGRAD_KD_BLK=64 --which wgrad 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_w1v2_layers.log
bash tools/ab_bench.sh FMRI_WGRAD_KD_BLK=64 2>&1 | tee gpurun_out/r06_w1v2_ab.log
