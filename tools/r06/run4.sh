#!/bin/bash
# round 6, GPU call 4: fp32 parity form + weight gradient tests, full-size fp32 figures (MFMA vs VALU kernels), W1 variants a/b/c, kd threshold
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_f32_mfma.py -q -p no:cacheprovider 2>&1 | tail -25 | tee gpurun_out/r06_f32_tests.log
FMRI_MEASURE=1 timeout 900 python -m pytest tests/test_gpu_fullsize_parity.py -x -q -p no:cacheprovider -s -k "fp32_on_the_mfma" 2>&1 | grep -E "kernel|MEASURED|passed|failed|Error" | tee gpurun_out/r06_f32_fullsize_mfma.log
FMRI_MEASURE=1 FMRI_F32_MFMA=0 timeout 900 python -m pytest tests/test_gpu_fullsize_parity.py -x -q -p no:cacheprovider -s -k "fp32_on_the_mfma" 2>&1 | grep -E "kernel|MEASURED|passed|failed|Error" | tee gpurun_out/r06_f32_fullsize_valu.log
AB=$PWD/build/ab
LIB=$PWD/fetal-mri-segmentation_amd/lib/libfmri_hip.so
timeout 900 python tools/bench_conv.py --libs $LIB,$AB/libfmri_hip_w1a.so@FMRI_WGRAD_KD_BLK=64,$AB/libfmri_hip_w1b.so@FMRI_WGRAD_KD_BLK=64,$AB/libfmri_hip_w1c.so@FMRI_WGRAD_KD_BLK=64,$LIB@FMRI_WGRAD_KD_MIN_GFLOP=20 --which wgrad 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_w1v3_layers.log
for v in a b c; do echo "== prof w1$v"; FMRI_WGRAD_KD_BLK=64 FMRI_LIB=$AB/libfmri_hip_w1${v}_prof.so timeout 600 python tools/prof_wgrad.py 2>&1 | grep -E "enc0b|dec1b|dec0b"; done | tee gpurun_out/r06_w1v3_prof.log
bash tools/ab_bench.sh "FMRI_LIB=$AB/libfmri_hip_w1a.so FMRI_WGRAD_KD_BLK=64" "FMRI_LIB=$AB/libfmri_hip_w1b.so FMRI_WGRAD_KD_BLK=64" FMRI_WGRAD_KD_MIN_GFLOP=20 2>&1 | tee gpurun_out/r06_w1v3_ab.log
