#!/bin/bash
# round 6, GPU call 2: the one-wave-per-SIMD weight-gradient kernel (FMRI_WGRAD_KD_BLK=64): exactness, per-layer A/B, step A/B, unit profile
mkdir -p gpurun_out
FMRI_WGRAD_KD_BLK=64 timeout 900 python -m pytest tests/test_gpu_fullsize_parity.py -x -q -p no:cacheprovider -k "kd_sharing or weight_gradient_is_exact" 2>&1 | tail -15
LIB=$PWD/fetal-mri-segmentation_amd/lib/libfmri_hip.so
timeout 900 python tools/bench_conv.py --libs $LIB,$LIB@FMRI_WGRAD_KD_BLK=64 --which wgrad 2>&1 | tee gpurun_out/r06_w1_layers.log
for v in 32 64; do echo "== prof KD_BLK=$v"; FMRI_WGRAD_KD_BLK=$v FMRI_LIB=$PWD/build/ab/libfmri_hip_prof.so timeout 600 python tools/prof_wgrad.py 2>&1 | grep -v Warn; done | tee gpurun_out/r06_w1_prof.log
bash tools/ab_bench.sh FMRI_WGRAD_KD_BLK=64 2>&1 | tee gpurun_out/r06_w1_ab.log
