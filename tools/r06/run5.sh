#!/bin/bash
# round 6, GPU call 5: stamp the suite on the F32 kernels, variant d (spread DMA in the 32-block kernel), the default bench line with parity_mode
mkdir -p gpurun_out
bash tools/gputest_stamp.sh r06 > gpurun_out/r06_gputest_tail.txt 2>&1
echo "gputest rc=$?"; tail -4 gpurun_out/r06_gputest_tail.txt
AB=$PWD/build/ab
LIB=$PWD/fetal-mri-segmentation_amd/lib/libfmri_hip.so
timeout 900 python tools/bench_conv.py --libs $LIB,$AB/libfmri_hip_w1d.so --which wgrad 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_kd32_spread_layers.log
bash tools/ab_bench.sh "FMRI_LIB=$AB/libfmri_hip_w1d.so" 2>&1 | tee gpurun_out/r06_kd32_spread_ab.log
( time python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err ) 2>&1 | tail -3
tail -c 1500 gpurun_out/r06_bench_default.json
