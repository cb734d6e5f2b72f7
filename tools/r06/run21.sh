#!/bin/bash
# round 6, first-layer kernels: the GPU tests that run them, then the step A/B against the library before the change (build/ab/libfmri_hip_base.so)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py tests/test_gpu_fullsize_parity.py tests/test_gpu_model.py -x -q 2>&1 | tail -5 > gpurun_out/first_suite.log
python tools/r06/bench_first.py --libs build/ab/libfmri_hip_base.so,fetal-mri-segmentation_amd/lib/libfmri_hip.so 2>&1 | grep -v amdgpu.ids > gpurun_out/first_layers.log
bash tools/ab_bench.sh FMRI_LIB=$PWD/build/ab/libfmri_hip_base.so > gpurun_out/first_ab.log 2>&1
python tools/bench_2d.py > gpurun_out/first_cfg3_new.log 2>&1
FMRI_LIB=$PWD/build/ab/libfmri_hip_base.so python tools/bench_2d.py > gpurun_out/first_cfg3_base.log 2>&1
python tools/bench_2d.py > gpurun_out/first_cfg3_new2.log 2>&1
FMRI_LIB=$PWD/build/ab/libfmri_hip_base.so python tools/bench_2d.py > gpurun_out/first_cfg3_base2.log 2>&1
cat gpurun_out/first_suite.log gpurun_out/first_layers.log gpurun_out/first_ab.log; tail -2 gpurun_out/first_cfg3_*.log
