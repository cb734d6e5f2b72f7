#!/bin/bash
# round 6, weight-image pack kernels (all loads in flight; only the matching taps in the parity pre-sums): per launch, tests, and what the repack costs
# the step with the library before (build/ab/libfmri_hip_base.so) and after
mkdir -p gpurun_out
echo "== after"  > gpurun_out/pack_layers.log; python tools/bench_pack.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/pack_layers.log
echo "== before" >> gpurun_out/pack_layers.log; FMRI_LIB=$PWD/build/ab/libfmri_hip_base.so python tools/bench_pack.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/pack_layers.log
python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py -x -q -k "pack or upcat or engine or unet" 2>&1 | tail -3 > gpurun_out/pack_tests.log
for rep in 1 2; do
  echo "== after rep$rep"; python tools/r06/pack_cost.py 5 40 2>&1 | grep -v amdgpu.ids
  echo "== before rep$rep"; FMRI_LIB=$PWD/build/ab/libfmri_hip_base.so python tools/r06/pack_cost.py 5 40 2>&1 | grep -v amdgpu.ids
done > gpurun_out/pack_cost_ab.log
cat gpurun_out/pack_layers.log gpurun_out/pack_tests.log gpurun_out/pack_cost_ab.log
