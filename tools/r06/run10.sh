#!/bin/bash
# round 6: tuning of k_conv_wgrad_up_kd (prefetch depth, DMA spread), its unit profile, and the exactness test with the new default
AB=$PWD/build/ab
timeout 900 python -m pytest tests/test_gpu_fullsize_parity.py -x -q -p no:cacheprovider -k "kd_sharing" 2>&1 | tail -3
echo "== prof"; FMRI_LIB=$AB/libfmri_hip_wuprof.so timeout 600 python tools/prof_wgrad.py --upcat 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_upw_kd_prof.log
for v in base pf5 spd1 spd3; do echo "== $v"; FMRI_LIB=$AB/libfmri_hip_wu$v.so timeout 600 python tools/bench_upcat.py 2>&1 | grep total; done | tee gpurun_out/r06_upw_kd_tune.log
