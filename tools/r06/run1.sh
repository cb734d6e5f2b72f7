#!/bin/bash
# round 6, GPU call 1: stamp the GPU suite on the split sources, cfg3 rocprof evidence, A/B of the round-5 library against this tree
HEAD=$1
bash tools/gputest_stamp.sh r06 > gpurun_out/r06_gputest_tail.txt 2>&1
echo "gputest rc=$?"; tail -3 gpurun_out/r06_gputest_tail.txt
bash tools/r06/collect_cfg3.sh r06 $HEAD
bash tools/ab_bench.sh FMRI_LIB=$PWD/build/ab/libfmri_hip_r05.so 2>&1 | tee gpurun_out/r06_ab_split_vs_r05.log
