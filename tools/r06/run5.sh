#!/bin/bash
# round 6, GPU call 5: stamp the suite on the F32 kernels, variant d (spread DMA in the 32-block kernel), the default bench line with parity_mode
mkdir -p gpurun_out
bash tools/gputest_stamp.sh r06 > gpurun_out/r06_gputest_tail.txt 2>&1
echo "gputest rc=$?"; tail -4 gpurun_out/r06_gputest_tail.txt
# (the first run of this script also compared a build of the 32-block weight-gradient kernel with its DMA pieces between the MFMAs, -DFMRI_KD32_SPREAD=1:
#  profiles/r06_kd32_spread_*.log; the flag is gone)
( time python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err ) 2>&1 | tail -3
tail -c 1500 gpurun_out/r06_bench_default.json
