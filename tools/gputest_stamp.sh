#!/bin/bash
# Runs the whole -m gpu suite the way the driver does (pytest -x -q) and keeps its log STAMPED with the hash of the kernel sources it
# ran on: gpurun_out/<tag>_gputest_final.log (copy it to profiles/).  tests/test_profiles_fresh.py (CPU) fails while that stamp differs
# from the tree's kernel_source_hash, i.e. whenever a kernel changed after the last recorded green run.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gputest_stamp.sh r04'
TAG=${1:-r04}
mkdir -p gpurun_out
HASH=$(python3 -c "import bench; print(bench.kernel_source_hash())")
SO=$(python3 tools/lib_code_hash.py fetal-mri-segmentation_amd/lib/libfmri_hip.so | cut -c1-16)
LOG=gpurun_out/${TAG}_gputest_final.log
{
    echo "# kernel_source_hash=$HASH libfmri_hip_code_sha256_16=$SO tag=$TAG date=$(date -u +%Y-%m-%dT%H:%MZ)"
    echo "# command: python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=10"
} > $LOG
python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=10 >> $LOG 2>&1
RC=$?
echo "# rc=$RC" >> $LOG
tail -8 $LOG
exit $RC
