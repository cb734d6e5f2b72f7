#!/bin/bash
# round 5: per-XCD contiguous, depth-first tile walk (make exp EXP=-DFMRI_EXP_XWALK): exactness tests, per-layer A/B, HBM traffic of the exp build
mkdir -p gpurun_out/r05
EXP=$PWD/fetal-mri-segmentation_amd/lib/libfmri_hip_exp.so
FMRI_LIB=$EXP timeout 1500 python3 -m pytest tests/test_gpu_fullsize_parity.py tests/test_gpu_fullsize.py tests/test_gpu_ops.py tests/test_gpu_engine.py -x -q -m gpu > gpurun_out/r05/t_xwalk.log 2>&1; echo "xwalk tests rc=$?"
tail -n 3 gpurun_out/r05/t_xwalk.log
REPS=3 bash tools/ab_layers.sh FMRI_LIB=$EXP 2>&1 | tee gpurun_out/r05/ab_xwalk.log | head -60
FMRI_LIB=$EXP PASSES=traffic bash tools/collect_profiles.sh xw exp > gpurun_out/r05/collect_xw.log 2>&1
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/prof_xw/pmc_traffic_per_step.json"))
print({k: v for k, v in d.items() if k.startswith("_") or "family" in k or "hbm" in k.lower()} if not isinstance(d, list) else "list")
for k, v in d.items():
    if isinstance(v, dict) and "hbm_mb_corrected" in v and ("k_conv_fwd" in k):
        print(k[:80], v["hbm_mb_corrected"])
PY
