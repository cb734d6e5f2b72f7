#!/bin/bash
# round 5: kd-sharing weight-gradient kernel with producer waves (FMRI_WGRAD_KD_PROD=1|2): correctness, exclusive launches, step A/B
mkdir -p gpurun_out/r05
for p in 2 1; do
  FMRI_WGRAD_KD_PROD=$p timeout 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize_parity.py -x -q -k "wgrad or weight_gradient or kd_sharing or n1_full or n4_bench" > gpurun_out/r05/t_kdprod$p.log 2>&1; echo "kd prod $p tests rc=$?" | tee -a gpurun_out/r05/summary.txt; tail -n 2 gpurun_out/r05/t_kdprod$p.log
done
python3 - <<'PY' 2>&1 | grep -v amdgpu | tee gpurun_out/r05/kdprod_micro.log
import json, os, subprocess, sys
rows = {}
for rd in range(3):
    for p in ("0", "1", "2"):
        env = dict(os.environ, FMRI_WGRAD_KD_PROD=p)
        o = subprocess.check_output([sys.executable, "tools/r05/wgrad_variants.py", "--child"], env=env).decode()
        rows.setdefault(p, []).append(json.loads([x for x in o.splitlines() if x.startswith("RESULT ")][0][7:]))
for p, rs in rows.items():
    print("FMRI_WGRAD_KD_PROD=%s" % p, "  ".join("%s min %.4f" % (k, min(r[k] for r in rs)) for k in rs[0]))
PY
REPS=3 bash tools/ab_layers.sh FMRI_WGRAD_KD_PROD=1 FMRI_WGRAD_KD_PROD=2 > gpurun_out/r05/ab_kdprod.log 2>&1
grep -v amdgpu gpurun_out/r05/ab_kdprod.log | head -6; grep -v amdgpu gpurun_out/r05/ab_kdprod.log | grep "wgrad"
