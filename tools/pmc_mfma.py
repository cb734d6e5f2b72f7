#!/usr/bin/env python3
"""MFMA utilisation per kernel from a rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE; optional second pass with the SQ
wave-state counters).  usage: pmc_mfma.py <dir with pmc_mfma/ [and pmc_sq/]> <steps profiled> [git head] [kernel source hash]

  busy      = SQ_VALU_MFMA_BUSY_CYCLES summed over the launch: shader cycles in which a SIMD's matrix pipe is busy; a
              v_mfma_f32_32x32x16_bf16 holds it for 32 (/opt/skills/guides/MI355X_MICROARCH.md, cycle constants), so
              busy / 32 = MFMA wave-instructions issued = MACs / 16,384 - checked against the launch's known MAC count by bench.py's callers
  active    = GRBM_GUI_ACTIVE / 8: the counter is the sum over the 8 XCDs of their active cycles
  mfma_util = busy / (active x 256 CUs x 4 SIMDs): the fraction of matrix-pipe cycles in use while the kernel ran, at whatever clock the
              chip held (a clock-independent figure; FLOP-derived fractions of the 2.5 PF peak also carry clock / 2.4 GHz)"""
import csv
import glob
import json
import os
import re
import sys

N_SIMD = 256 * 4


def load(d):
    """-> {kernel: {counter: [launches, sum]}}"""
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = re.sub(r"^void ", "", row["Kernel_Name"])
                name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0][:48]
                e = out.setdefault(name, {}).setdefault(row["Counter_Name"], [0, 0.0])
                e[0] += 1
                e[1] += float(row["Counter_Value"])
    return out


def family(name):
    if name.startswith("k_conv_fwd_"):
        return "conv_fwd_dgrad (k_conv_fwd_*)"
    if name.startswith("k_conv_wgrad"):
        return "conv_wgrad (k_conv_wgrad_*)"
    if name.startswith("k_conv_first"):
        return "first conv (k_conv_first_*)"
    return None


def main():
    root, steps = sys.argv[1], int(sys.argv[2])
    a = load(os.path.join(root, "pmc_mfma"))
    sq = load(os.path.join(root, "pmc_sq")) if os.path.isdir(os.path.join(root, "pmc_sq")) else {}
    res = {"_meta": {"git_head": sys.argv[3] if len(sys.argv) > 3 else None, "kernel_source_hash": sys.argv[4] if len(sys.argv) > 4 else None,
                     "steps_profiled": steps, "command": "bench.py --steps 1 --warmup 1 --serialize-streams (one stream: exclusive launches)",
                     "definition": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)"}, "kernels": {}, "families": {}}
    fam = {}
    for k, c in sorted(a.items()):
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
            continue
        busy, act = c["SQ_VALU_MFMA_BUSY_CYCLES"][1], c["GRBM_GUI_ACTIVE"][1] / 8.0
        if busy <= 0:
            continue
        e = {"launches_per_step": c["GRBM_GUI_ACTIVE"][0] / steps, "mfma_busy_cycles_per_step": busy / steps, "active_cycles_per_step": act / steps,
             "mfma_util": busy / (act * N_SIMD) if act else None, "mfma_wave_instr_32x32x16_per_step": busy / 32.0 / steps}
        for name, v in sq.get(k, {}).items():
            e[name + "_per_step"] = v[1] / steps
        res["kernels"][k] = e
        f = family(k)
        if f:
            t = fam.setdefault(f, [0.0, 0.0, 0])
            t[0] += busy
            t[1] += act
            t[2] += c["GRBM_GUI_ACTIVE"][0]
    for f, (busy, act, n) in fam.items():
        res["families"][f] = {"launches_per_step": n / steps, "mfma_busy_cycles_per_step": busy / steps, "active_cycles_per_step": act / steps,
                              "mfma_util": busy / (act * N_SIMD) if act else None, "tmac_per_step_if_32x32x16": busy / 32.0 * 16384 / steps / 1e12}
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
