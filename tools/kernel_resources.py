#!/usr/bin/env python3
"""Registers / spills / scratch per kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage), demangled.
usage: tools/kernel_resources.py fetal-mri-segmentation_amd/csrc/conv3d_mfma.hip [name filter]"""
import re
import subprocess
import sys


def per_file_flags(src):
    """the extra flags csrc/Makefile gives this file (FLAGS_<stem> = ...): the numbers printed here must describe the binary that ships"""
    import os
    mk = os.path.join(os.path.dirname(os.path.abspath(src)), "Makefile")
    stem = os.path.splitext(os.path.basename(src))[0]
    if not os.path.exists(mk):
        return []
    m = re.search(r"^FLAGS_%s\s*=\s*(.*)$" % re.escape(stem), open(mk).read(), re.M)
    return m.group(1).split() if m else []

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-comment", "-Rpass-analysis=kernel-resource-usage",
                      "-c", src, "-o", "/dev/null"] + per_file_flags(src) + sys.argv[3:], capture_output=True, text=True).stderr
cur = None
rows = {}
for ln in out.splitlines():
    m = re.search(r"remark: (?:\s*)Function Name: (\S+)", ln)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+(?:\[[^\]]*\])?): (\d+)", ln)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
for mangled, name in zip(rows, names):
    name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0].replace("void ", "")
    if flt not in name:
        continue
    r = rows[mangled]
    print("%-52s VGPR %3d AGPR %3d SGPR %3d | spill V %3d S %3d | scratch %4d B | occ %d | LDS %6d" % (
        name, r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("SGPRs", -1), r.get("VGPRs Spill", -1), r.get("SGPRs Spill", -1),
        r.get("ScratchSize [bytes/lane]", -1), r.get("Occupancy [waves/SIMD]", -1), r.get("LDS Size [bytes/block]", -1)))
