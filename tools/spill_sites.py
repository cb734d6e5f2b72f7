#!/usr/bin/env python3
"""Where does a kernel spill?  Compiles a .hip file to gfx950 assembly and reports, per kernel whose mangled name contains the pattern, the
scratch instructions before / inside / after the span of its MFMA instructions and those within 40 lines of an MFMA (a spill inside the
matrix loop is a performance bug, one in a prologue or in a producer wave's path usually is not).
usage: tools/spill_sites.py <file.hip> <mangled-name substring> [extra hipcc flags]"""
import bisect
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

src, pat = sys.argv[1], sys.argv[2]
asm = "/tmp/_spill_sites.s"
subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-comment", "-S", "--cuda-device-only", "-o", asm, src] + per_file_flags(src) + sys.argv[3:],
               check=True, stderr=subprocess.DEVNULL)
lines = open(asm).read().split("\n")
for i, l in enumerate(lines):
    if l.startswith("_Z") and pat in l and l.rstrip().split(":")[0].startswith("_Z") and "@" in l:
        end = i
        while not lines[end].startswith(".Lfunc_end"):
            end += 1
        L = lines[i:end]
        sc = [k for k, x in enumerate(L) if "scratch_" in x]
        mf = [k for k, x in enumerate(L) if "v_mfma" in x]
        name = subprocess.run(["c++filt", l.split(":")[0]], capture_output=True, text=True).stdout.strip().split("(")[0]
        if not mf:
            print("%-60s no MFMA; scratch ops %d" % (name[-60:], len(sc)))
            continue
        near = 0
        for k in sc:
            j = bisect.bisect(mf, k)
            d = min(abs(k - mf[j - 1]) if j > 0 else 10 ** 9, abs(mf[j] - k) if j < len(mf) else 10 ** 9)
            near += d < 40
        print("%-60s mfma %4d | scratch ops %4d: before %3d  inside span %3d (near an MFMA %3d)  after %3d | stores %3d loads(non-lds) %3d barriers %3d" % (
            name[-60:], len(mf), len(sc), sum(k < mf[0] for k in sc), sum(mf[0] <= k <= mf[-1] for k in sc), near, sum(k > mf[-1] for k in sc),
            sum("global_store" in x for x in L), sum("global_load_dword" in x and "lds" not in x for x in L), sum("s_barrier" in x for x in L)))
