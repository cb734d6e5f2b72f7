#!/usr/bin/env python3
"""sha256 over the DEVICE CODE of libfmri_hip.so: the .text and .rodata sections (instructions, kernel descriptors, constants) of every
gfx950 code object in the library's .hip_fatbin bundles, in file order.

Why not the file's sha256: hipcc is deterministic in the code it emits, but two compilations of the same source can lay out a few
device-side .bss symbols in a different order (10 bytes of symbol values / segment sizes differ in conv3d_mfma.o; the two variants were
both seen from identical command lines), so the whole-file hash of a clean rebuild can take one of two values.  This hash is the same for
both.  `make` records it in lib/libfmri_hip.so.sha256 (tracked), tools/gputest_stamp.sh stamps the recorded GPU suite run with the hash of
the library it loaded, tests/test_profiles_fresh.py compares them.

    python tools/lib_code_hash.py [path/to/libfmri_hip.so]
"""
import hashlib
import struct
import sys

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _elf_sections(e):
    shoff = struct.unpack_from("<Q", e, 0x28)[0]
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", e, 0x3A)
    secs = []
    for k in range(shnum):
        nm, ty, fl, ad, of, si = struct.unpack_from("<IIQQQQ", e, shoff + k * shentsize)
        secs.append((nm, ty, of, si))
    st = secs[shstrndx][2]

    def name(nm):
        return e[st + nm:e.index(b"\0", st + nm)].decode()
    return [(name(nm), ty, of, si) for nm, ty, of, si in secs]


def code_hash(path):
    data = open(path, "rb").read()
    h = hashlib.sha256()
    n_obj = 0
    pos = data.find(MAGIC)
    while pos >= 0:
        n = struct.unpack_from("<Q", data, pos + 24)[0]
        off = pos + 32
        for _ in range(n):
            o, sz, tl = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + tl]
            off += tl
            if sz and b"amdgcn" in triple:
                e = data[pos + o:pos + o + sz]
                assert e[:4] == b"\x7fELF", "device entry is not an ELF image"
                for sname, ty, of, si in _elf_sections(e):
                    if sname in (".text", ".rodata") and ty != 8:
                        h.update(sname.encode())
                        h.update(e[of:of + si])
                n_obj += 1
        pos = data.find(MAGIC, pos + 1)
    assert n_obj > 0, "no device code objects found in %s" % path
    return h.hexdigest(), n_obj


if __name__ == "__main__":
    import os
    p = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fetal-mri-segmentation_amd", "lib",
                                                          "libfmri_hip.so")
    print(code_hash(p)[0])
