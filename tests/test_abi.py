"""The C-ABI library loads and exports exactly the symbols include/fmri_hip.h declares (no compute calls: CPU-only test)."""
import os
import re
import subprocess

import pytest

from conftest import ROOT


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "fmri_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fmri_[a-z0-9_]+)\s*\(", src)))


def test_header_matches_binding_table():
    from fmri_hip._lib import SIGNATURES
    assert _header_symbols() == sorted(SIGNATURES)


def test_library_exports_every_symbol():
    from fmri_hip._lib import LIB_PATH, SIGNATURES, lib
    if not os.path.exists(LIB_PATH):
        import __graft_entry__ as g
        g.build()
    L = lib()
    for name in SIGNATURES:
        assert hasattr(L, name), name
    out = subprocess.check_output(["nm", "-D", "--defined-only", LIB_PATH]).decode()
    exported = sorted(set(re.findall(r"\bT (fmri_[a-z0-9_]+)\b", out)))
    assert exported == _header_symbols()
    assert L.fmri_version() >= 100
    assert L.fmri_error_string(-1).decode() == "unsupported shape"


def test_mfma_dispatch_query():
    from fmri_hip._lib import BF16, F32, lib
    L = lib()
    # BASELINE config 2 shapes: every conv except the single-channel first layer takes the MFMA kernels
    assert L.fmri_conv3d_uses_mfma(32, 0, 64, 64, 128, 128, BF16) == 3
    assert L.fmri_conv3d_uses_mfma(128, 64, 64, 64, 128, 128, BF16) == 3
    assert L.fmri_conv3d_uses_mfma(256, 0, 512, 8, 16, 16, BF16) == 3
    assert L.fmri_conv3d_uses_mfma(1, 0, 32, 64, 128, 128, BF16) == 4
    # fp32 (round 6): the fp32 instantiation of the same kernels where channels come in multiples of 16 (forward) / 32 (weight gradient)
    assert L.fmri_conv3d_uses_mfma(32, 0, 64, 64, 128, 128, F32) == 3
    assert L.fmri_conv3d_uses_mfma(16, 0, 32, 64, 128, 128, F32) == 1
    assert L.fmri_conv3d_uses_mfma(8, 0, 16, 16, 64, 64, F32) == 0       # config 1 (base 8) in fp32 -> generic path
    assert L.fmri_conv3d_uses_mfma(8, 0, 16, 16, 64, 64, BF16) == 0     # config 1 (base 8) -> generic path


def test_ops_refuse_cpu_tensors():
    import torch
    from fmri_hip import ops
    with pytest.raises(RuntimeError):
        ops.maxpool_fwd(torch.zeros(1, 2, 2, 2, 4), torch.zeros(1, 1, 1, 1, 4))
