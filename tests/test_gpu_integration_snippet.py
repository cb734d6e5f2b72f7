"""INTEGRATION.md section 3 - the raw ctypes binding a maintainer of the reference would add - executed VERBATIM.

CPU part: the code block's argtypes list has as many entries as include/fmri_hip.h declares parameters for fmri_conv3d_fwd (round 1
shipped a 19-entry list against a 20-parameter prototype: copied as written it passed the stream pointer as `planar`).
GPU part: the block runs as it stands in the document and its output is compared with a float64 restatement of
relu(conv3x3x3(concat([up2(x_low), x_skip])) + b)  (reference unet3d/unet.py:61,102,113,138)."""
import os
import re

import pytest

from conftest import ROOT


def _snippet():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = doc[doc.index("## 3. Binding the C ABI directly"):]
    m = re.search(r"```python\n(.*?)```", sec, flags=re.S)
    assert m, "INTEGRATION.md section 3 lost its python block"
    return m.group(1)


def _header_param_count(name):
    src = open(os.path.join(ROOT, "include", "fmri_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    m = re.search(r"\b%s\s*\((.*?)\)\s*;" % name, src, flags=re.S)
    assert m, name
    return len([a for a in m.group(1).split(",") if a.strip()])


def test_snippet_arity_matches_the_header():
    code = _snippet()
    m = re.search(r"fmri_conv3d_fwd\.argtypes\s*=\s*\[(.*?)\]", code, flags=re.S)
    n_types = len([a for a in m.group(1).split(",") if a.strip()])
    assert n_types == _header_param_count("fmri_conv3d_fwd") == 20
    call = re.search(r"L\.fmri_conv3d_fwd\((.*?)\)\nassert", code, flags=re.S).group(1)
    depth, n_args, cur = 0, 0, ""
    for ch in call:                                   # split the call's arguments at top-level commas
        if ch in "([":
            depth += 1
        elif ch in ")]":
            depth -= 1
        if ch == "," and depth == 0:
            n_args += 1
            cur = ""
        else:
            cur += ch
    n_args += 1 if cur.strip() else 0
    assert n_args == 20, n_args


@pytest.mark.gpu
def test_snippet_runs_verbatim_and_matches_fp64():
    import torch
    from fmri_hip._lib import LIB_PATH
    from gpu_util import assert_close, ref_conv_fwd, rnd, f64
    N, D, H, W = 1, 8, 16, 32
    bf = torch.bfloat16
    x_low = rnd((N, D // 2, H // 2, W // 2, 128), 1, bf)
    x_skip = rnd((N, D, H, W, 64), 2, bf)
    w = rnd((27, 64, 192), 3, bf, scale=0.03)
    bias = rnd((64,), 4, torch.float32, scale=0.1)
    y = torch.full((N, D, H, W, 64), float("nan"), dtype=bf, device="cuda")
    ns = dict(FMRI_LIB_PATH=LIB_PATH, x_low=x_low, x_skip=x_skip, w=w, bias=bias, y=y, N=N, D=D, H=H, W=W)
    exec(compile(_snippet(), "INTEGRATION.md#3", "exec"), ns)
    torch.cuda.synchronize()
    assert ns["rc"] == 0
    ref = ref_conv_fwd(f64(x_low), f64(x_skip), True, f64(w), f64(bias), 1)
    # bf16 inputs are exact in fp64; the only roundings are the fp32 accumulation and the final bf16 store (2^-9 relative)
    err = assert_close(y, ref, 2.0 ** -8, 2e-5, "INTEGRATION.md snippet")
    print("snippet: max err / scale = %.2e" % err)
