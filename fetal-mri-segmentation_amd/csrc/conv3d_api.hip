// C-ABI dispatch for the 3x3x3 convolution family: picks the bf16 MFMA kernels when the shape allows, else the generic path.
#include "common.h"

int conv3d_fwd_generic(const void*, int, int, int, const void*, int, const void*, const float*, const void*, void*, int, int, int, int, int,
                       int, float, int, hipStream_t);
int conv3d_wgrad_generic(const void*, int, int, int, const void*, int, const void*, float*, float*, int, int, int, int, int, int,
                         hipStream_t);
bool conv3d_fwd_mfma_ok(int C0, int C1, int Cout, int D, int H, int W, int dtype);
bool conv3d_wgrad_mfma_ok(int C0, int C1, int Cout, int D, int H, int W, int dtype);
int conv3d_fwd_mfma(const void*, int, int, int, const void*, int, const void*, const float*, const void*, void*, int, int, int, int, int, int,
                    float, hipStream_t);
int conv3d_wgrad_mfma(const void*, int, int, int, const void*, int, const void*, float*, float*, int, int, int, int, int, void*, int64_t,
                      hipStream_t);
int64_t conv3d_wgrad_mfma_ws_bytes(int C0, int C1, int Cout, int N, int D, int H, int W, int planar);

bool conv3d_first_ok(int C0, int C1, int Cout, int D, int H, int W, int dtype, int up0);
int conv3d_first_fwd(const void*, const void*, const float*, void*, int, int, int, int, int, int, float, hipStream_t);
int conv3d_first_wgrad(const void*, const void*, float*, float*, int, int, int, int, int, hipStream_t);

static int check_common(const void* src0, int C0, int up0, int planar, const void* src1, int C1, int N, int D, int H, int W, int Cout) {
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cout <= 0 || C0 <= 0 || C1 < 0) return FMRI_E_SHAPE;
    if (!src0 || (C1 > 0 && !src1)) return FMRI_E_SHAPE;
    if (up0 && (((planar ? 0 : D) | H | W) & 1)) return FMRI_E_SHAPE;
    return FMRI_OK;
}

extern "C" int fmri_conv3d_uses_mfma(int C0, int C1, int Cout, int D, int H, int W, int dtype) {
    // bit 0: forward/dgrad MFMA kernel, bit 1: wgrad MFMA kernel, bit 2: single-channel first-layer MFMA kernels
    return (conv3d_fwd_mfma_ok(C0, C1, Cout, D, H, W, dtype) ? 1 : 0) | (conv3d_wgrad_mfma_ok(C0, C1, Cout, D, H, W, dtype) ? 2 : 0) |
           (conv3d_first_ok(C0, C1, Cout, D, H, W, dtype, 0) ? 4 : 0);
}

extern "C" int fmri_conv3d_fwd(const void* src0, int C0, int up0, const void* src1, int C1, const void* w, const float* bias,
                               const void* mask, void* y, int N, int D, int H, int W, int Cout, int act, float alpha, int dtype,
                               int impl, int planar, fmri_stream_t stream) {
    int rc = check_common(src0, C0, up0, planar, src1, C1, N, D, H, W, Cout);
    if (rc) return rc;
    if (dtype != FMRI_F32 && dtype != FMRI_BF16) return FMRI_E_DTYPE;
    if (impl != FMRI_IMPL_GENERIC && !mask && !planar && conv3d_first_ok(C0, C1, Cout, D, H, W, dtype, up0))
        return conv3d_first_fwd(src0, w, bias, y, N, D, H, W, Cout, act, alpha, as_stream(stream));
    const bool can = conv3d_fwd_mfma_ok(C0, C1, Cout, D, H, W, dtype);
    if (impl == FMRI_IMPL_MFMA && !can) return FMRI_E_SHAPE;
    if (can && impl != FMRI_IMPL_GENERIC) {
        if ((((uintptr_t)src0) | ((uintptr_t)src1) | ((uintptr_t)w) | ((uintptr_t)y) | ((uintptr_t)mask)) & 15) return FMRI_E_ALIGN;
        return conv3d_fwd_mfma(src0, C0, up0, planar, src1, C1, w, bias, mask, y, N, D, H, W, Cout, act, alpha, as_stream(stream));
    }
    return conv3d_fwd_generic(src0, C0, up0, planar, src1, C1, w, bias, mask, y, N, D, H, W, Cout, act, alpha, dtype, as_stream(stream));
}

extern "C" int fmri_conv3d_dgrad(const void* dy, int Cout, const void* w_dgrad, const void* mask, void* dx, int N, int D, int H,
                                 int W, int Cin, int dtype, int impl, int planar, fmri_stream_t stream) {
    // input gradient = the same direct convolution over dy with the tap-flipped transposed filters
    return fmri_conv3d_fwd(dy, Cout, 0, nullptr, 0, w_dgrad, nullptr, mask, dx, N, D, H, W, Cin, FMRI_ACT_NONE, 0.f, dtype, impl,
                           planar, stream);
}

extern "C" int fmri_conv3d_wgrad(const void* src0, int C0, int up0, const void* src1, int C1, const void* dy, float* dw, float* db,
                                 int N, int D, int H, int W, int Cout, int dtype, int impl, int planar, void* workspace,
                                 int64_t workspace_bytes, fmri_stream_t stream) {
    int rc = check_common(src0, C0, up0, planar, src1, C1, N, D, H, W, Cout);
    if (rc) return rc;
    if (dtype != FMRI_F32 && dtype != FMRI_BF16) return FMRI_E_DTYPE;
    if (!dy || !dw) return FMRI_E_SHAPE;
    if (impl != FMRI_IMPL_GENERIC && !planar && conv3d_first_ok(C0, C1, Cout, D, H, W, dtype, up0))
        return conv3d_first_wgrad(src0, dy, dw, db, N, D, H, W, Cout, as_stream(stream));
    const bool can = conv3d_wgrad_mfma_ok(C0, C1, Cout, D, H, W, dtype);
    if (impl == FMRI_IMPL_MFMA && !can) return FMRI_E_SHAPE;
    if (can && impl != FMRI_IMPL_GENERIC) {
        if ((((uintptr_t)src0) | ((uintptr_t)src1) | ((uintptr_t)dy)) & 15) return FMRI_E_ALIGN;
        return conv3d_wgrad_mfma(src0, C0, up0, planar, src1, C1, dy, dw, db, N, D, H, W, Cout, workspace, workspace_bytes, as_stream(stream));
    }
    return conv3d_wgrad_generic(src0, C0, up0, planar, src1, C1, dy, dw, db, N, D, H, W, Cout, dtype, as_stream(stream));
}

extern "C" int64_t fmri_conv3d_wgrad_workspace_bytes(int C0, int C1, int Cout, int N, int D, int H, int W, int dtype, int planar) {
    if (!conv3d_wgrad_mfma_ok(C0, C1, Cout, D, H, W, dtype)) return 0;
    if (!planar && conv3d_first_ok(C0, C1, Cout, D, H, W, dtype, 0)) return 0;
    return conv3d_wgrad_mfma_ws_bytes(C0, C1, Cout, N, D, H, W, planar);
}
