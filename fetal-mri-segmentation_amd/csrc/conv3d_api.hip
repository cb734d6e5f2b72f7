// C-ABI dispatch for the 3x3x3 convolution family: picks the bf16 MFMA kernels when the shape allows, else the generic path.
#include "common.h"

int conv3d_fwd_generic(const void*, int, int, int, const void*, int, const void*, const float*, const void*, void*, int, int, int, int, int,
                       int, float, int, hipStream_t);
int conv3d_wgrad_generic(const void*, int, int, int, const void*, int, const void*, float*, float*, int, int, int, int, int, int,
                         hipStream_t);
bool conv3d_fwd_mfma_ok(int C0, int C1, int Cout, int D, int H, int W, int dtype);
bool conv3d_wgrad_mfma_ok(int C0, int C1, int Cout, int D, int H, int W, int dtype);
bool conv3d_fwd_needs_cube(int D, int H, int W);
bool conv3d_wgrad_cout32_ok(int C0, int C1, int Cout, int N, int D, int H, int W, int dtype, int planar, int up0);
int conv3d_fwd_mfma(const void*, int, int, int, const void*, int, const void*, const float*, const void*, void*, int, int, int, int, int, int,
                    float, int, hipStream_t);
int conv3d_wgrad_mfma_f32(const void*, int, int, const void*, int, const void*, float*, float*, int, int, int, int, int, hipStream_t);
int conv3d_wgrad_mfma(const void*, int, int, int, const void*, int, const void*, float*, float*, int, int, int, int, int, void*, int64_t,
                      hipStream_t);
int64_t conv3d_wgrad_mfma_ws_bytes(int C0, int C1, int Cout, int N, int D, int H, int W, int planar);
int conv3d_upcat_wgrad_mfma(const void*, int, const void*, int, const void*, float*, float*, float*, int, int, int, int, int, int, void*, int64_t,
                            hipStream_t);
int conv3d_fwd_mfma_ex(int, const void*, int, int, int, const void*, int, const void*, const float*, const void*, const void*, void*, int, int, int,
                       int, int, int, float, int, hipStream_t);

int conv3d_upcat_wgrad_mfma_ex(const void*, int, const void*, int, const void*, float*, float*, float*, int, int, int, int, int, int, int, void*,
                               int64_t, hipStream_t);
int conv3d_fwd_mfma_res_b27(const void*, int, const void*, const float*, const void*, void*, int, int, int, int, int, int, float, int, hipStream_t);
int conv3d_fwd_tail_ok(int C0, int Cout, int N, int D, int H, int W, int dtype, int planar);
int conv3d_fwd_mfma_tail(const void*, int, const void*, const float*, void*, void*, const float*, const float*, float*, int, int, int, int, int, int,
                         float, int, int, hipStream_t);

int conv3d_fwd_ntail_ok(int C0, int C1, int Cout, int N, int D, int H, int W, int dtype);
int conv3d_fwd_mfma_ntail(int, const void*, int, int, const void*, int, const void*, const float*, const void*, void*, int, int, int, int, int, int,
                          float, double*, int, const float*, hipStream_t);
int norm_ws_zero(double* ws, int n, hipStream_t s);
int norm_ws_fold(double* ws, int nslot, int n, hipStream_t s);
int conv3d_fwd_ntail_slots();

bool conv3d_first_ok(int C0, int C1, int Cout, int D, int H, int W, int dtype, int up0, int planar);
int conv3d_first_fwd(const void*, int, int, const void*, const float*, void*, int, int, int, int, int, int, float, hipStream_t);
int conv3d_first_wgrad(const void*, int, int, const void*, float*, float*, int, int, int, int, int, hipStream_t);

static int check_common(const void* src0, int C0, int up0, int planar, const void* src1, int C1, int N, int D, int H, int W, int Cout) {
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cout <= 0 || C0 <= 0 || C1 < 0) return FMRI_E_SHAPE;
    if (!src0 || (C1 > 0 && !src1)) return FMRI_E_SHAPE;
    if (up0 && (((planar ? 0 : D) | H | W) & 1)) return FMRI_E_SHAPE;
    return FMRI_OK;
}

extern "C" int fmri_conv3d_uses_mfma(int C0, int C1, int Cout, int D, int H, int W, int dtype) {
    // bit 0: forward/dgrad MFMA kernel, bit 1: wgrad MFMA kernel, bit 2: single-channel first-layer MFMA kernels
    return (conv3d_fwd_mfma_ok(C0, C1, Cout, D, H, W, dtype) ? 1 : 0) | (conv3d_wgrad_mfma_ok(C0, C1, Cout, D, H, W, dtype) ? 2 : 0) |
           (conv3d_first_ok(C0, C1, Cout, D, H, W, dtype, 0, 0) ? 4 : 0);
}

extern "C" int fmri_conv3d_fwd(const void* src0, int C0, int up0, const void* src1, int C1, const void* w, const float* bias,
                               const void* mask, void* y, int N, int D, int H, int W, int Cout, int act, float alpha, int dtype,
                               int impl, int planar, fmri_stream_t stream) {
    int rc = check_common(src0, C0, up0, planar, src1, C1, N, D, H, W, Cout);
    if (rc) return rc;
    if (dtype != FMRI_F32 && dtype != FMRI_BF16) return FMRI_E_DTYPE;
    if (impl != FMRI_IMPL_GENERIC && !mask && conv3d_first_ok(C0, C1, Cout, D, H, W, dtype, up0, planar))
        return conv3d_first_fwd(src0, C0, planar, w, bias, y, N, D, H, W, Cout, act, alpha, as_stream(stream));
    const bool can = conv3d_fwd_mfma_ok(C0, C1, Cout, D, H, W, dtype) && !(planar && (dtype == FMRI_F32 || conv3d_fwd_needs_cube(D, H, W)));    // (the fp32 MFMA form covers 3-D launches)
    if (impl == FMRI_IMPL_MFMA && !can) return FMRI_E_SHAPE;
    if (can && impl != FMRI_IMPL_GENERIC) {
        if ((((uintptr_t)src0) | ((uintptr_t)src1) | ((uintptr_t)w) | ((uintptr_t)y) | ((uintptr_t)mask)) & 15) return FMRI_E_ALIGN;
        return conv3d_fwd_mfma(src0, C0, up0, planar, src1, C1, w, bias, mask, y, N, D, H, W, Cout, act, alpha, dtype, as_stream(stream));
    }
    return conv3d_fwd_generic(src0, C0, up0, planar, src1, C1, w, bias, mask, y, N, D, H, W, Cout, act, alpha, dtype, as_stream(stream));
}

extern "C" int fmri_conv3d_fwd_tail_ok(int C0, int Cout, int N, int D, int H, int W, int dtype) {
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || C0 <= 0 || Cout <= 0) return 0;
    return conv3d_fwd_tail_ok(C0, Cout, N, D, H, W, dtype, 0);
}

extern "C" int fmri_conv3d_fwd_tail(const void* src0, int C0, const void* w, const float* bias, void* y, void* y_pool, const float* w1,
                                    const float* b1, float* logits, int N, int D, int H, int W, int Cout, int act, float alpha, int dtype,
                                    fmri_stream_t stream) {
    int rc = check_common(src0, C0, 0, 0, nullptr, 0, N, D, H, W, Cout);
    if (rc) return rc;
    if (dtype != FMRI_BF16 && dtype != FMRI_F32) return FMRI_E_DTYPE;
    if (!w || !y || (!y_pool && !logits)) return FMRI_E_SHAPE;
    if ((((uintptr_t)src0) | ((uintptr_t)w) | ((uintptr_t)y) | ((uintptr_t)y_pool) | ((uintptr_t)w1)) & 15) return FMRI_E_ALIGN;
    return conv3d_fwd_mfma_tail(src0, C0, w, bias, y, y_pool, w1, b1, logits, N, D, H, W, Cout, act, alpha, dtype, 0, as_stream(stream));
}

// the same for the 2-D models: D slices of H x W (planar launches), y_pool = MaxPooling2D(2) per slice
extern "C" int fmri_conv3d_fwd_tail_planar_ok(int C0, int Cout, int N, int D, int H, int W, int dtype) {
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || C0 <= 0 || Cout <= 0) return 0;
    return conv3d_fwd_tail_ok(C0, Cout, N, D, H, W, dtype, 1);
}
extern "C" int fmri_conv3d_fwd_tail_planar(const void* src0, int C0, const void* w, const float* bias, void* y, void* y_pool, const float* w1,
                                           const float* b1, float* logits, int N, int D, int H, int W, int Cout, int act, float alpha,
                                           int dtype, fmri_stream_t stream) {
    int rc = check_common(src0, C0, 0, 1, nullptr, 0, N, D, H, W, Cout);
    if (rc) return rc;
    if (dtype != FMRI_BF16) return FMRI_E_DTYPE;
    if (!w || !y || (!y_pool && !logits)) return FMRI_E_SHAPE;
    if ((((uintptr_t)src0) | ((uintptr_t)w) | ((uintptr_t)y) | ((uintptr_t)y_pool) | ((uintptr_t)w1)) & 15) return FMRI_E_ALIGN;
    return conv3d_fwd_mfma_tail(src0, C0, w, bias, y, y_pool, w1, b1, logits, N, D, H, W, Cout, act, alpha, dtype, 1, as_stream(stream));
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
    if (impl != FMRI_IMPL_GENERIC && conv3d_first_ok(C0, C1, Cout, D, H, W, dtype, up0, planar))
        return conv3d_first_wgrad(src0, C0, planar, dy, dw, db, N, D, H, W, Cout, as_stream(stream));
    bool can = conv3d_wgrad_mfma_ok(C0, C1, Cout, D, H, W, dtype) && !(planar && dtype == FMRI_F32);       // (the fp32 MFMA form covers 3-D launches)
    if (!can && (Cout % 64) && conv3d_wgrad_cout32_ok(C0, C1, Cout, N, D, H, W, dtype, planar, up0)) {
        can = true;                       // 32-wide Cout on the kd-sharing kernel (32 x 32 blocks): without a workspace, so that it takes the launch
        workspace = nullptr;
        workspace_bytes = 0;
    }
    if (impl == FMRI_IMPL_MFMA && !can) return FMRI_E_SHAPE;
    if (can && impl != FMRI_IMPL_GENERIC) {
        if ((((uintptr_t)src0) | ((uintptr_t)src1) | ((uintptr_t)dy)) & 15) return FMRI_E_ALIGN;
        if (dtype == FMRI_F32) return conv3d_wgrad_mfma_f32(src0, C0, up0, src1, C1, dy, dw, db, N, D, H, W, Cout, as_stream(stream));
        return conv3d_wgrad_mfma(src0, C0, up0, planar, src1, C1, dy, dw, db, N, D, H, W, Cout, workspace, workspace_bytes, as_stream(stream));
    }
    return conv3d_wgrad_generic(src0, C0, up0, planar, src1, C1, dy, dw, db, N, D, H, W, Cout, dtype, as_stream(stream));
}

extern "C" int fmri_conv3d_wgrad_cout32_ok(int C0, int C1, int Cout, int N, int D, int H, int W, int dtype, int planar, int up0) {
    return conv3d_wgrad_cout32_ok(C0, C1, Cout, N, D, H, W, dtype, planar, up0) ? 1 : 0;
}
extern "C" int64_t fmri_conv3d_wgrad_workspace_bytes(int C0, int C1, int Cout, int N, int D, int H, int W, int dtype, int planar) {
    if (dtype != FMRI_BF16 || !conv3d_wgrad_mfma_ok(C0, C1, Cout, D, H, W, dtype)) return 0;      // (the fp32 form has no slab flush)
    if (conv3d_first_ok(C0, C1, Cout, D, H, W, dtype, 0, planar)) return 0;
    return conv3d_wgrad_mfma_ws_bytes(C0, C1, Cout, N, D, H, W, planar);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Convolution over [nearest_up2(src0) | src1] without the redundant taps (reference unet.py:132-138 UpSampling3D -> :61 concatenate ->
// :102 Conv3D).  Output voxel 2g+p of an up-sampled source reads, per axis, the low-res voxels g-1,g (p = 0: taps {k0} and {k1+k2})
// or g,g+1 (p = 1: taps {k0+k1} and {k2}): 8 parity classes x 8 pre-summed taps instead of 27 taps on 8x the voxels = 3.4x fewer
// MACs on the up-sampled channels.  Combined weights (fp32 sums of the master weights, then one rounding):
//   w_up_fwd   [8 p][2][2][2][Cout][C0]     Wc[p][t'] = sum of w[k] over the taps k that parity p maps onto low-res offset t'
//   w_up_dgrad [8 p][2][2][2][C0][Cout]     = Wc[p][1-t']^T   (the kernel walks the mirrored taps over the space-to-depth view of dy)
//   w_skip_fwd [27][Cout][C1], w_skip_dgrad [27][C1][Cout] (tap-flipped): the skip channels keep the plain 27-tap convolution
namespace {
__device__ __forceinline__ int tap_class(int p, int k) { return p == 0 ? (k >= 1) : (k >= 2); }   // 3-tap index -> combined tap t'

// One 64 (Cout) x 64 (Cin) tile per workgroup: blocks [0, n_up) cover (class, Cout tile, C0 tile) of the pre-summed parity filters, the rest
// (tap, Cout tile, C1 tile) of the skip filters; both images of a tile (row-major forward, transposed input-gradient) are written coalesced
// (pack_tile, common.h).  The pre-sums run over (kd, kh, kw) in ascending order in fp32, then one rounding.
template <typename T>
__device__ __forceinline__ void pack_up_block(int b, const float* __restrict__ w, int C0, int C1, int Cout, int planar, T* __restrict__ up_f,
                                              T* __restrict__ up_d, T* __restrict__ sk_f, T* __restrict__ sk_d, T (*tile)[PACK_PITCH(T)]) {
    const int Cin = C0 + C1;
    const int tco = (Cout + 63) >> 6, tc0 = (C0 + 63) >> 6, tc1 = (C1 + 63) >> 6;
    const int ncls = planar ? 16 : 64;
    const int n_up = ncls * tco * tc0;
    if (b < n_up) {
        const int ci0 = (b % tc0) << 6; b /= tc0;
        const int co0 = (b % tco) << 6;
        const int q = b / tco;                                     // class = (parity, combined tap)
        int mirrored;
        if (planar) {                                              // 2-D slices: parity (ph, pw), combined taps (kh', kw'), centre kd plane only
            const int tw = q & 1, th = (q >> 1) & 1, p = q >> 2, ph = p >> 1, pw = p & 1;
            mirrored = (p * 2 + (1 - th)) * 2 + (1 - tw);
            pack_tile<T>([&](int co, int c0) {
                float acc = 0.f;
                for (int kh = 0; kh < 3; ++kh)
                    for (int kw = 0; kw < 3; ++kw)
                        if (tap_class(ph, kh) == th && tap_class(pw, kw) == tw) acc += w[((int64_t)(9 + kh * 3 + kw) * Cout + co) * Cin + c0];
                return acc;
            }, up_f ? up_f + (int64_t)q * Cout * C0 : nullptr, C0, up_d ? up_d + (int64_t)mirrored * C0 * Cout : nullptr, Cout, co0, ci0, Cout, C0, tile);
            return;
        }
        const int tw = q & 1, th = (q >> 1) & 1, td = (q >> 2) & 1, p = q >> 3;
        const int pd = p >> 2, ph = (p >> 1) & 1, pw = p & 1;
        mirrored = ((p * 2 + (1 - td)) * 2 + (1 - th)) * 2 + (1 - tw);
        pack_tile<T>([&](int co, int c0) {
            float acc = 0.f;
            for (int kd = 0; kd < 3; ++kd)
                for (int kh = 0; kh < 3; ++kh)
                    for (int kw = 0; kw < 3; ++kw)
                        if (tap_class(pd, kd) == td && tap_class(ph, kh) == th && tap_class(pw, kw) == tw)
                            acc += w[((int64_t)((kd * 3 + kh) * 3 + kw) * Cout + co) * Cin + c0];
            return acc;
        }, up_f ? up_f + (int64_t)q * Cout * C0 : nullptr, C0, up_d ? up_d + (int64_t)mirrored * C0 * Cout : nullptr, Cout, co0, ci0, Cout, C0, tile);
        return;
    }
    b -= n_up;
    const int ci0 = (b % tc1) << 6; b /= tc1;
    const int co0 = (b % tco) << 6;
    const int t = b / tco;
    const float* const wt = w + (int64_t)t * Cout * Cin + C0;
    pack_tile<T>([&](int co, int c1) { return wt[(int64_t)co * Cin + c1]; }, sk_f ? sk_f + (int64_t)t * Cout * C1 : nullptr, C1,
                 sk_d ? sk_d + (int64_t)(26 - t) * C1 * Cout : nullptr, Cout, co0, ci0, Cout, C1, tile);
}
template <typename T>
__global__ void __launch_bounds__(256) k_pack_up_weights(const float* __restrict__ w, int C0, int C1, int Cout, int planar, T* __restrict__ up_f,
                                                         T* __restrict__ up_d, T* __restrict__ sk_f, T* __restrict__ sk_d) {
    __shared__ T tile[64][PACK_PITCH(T)];
    pack_up_block<T>(blockIdx.x, w, C0, C1, Cout, planar, up_f, up_d, sk_f, sk_d, tile);
}

// Every weight image of a model in ONE launch (round 6; fmri_pack_weights_batched).  table[layer][10] int64:
//   [0] kind (0: fmri_conv3d_pack_weights, 1: fmri_conv3d / conv2d_pack_up_weights)   [1] first workgroup of the layer
//   [2] w (fp32 master)   [3..6] destinations: kind 0 {w_fwd, w_dgrad, -, -}, kind 1 {w_up_fwd, w_up_dgrad, w_skip_fwd, w_skip_dgrad}
//   [7] Cout   [8] Cin (kind 0) / C0 (kind 1)   [9] C1 | planar << 32 (kind 1)
// A workgroup finds its layer (the table is small and wave-uniform) and runs the layer's own per-launch code on its local block number:
// the images are the bits the per-layer launches write.
template <typename T>
__global__ void __launch_bounds__(256) k_pack_batched(const long long* __restrict__ table, int n_layers) {
    __shared__ T tile[64][PACK_PITCH(T)];
    int L = 0;
    for (int i = 1; i < n_layers; ++i)
        if ((long long)blockIdx.x >= table[i * 10 + 1]) L = i;
    const long long* const r = table + L * 10;
    const int b = (int)(blockIdx.x - r[1]);
    const float* const w = reinterpret_cast<const float*>(r[2]);
    if (r[0] == 0)
        pack_plain_block<T>(b, w, reinterpret_cast<T*>(r[3]), reinterpret_cast<T*>(r[4]), (int)r[7], (int)r[8], tile);
    else
        pack_up_block<T>(b, w, (int)r[8], (int)(r[9] & 0xffffffffll), (int)r[7], (int)(r[9] >> 32), reinterpret_cast<T*>(r[3]), reinterpret_cast<T*>(r[4]),
                         reinterpret_cast<T*>(r[5]), reinterpret_cast<T*>(r[6]), tile);
}

// D,H,W = output (full-resolution) dims; planar: D = number of slices (not up-sampled).  bit 0: forward + input gradients, bit 1: weight gradient
int upcat_ok(int C0, int C1, int Cout, int D, int H, int W, int dtype, int planar) {
    if (((planar ? 0 : D) | H | W) & 1) return 0;
    if (planar && dtype == FMRI_F32) return 0;           // fp32 on the MFMA kernels: 3-D launches only
    const int Dl = planar ? D : D / 2;
    if (C0 <= 0 || C1 < 0 || Cout * (planar ? 4 : 8) * (dtype == FMRI_F32 ? 2 : 1) > 4096) return 0;       // up-backward sends (classes x Cout) channels of dy through the zero page
    // C1 = 0: a convolution of a purely up-sampled tensor (reference isensee2017.py:101-104 create_up_sampling_module)
    if (conv3d_fwd_needs_cube(Dl, H / 2, W / 2) || conv3d_fwd_needs_cube(D, H, W)) return 0;      // the parity launches use the 4x8x16 tiling
    const bool fb = conv3d_fwd_mfma_ok(C0, 0, Cout, Dl, H / 2, W / 2, dtype) && conv3d_fwd_mfma_ok(Cout, 0, C0, Dl, H / 2, W / 2, dtype) &&
                    (C1 == 0 || (conv3d_fwd_mfma_ok(C1, 0, Cout, D, H, W, dtype) && conv3d_fwd_mfma_ok(Cout, 0, C1, D, H, W, dtype)));
    // (fp32: the parity-form weight gradient is not built - the 27-tap kernel with the fused up-sampled source takes the launch)
    const bool wg_ = dtype == FMRI_BF16 && conv3d_wgrad_mfma_ok(C0, 0, Cout, Dl, H / 2, W / 2, dtype) && (C1 == 0 || conv3d_wgrad_mfma_ok(C1, 0, Cout, D, H, W, dtype));
    return (fb ? 1 : 0) | (fb && wg_ ? 2 : 0);
}

}  // namespace
extern "C" int fmri_pack_weights_batched(const int64_t* table, int n_layers, int n_blocks, int dtype, fmri_stream_t stream) {
    if (!table || n_layers <= 0 || n_blocks <= 0) return FMRI_E_SHAPE;
    if (dtype == FMRI_BF16) k_pack_batched<bf16_t><<<n_blocks, 256, 0, as_stream(stream)>>>(reinterpret_cast<const long long*>(table), n_layers);
    else if (dtype == FMRI_F32) k_pack_batched<float><<<n_blocks, 256, 0, as_stream(stream)>>>(reinterpret_cast<const long long*>(table), n_layers);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
namespace {
int upcat_pack(const float* w, int C0, int C1, int Cout, void* w_up_fwd, void* w_up_dgrad, void* w_skip_fwd, void* w_skip_dgrad, int dtype,
               int planar, fmri_stream_t stream) {
    if (!w || C0 <= 0 || C1 < 0 || Cout <= 0) return FMRI_E_SHAPE;
    const int tco = (Cout + 63) / 64;
    const int grid = (planar ? 16 : 64) * tco * ((C0 + 63) / 64) + 27 * tco * ((C1 + 63) / 64);
    if (dtype == FMRI_BF16)
        k_pack_up_weights<bf16_t><<<grid, 256, 0, as_stream(stream)>>>(w, C0, C1, Cout, planar, (bf16_t*)w_up_fwd, (bf16_t*)w_up_dgrad,
                                                                        (bf16_t*)w_skip_fwd, (bf16_t*)w_skip_dgrad);
    else if (dtype == FMRI_F32)
        k_pack_up_weights<float><<<grid, 256, 0, as_stream(stream)>>>(w, C0, C1, Cout, planar, (float*)w_up_fwd, (float*)w_up_dgrad,
                                                                       (float*)w_skip_fwd, (float*)w_skip_dgrad);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

int upcat_fwd(const void* src0_low, int C0, const void* src1, int C1, const void* w_up_fwd, const void* w_skip_fwd, const float* bias, void* y,
              int N, int D, int H, int W, int Cout, int act, float alpha, int dtype, int planar, fmri_stream_t stream) {
    if (!src0_low || !w_up_fwd || !y || N <= 0 || (C1 > 0 && (!src1 || !w_skip_fwd))) return FMRI_E_SHAPE;
    if (!(upcat_ok(C0, C1, Cout, D, H, W, dtype, planar) & 1)) return FMRI_E_SHAPE;
    if ((((uintptr_t)src0_low) | ((uintptr_t)src1) | ((uintptr_t)w_up_fwd) | ((uintptr_t)w_skip_fwd) | ((uintptr_t)y)) & 15) return FMRI_E_ALIGN;
    const int Dl = planar ? D : D / 2;
    if (C1 == 0)        // nothing to add: the parity launch finishes the output itself
        return conv3d_fwd_mfma_ex(1, src0_low, C0, 0, planar, nullptr, 0, w_up_fwd, bias, nullptr, nullptr, y, N, Dl, H / 2, W / 2, Cout, act, alpha,
                                  dtype, as_stream(stream));
    // 1. partial sums of the up-sampled channels, scattered by parity class into y
    int rc = conv3d_fwd_mfma_ex(1, src0_low, C0, 0, planar, nullptr, 0, w_up_fwd, nullptr, nullptr, nullptr, y, N, Dl, H / 2, W / 2, Cout,
                                FMRI_ACT_NONE, 0.f, dtype, as_stream(stream));
    if (rc) return rc;
    // 2. plain conv over the skip channels; its epilogue adds the partial sums (in place), the bias, and applies the activation
    return conv3d_fwd_mfma_ex(0, src1, C1, 0, planar, nullptr, 0, w_skip_fwd, bias, nullptr, y, y, N, D, H, W, Cout, act, alpha, dtype, as_stream(stream));
}

int upcat_dgrad(const void* dy, int Cout, const void* w_up_dgrad, const void* w_skip_dgrad, const void* mask_low, const void* mask_skip,
                void* dx_low, void* dx_skip, int N, int D, int H, int W, int C0, int C1, int dtype, int planar, fmri_stream_t stream) {
    if (!dy || !w_up_dgrad || !dx_low || N <= 0 || (C1 > 0 && (!w_skip_dgrad || !dx_skip))) return FMRI_E_SHAPE;
    if (!(upcat_ok(C0, C1, Cout, D, H, W, dtype, planar) & 1)) return FMRI_E_SHAPE;
    if ((((uintptr_t)dy) | ((uintptr_t)w_up_dgrad) | ((uintptr_t)w_skip_dgrad) | ((uintptr_t)dx_low) | ((uintptr_t)dx_skip) | ((uintptr_t)mask_low) |
         ((uintptr_t)mask_skip)) & 15)
        return FMRI_E_ALIGN;
    // gradient of the low-res tensor: one launch over the space-to-depth view of dy (parity classes x Cout channels, mirrored taps)
    int rc = conv3d_fwd_mfma_ex(2, dy, Cout, 0, planar, nullptr, 0, w_up_dgrad, nullptr, mask_low, nullptr, dx_low, N, planar ? D : D / 2, H / 2,
                                W / 2, C0, FMRI_ACT_NONE, 0.f, dtype, as_stream(stream));
    if (rc || C1 == 0) return rc;
    // gradient of the skip tensor: the plain tap-flipped transposed convolution restricted to the skip rows
    return conv3d_fwd_mfma_ex(0, dy, Cout, 0, planar, nullptr, 0, w_skip_dgrad, nullptr, mask_skip, nullptr, dx_skip, N, D, H, W, C1, FMRI_ACT_NONE,
                              0.f, dtype, as_stream(stream));
}

int upcat_wgrad(const void* src0_low, int C0, const void* src1, int C1, const void* dy, float* dw, float* db, float* dwc_scratch, int N, int D,
                int H, int W, int Cout, int dtype, int planar, void* workspace, int64_t workspace_bytes, fmri_stream_t stream) {
    if (!src0_low || !dy || !dw || !dwc_scratch || N <= 0 || (C1 > 0 && !src1)) return FMRI_E_SHAPE;
    if (!(upcat_ok(C0, C1, Cout, D, H, W, dtype, planar) & 2)) return FMRI_E_SHAPE;
    if ((((uintptr_t)src0_low) | ((uintptr_t)src1) | ((uintptr_t)dy)) & 15) return FMRI_E_ALIGN;
    return conv3d_upcat_wgrad_mfma(src0_low, C0, src1, C1, dy, dw, db, dwc_scratch, N, D, H, W, Cout, planar, workspace, workspace_bytes,
                                   as_stream(stream));
}
}  // namespace

// ---- Deconvolution3D(k 2, s 2) -> concatenate -> Conv3D folded into ONE parity-form convolution of the low-res tensor (round 3).
// Output voxel 2g+p of the transposed conv is Wt[p] x[g] + bt; tap k of the following 3x3x3 conv at output voxel 2g+p therefore reads
// low-res voxel g + floor((p+k-1)/2) through Wt[(p+k-1) mod 2] - the same two low-res neighbours per axis the nearest-upsample parity
// form reads, with pre-MULTIPLIED filters W3[k] Wt[a] summed per (parity, neighbour) instead of pre-summed ones (the caller forms them).
// The transposed conv's bias reaches an output voxel through the in-volume taps only: bias27 holds the effective bias per border class.
namespace {
// sums of dy per border class of the voxel: out [27][C] fp32 (zeroed by the caller), class = (cd*3 + ch)*3 + cw, c = 0 / 1 / 2 as in
// FwdTail::bias27.  One workgroup per (n, d) plane, one thread per channel (channels beyond the block size in further rounds): the plane's
// nine (ch, cw) classes are accumulated in registers - rows off the h faces contribute their two end voxels only, the interior class
// (1, 1, 1) is never read: the caller gets it as (sum over all voxels) - (the 26 border classes) - and flushed with ONE atomic per class and
// channel (the first version, a workgroup per row, spent 0.35 ms per launch on 4 M atomics into 27 x C addresses).
template <typename T>
__global__ void k_border_class_sums(const T* __restrict__ dy, float* __restrict__ out, int D, int H, int W, int C) {
    const int plane = blockIdx.x;                     // n * D + d
    const int d = plane % D;
    const int cd = d == 0 ? 0 : (d == D - 1 ? 2 : 1);
    const T* const base = dy + (int64_t)plane * H * W * C;
    // 256 threads = G row groups x C channels (C <= 256 and a power of two), or one group striding over the channels; blockIdx.y cuts the
    // rows further (the two d-face planes of a sample read every voxel)
    const int G = (C <= 256 && 256 % C == 0) ? 256 / C : 1;
    const int grp = G > 1 ? threadIdx.x / C : 0;
    for (int c = G > 1 ? threadIdx.x % C : threadIdx.x; c < C; c += (G > 1 ? C : blockDim.x)) {
        float acc[3][3] = {};
        for (int h = blockIdx.y * G + grp; h < H; h += gridDim.y * G) {
            const int ch = h == 0 ? 0 : (h == H - 1 ? 2 : 1);
            const T* const row = base + (int64_t)h * W * C + c;
            acc[ch][0] += to_f<T>(row[0]);
            acc[ch][2] += to_f<T>(row[(int64_t)(W - 1) * C]);
            if (cd != 1 || ch != 1) {
                float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
                int w = 1;
                for (; w + 3 < W - 1; w += 4) {
                    m0 += to_f<T>(row[(int64_t)w * C]);
                    m1 += to_f<T>(row[(int64_t)(w + 1) * C]);
                    m2 += to_f<T>(row[(int64_t)(w + 2) * C]);
                    m3 += to_f<T>(row[(int64_t)(w + 3) * C]);
                }
                for (; w < W - 1; ++w) m0 += to_f<T>(row[(int64_t)w * C]);
                acc[ch][1] += (m0 + m1) + (m2 + m3);
            }
        }
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
#pragma unroll
            for (int cw = 0; cw < 3; ++cw)
                if (!(cd == 1 && ch == 1 && cw == 1) && acc[ch][cw] != 0.f) atomicAdd(&out[((cd * 3 + ch) * 3 + cw) * C + c], acc[ch][cw]);
    }
}
}  // namespace
extern "C" int fmri_conv3d_upcat_fwd_bias27(const void* src0_low, int C0, const void* src1, int C1, const void* w_up_fwd, const void* w_skip_fwd,
                                            const float* bias27, void* y, int N, int D, int H, int W, int Cout, int act, float alpha, int dtype,
                                            fmri_stream_t stream) {
    if (!src0_low || !src1 || !w_up_fwd || !w_skip_fwd || !bias27 || !y || N <= 0 || C1 <= 0) return FMRI_E_SHAPE;
    if (!(upcat_ok(C0, C1, Cout, D, H, W, dtype, 0) & 1)) return FMRI_E_SHAPE;
    if ((((uintptr_t)src0_low) | ((uintptr_t)src1) | ((uintptr_t)w_up_fwd) | ((uintptr_t)w_skip_fwd) | ((uintptr_t)y) | ((uintptr_t)bias27)) & 15) return FMRI_E_ALIGN;
    int rc = conv3d_fwd_mfma_ex(1, src0_low, C0, 0, 0, nullptr, 0, w_up_fwd, nullptr, nullptr, nullptr, y, N, D / 2, H / 2, W / 2, Cout, FMRI_ACT_NONE, 0.f,
                                dtype, as_stream(stream));
    if (rc) return rc;
    return conv3d_fwd_mfma_res_b27(src1, C1, w_skip_fwd, bias27, y, y, N, D, H, W, Cout, act, alpha, dtype, as_stream(stream));
}
extern "C" int fmri_conv3d_upcat_wgrad_parts(const void* src0_low, int C0, const void* src1, int C1, const void* dy, float* dw, float* db,
                                             float* dwc, int N, int D, int H, int W, int Cout, int dtype, void* workspace, int64_t workspace_bytes,
                                             fmri_stream_t stream) {
    if (!src0_low || !dy || !dw || !dwc || N <= 0 || C1 <= 0 || !src1) return FMRI_E_SHAPE;
    if (!(upcat_ok(C0, C1, Cout, D, H, W, dtype, 0) & 2)) return FMRI_E_SHAPE;
    if ((((uintptr_t)src0_low) | ((uintptr_t)src1) | ((uintptr_t)dy)) & 15) return FMRI_E_ALIGN;
    return conv3d_upcat_wgrad_mfma_ex(src0_low, C0, src1, C1, dy, dw, db, dwc, N, D, H, W, Cout, 0, 0, workspace, workspace_bytes, as_stream(stream));
}
extern "C" int fmri_border_class_sums(const void* dy, float* out27, int N, int D, int H, int W, int C, int dtype, fmri_stream_t stream) {
    if (!dy || !out27 || N <= 0 || D < 2 || H < 2 || W < 2 || C <= 0) return FMRI_E_SHAPE;
    hipStream_t s = as_stream(stream);
    const dim3 grid(N * D, H >= 64 ? 8 : (H >= 16 ? 4 : 1));
    if (dtype == FMRI_BF16) k_border_class_sums<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)dy, out27, D, H, W, C);
    else if (dtype == FMRI_F32) k_border_class_sums<float><<<grid, 256, 0, s>>>((const float*)dy, out27, D, H, W, C);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

// ---- normalisation tails (round 3): the conv in front of a BatchNormalization / InstanceNormalization layer (reference
// create_convolution_block, unet.py:103-115; isensee2017.py:12) sums its own output for the layer's statistics, and the input-gradient launch
// behind a normalised block applies the activation's derivative and forms the two reductions of the normalisation's backward pass - in the
// asynchronous epilogue of the warp-specialised kernel (k_conv_fwd_ws EPI 4 / 6 / 5), i.e. only for launches with more (tile, channel block)
// pairs than CUs: fmri_conv3d_fwd_ntail_ok() tells, per shape, whether these entry points take the launch (the caller otherwise keeps
// fmri_conv3d_fwd + fmri_norm_act_fwd / fmri_norm_act_bwd_x, which give the same results up to summation order).
// doubles of `ws` the three entry points below need for G groups of C channels: one [G][C][2] block for the totals (what
// fmri_norm_act_fwd_pre / fmri_norm_act_bwd_pre read) + one per workgroup of the persistent launch for its partial sums
extern "C" int64_t fmri_norm_tail_ws_doubles(int G, int C) { return (int64_t)(1 + conv3d_fwd_ntail_slots()) * G * C * 2; }
extern "C" int fmri_conv3d_fwd_ntail_ok(int C0, int C1, int Cout, int N, int D, int H, int W, int dtype) {
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || C0 <= 0 || C1 < 0 || Cout <= 0) return 0;
    return conv3d_fwd_ntail_ok(C0, C1, Cout, N, D, H, W, dtype);
}
// y = act(conv(src0 | src1) + bias) and ws[g][Cout][2] (double; zero on entry like every fmri_norm_* scratch, see the header) = {sum y, sum y^2} over the voxels of group g (g = sample when
// per_instance, else 0), summed over the bf16 values as stored: what fmri_norm_act_fwd's own reduction pass would read
extern "C" int fmri_conv3d_fwd_stats(const void* src0, int C0, int up0, const void* src1, int C1, const void* w, const float* bias, void* y,
                                     int N, int D, int H, int W, int Cout, int act, float alpha, double* ws, int per_instance, int dtype,
                                     fmri_stream_t stream) {
    int rc = check_common(src0, C0, up0, 0, src1, C1, N, D, H, W, Cout);
    if (rc) return rc;
    if (!ws || !w || !y || !conv3d_fwd_ntail_ok(C0, C1, Cout, N, D, H, W, dtype)) return FMRI_E_SHAPE;
    if ((((uintptr_t)src0) | ((uintptr_t)src1) | ((uintptr_t)w) | ((uintptr_t)y) | ((uintptr_t)ws)) & 15) return FMRI_E_ALIGN;
    const int nent = (per_instance ? N : 1) * Cout * 2, nslot = conv3d_fwd_ntail_slots();
    rc = conv3d_fwd_mfma_ntail(0, src0, C0, up0, src1, C1, w, bias, nullptr, y, N, D, H, W, Cout, act, alpha, ws, per_instance, nullptr,
                               as_stream(stream));
    if (rc) return rc;
    return norm_ws_fold(ws, nslot, nent, as_stream(stream));
}
// the same for UpSampling3D -> concatenate -> Conv3D in parity form (fmri_conv3d_upcat_fwd): the skip launch, which finishes the output, sums it
extern "C" int fmri_conv3d_upcat_fwd_stats(const void* src0_low, int C0, const void* src1, int C1, const void* w_up_fwd, const void* w_skip_fwd,
                                           const float* bias, void* y, int N, int D, int H, int W, int Cout, int act, float alpha, double* ws,
                                           int per_instance, int dtype, fmri_stream_t stream) {
    if (!src0_low || !src1 || !w_up_fwd || !w_skip_fwd || !y || !ws || N <= 0 || C1 <= 0) return FMRI_E_SHAPE;
    if (!(upcat_ok(C0, C1, Cout, D, H, W, dtype, 0) & 1) || !conv3d_fwd_ntail_ok(C1, 0, Cout, N, D, H, W, dtype)) return FMRI_E_SHAPE;
    if ((((uintptr_t)src0_low) | ((uintptr_t)src1) | ((uintptr_t)w_up_fwd) | ((uintptr_t)w_skip_fwd) | ((uintptr_t)y) | ((uintptr_t)ws)) & 15) return FMRI_E_ALIGN;
    const int nent = (per_instance ? N : 1) * Cout * 2, nslot = conv3d_fwd_ntail_slots();
    int rc;
    rc = conv3d_fwd_mfma_ex(1, src0_low, C0, 0, 0, nullptr, 0, w_up_fwd, nullptr, nullptr, nullptr, y, N, D / 2, H / 2, W / 2, Cout, FMRI_ACT_NONE, 0.f,
                            dtype, as_stream(stream));
    if (rc) return rc;
    rc = conv3d_fwd_mfma_ntail(0, src1, C1, 0, nullptr, 0, w_skip_fwd, bias, y, y, N, D, H, W, Cout, act, alpha, ws, per_instance, nullptr,
                               as_stream(stream));
    if (rc) return rc;
    return norm_ws_fold(ws, nslot, nent, as_stream(stream));
}
// Input gradient of a conv whose INPUT is the output of a normalised block: dz = conv_dgrad(dy) * act'(z), z = fma(x, sc, sh) recomputed
// from that block's conv output x and nss[g][Cin][2] = {sc, sh} (fmri_norm_scale_shift), and ws[g][Cin][2] = {sum dz, sum dz * x}
// over the stored bf16 values of dz - fmri_norm_act_bwd_pre() finishes the normalisation's backward pass from these.  act / alpha: the
// normalised block's activation.  Cout = channels of dy, Cin = channels of dz and x.
extern "C" int fmri_conv3d_dgrad_norm(const void* dy, int Cout, const void* w_dgrad, const void* x, const float* nss, void* dz, int N, int D, int H,
                                      int W, int Cin, int act, float alpha, double* ws, int per_instance, int dtype, fmri_stream_t stream) {
    int rc = check_common(dy, Cout, 0, 0, nullptr, 0, N, D, H, W, Cin);
    if (rc) return rc;
    if (!ws || !w_dgrad || !x || !nss || !dz || !conv3d_fwd_ntail_ok(Cout, 0, Cin, N, D, H, W, dtype)) return FMRI_E_SHAPE;
    if ((((uintptr_t)dy) | ((uintptr_t)w_dgrad) | ((uintptr_t)x) | ((uintptr_t)dz) | ((uintptr_t)ws) | ((uintptr_t)nss)) & 15) return FMRI_E_ALIGN;
    const int nent = (per_instance ? N : 1) * Cin * 2, nslot = conv3d_fwd_ntail_slots();
    rc = conv3d_fwd_mfma_ntail(1, dy, Cout, 0, nullptr, 0, w_dgrad, nullptr, x, dz, N, D, H, W, Cin, act, alpha, ws, per_instance, nss,
                               as_stream(stream));
    if (rc) return rc;
    return norm_ws_fold(ws, nslot, nent, as_stream(stream));
}

extern "C" int fmri_conv3d_upcat_ok(int C0, int C1, int Cout, int D, int H, int W, int dtype) { return upcat_ok(C0, C1, Cout, D, H, W, dtype, 0); }
extern "C" int fmri_conv3d_pack_up_weights(const float* w, int C0, int C1, int Cout, void* w_up_fwd, void* w_up_dgrad, void* w_skip_fwd,
                                           void* w_skip_dgrad, int dtype, fmri_stream_t stream) {
    return upcat_pack(w, C0, C1, Cout, w_up_fwd, w_up_dgrad, w_skip_fwd, w_skip_dgrad, dtype, 0, stream);
}
extern "C" int fmri_conv3d_upcat_fwd(const void* src0_low, int C0, const void* src1, int C1, const void* w_up_fwd, const void* w_skip_fwd,
                                     const float* bias, void* y, int N, int D, int H, int W, int Cout, int act, float alpha, int dtype,
                                     fmri_stream_t stream) {
    return upcat_fwd(src0_low, C0, src1, C1, w_up_fwd, w_skip_fwd, bias, y, N, D, H, W, Cout, act, alpha, dtype, 0, stream);
}
extern "C" int fmri_conv3d_upcat_dgrad(const void* dy, int Cout, const void* w_up_dgrad, const void* w_skip_dgrad, const void* mask_low,
                                       const void* mask_skip, void* dx_low, void* dx_skip, int N, int D, int H, int W, int C0, int C1, int dtype,
                                       fmri_stream_t stream) {
    return upcat_dgrad(dy, Cout, w_up_dgrad, w_skip_dgrad, mask_low, mask_skip, dx_low, dx_skip, N, D, H, W, C0, C1, dtype, 0, stream);
}
// Conv3D(3x3x3, strides 2, 'same' on even dims) forward on the gather launch above (fmri_hip/strided_parity.py: the filter's 27 taps in 27 of
// the 64 (parity, block offset) slots), with the conv's own fp32 bias in the accumulators like every other forward launch
extern "C" int fmri_conv3d_stride2_fwd(const void* x, int Cin, const void* w_s2_fwd, const float* bias, void* y, int N, int D, int H, int W, int Cout,
                                       int dtype, fmri_stream_t stream) {
    if (!x || !w_s2_fwd || !y || N <= 0) return FMRI_E_SHAPE;
    if (!(upcat_ok(Cout, 0, Cin, D, H, W, dtype, 0) & 1)) return FMRI_E_SHAPE;
    if ((((uintptr_t)x) | ((uintptr_t)w_s2_fwd) | ((uintptr_t)y) | ((uintptr_t)bias)) & 15) return FMRI_E_ALIGN;
    return conv3d_fwd_mfma_ex(2, x, Cin, 0, 0, nullptr, 0, w_s2_fwd, bias, nullptr, nullptr, y, N, D / 2, H / 2, W / 2, Cout, FMRI_ACT_NONE, 0.f,
                              dtype, as_stream(stream));
}
extern "C" int fmri_conv3d_upcat_wgrad(const void* src0_low, int C0, const void* src1, int C1, const void* dy, float* dw, float* db,
                                       float* dwc_scratch, int N, int D, int H, int W, int Cout, int dtype, void* workspace,
                                       int64_t workspace_bytes, fmri_stream_t stream) {
    return upcat_wgrad(src0_low, C0, src1, C1, dy, dw, db, dwc_scratch, N, D, H, W, Cout, dtype, 0, workspace, workspace_bytes, stream);
}

// 2-D twins (reference fetal_net/model/unet/unet.py:60-66: UpSampling2D -> concatenate -> Conv2D): S slices of H x W (output dims), tensors
// [S][H][W][C] with the slices on the kernels' D axis, 4 parity classes (ph, pw) x 4 pre-summed taps instead of 9 taps on 4x the pixels.
// Weight images: w_up_fwd [4][2][2][Cout][C0], w_up_dgrad [4][2][2][C0][Cout]; w / dw / the skip images keep the 27-tap layout whose
// centre kd plane is the 3x3 kernel (as everywhere in the planar path); dwc_scratch: 16*Cout*C0 floats.
extern "C" int fmri_conv2d_upcat_ok(int C0, int C1, int Cout, int S, int H, int W, int dtype) { return upcat_ok(C0, C1, Cout, S, H, W, dtype, 1); }
extern "C" int fmri_conv2d_pack_up_weights(const float* w, int C0, int C1, int Cout, void* w_up_fwd, void* w_up_dgrad, void* w_skip_fwd,
                                           void* w_skip_dgrad, int dtype, fmri_stream_t stream) {
    return upcat_pack(w, C0, C1, Cout, w_up_fwd, w_up_dgrad, w_skip_fwd, w_skip_dgrad, dtype, 1, stream);
}
extern "C" int fmri_conv2d_upcat_fwd(const void* src0_low, int C0, const void* src1, int C1, const void* w_up_fwd, const void* w_skip_fwd,
                                     const float* bias, void* y, int S, int H, int W, int Cout, int act, float alpha, int dtype,
                                     fmri_stream_t stream) {
    return upcat_fwd(src0_low, C0, src1, C1, w_up_fwd, w_skip_fwd, bias, y, 1, S, H, W, Cout, act, alpha, dtype, 1, stream);
}
extern "C" int fmri_conv2d_upcat_dgrad(const void* dy, int Cout, const void* w_up_dgrad, const void* w_skip_dgrad, const void* mask_low,
                                       const void* mask_skip, void* dx_low, void* dx_skip, int S, int H, int W, int C0, int C1, int dtype,
                                       fmri_stream_t stream) {
    return upcat_dgrad(dy, Cout, w_up_dgrad, w_skip_dgrad, mask_low, mask_skip, dx_low, dx_skip, 1, S, H, W, C0, C1, dtype, 1, stream);
}
extern "C" int fmri_conv2d_upcat_wgrad(const void* src0_low, int C0, const void* src1, int C1, const void* dy, float* dw, float* db,
                                       float* dwc_scratch, int S, int H, int W, int Cout, int dtype, void* workspace, int64_t workspace_bytes,
                                       fmri_stream_t stream) {
    return upcat_wgrad(src0_low, C0, src1, C1, dy, dw, db, dwc_scratch, 1, S, H, W, Cout, dtype, 1, workspace, workspace_bytes, stream);
}
