/*
 * fmri_hip.h — C ABI of the MI355X (gfx950) hot path for the fetal-MRI 3D/2D U-Net.
 *
 * The reference (GalDude33/Fetal-MRI-Segmentation) has no FFI layer: its hot path is the chain of TensorFlow ops
 * that Keras emits for fetal_net/model/unet3d/unet.py:17-138 (+ metrics.py:11-32, Adam at unet.py:85) and the numpy
 * overlap-add of fetal_net/prediction.py:118-210.  Each entry point below names the reference line whose implicit
 * TF/numpy op it replaces.  The host side (fetal-mri-segmentation_amd/fetal_net/) binds these with ctypes.
 *
 * Conventions
 *   - extern "C", int return: FMRI_OK or a negative FMRI_E_* code; no exceptions, no allocation, no host sync.
 *   - every call only ENQUEUES work on the caller's HIP stream (`stream` = hipStream_t, may be NULL = default stream).
 *   - all pointers are caller-owned DEVICE pointers unless stated otherwise.
 *   - activations are channels-last: [N][D][H][W][C] ("NDHWC").  `planar` != 0 selects the 2-D slice semantics of the
 *     reference's 2-D builders (fetal_net/model/unet/unet.py): D indexes independent slices — convolutions use only the
 *     centre kd plane of the 3x3x3 filter (a 3x3 Conv2D), pooling / up-sampling act on (H, W) only.
 *   - dtype = FMRI_F32 (parity mode, plain fp32 arithmetic) or FMRI_BF16 (bf16 storage, fp32 accumulate; MFMA path
 *     when channel counts are multiples of 32 and the spatial tile divides, generic VALU path otherwise).
 *   - 3x3x3 filters are stored [27 taps = kd*9+kh*3+kw][Cout][Cin] (Cin contiguous).  Keras' (kD,kH,kW,Cin,Cout)
 *     kernels are permuted by the host when weights are loaded.
 */
#ifndef FMRI_HIP_H
#define FMRI_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* fmri_stream_t; /* hipStream_t */

enum { FMRI_OK = 0, FMRI_E_SHAPE = -1, FMRI_E_ALIGN = -2, FMRI_E_ARCH = -3, FMRI_E_LAUNCH = -4, FMRI_E_DTYPE = -5 };
enum { FMRI_F32 = 0, FMRI_BF16 = 1, FMRI_U8 = 2 /* label volumes of fmri_affine_sample only */ };
enum { FMRI_ACT_NONE = 0, FMRI_ACT_RELU = 1, FMRI_ACT_LEAKY = 2 };
/* which implementation a conv call may use: AUTO picks MFMA when the shape allows it */
enum { FMRI_IMPL_AUTO = 0, FMRI_IMPL_GENERIC = 1, FMRI_IMPL_MFMA = 2 };

int fmri_version(void);
const char* fmri_error_string(int code);
/* Measurement aid (no reference counterpart): writes one {s_memtime, s_memrealtime} pair per XCD into out16[2 * xcd + {0, 1}]
 * (16 x uint64 of device memory; an XCD no workgroup of the launch reached keeps its old value).  Two stamps on one stream bracket a
 * region: average shader clock = d(memtime) / d(memrealtime) x 0.1 GHz per XCD - the `clock_ghz` of the bench line (the chip is
 * power-limited under this workload: the clock explains box-to-box and recipe-to-recipe differences). */
int fmri_clock_stamp(unsigned long long* out16, fmri_stream_t stream);
/* 0/1: would fmri_conv3d_fwd / _wgrad take the MFMA path for this shape and dtype? (host-side query, no GPU needed)
 * Size limit of the MFMA path: its LDS-DMA addresses a halo box / a plane with 32-bit byte offsets below 2^31 (an offset at or above
 * 2^31 is the hardware's "write zeros" encoding): forward needs 2 * (min(D, 4) + 2) * H * W * max(C0, C1) * 2 B < 2^31, the weight
 * gradient H * W * max(C0, C1, Cout) * 2 B < 2^31; larger layers fall back to the generic kernels (AUTO) or return FMRI_E_SHAPE (MFMA).
 * dtype FMRI_F32 (round 6): the fp32 instantiation of the same kernels (v_mfma_f32_32x32x2_f32: an exact fp32 fmaf chain per output) takes
 * 3-D launches on the 4x8x16 tiling whose channel counts are multiples of 16 (forward / input gradient; Cout in multiples of 32) resp. 32
 * (weight gradient); the size limits above count 4 bytes per element.  Environment FMRI_F32_MFMA=0 keeps fp32 on the generic kernels. */
int fmri_conv3d_uses_mfma(int C0, int C1, int Cout, int D, int H, int W, int dtype);

/* ---- Conv3D 3x3x3, stride 1, 'same' (+BiasAdd +activation) — reference unet3d/unet.py:102,113 (create_convolution_block)
 * The input is the channel concatenation [src0 | src1] (reference unet.py:61 `concatenate([up, skip], axis=1)`):
 *   src0: C0 channels.  up0 != 0: src0 is stored at HALF resolution [N][D/2][H/2][W/2][C0] and is read through a fused
 *         nearest-neighbour x2 (reference unet.py:138 UpSampling3D) — the up-sampled tensor / concat is never built.
 *   src1: C1 channels at full resolution, or NULL with C1 == 0.
 * w [27][Cout][C0+C1] (dtype), bias fp32 [Cout] or NULL.  act per FMRI_ACT_*, leaky slope `alpha`.
 * mask (optional, [N][D][H][W][Cout], dtype): y = mask > 0 ? y : 0  — the ReluGrad of the producing layer, used when this
 * kernel runs as Conv3DBackpropInput. */
int fmri_conv3d_fwd(const void* src0, int C0, int up0, const void* src1, int C1, const void* w, const float* bias,
                    const void* mask, void* y, int N, int D, int H, int W, int Cout, int act, float alpha, int dtype,
                    int impl, int planar, fmri_stream_t stream);

/* ---- Conv3D block with its HBM-bound consumer folded into the epilogue (MFMA path only; single full-resolution source; bf16, and fp32 for
 * the pooled copy - the fp32 form works on 32-wide blocks, which never see a voxel's whole channel range: no logits there).
 * The tile of y is still in LDS when it is stored; from those same stored values the kernel can also produce
 *   y_pool [N][D/2][H/2][W/2][Cout]  = MaxPooling3D(2) of y            (reference unet3d/unet.py:51 behind the encoder block :45-50)
 *   logits [N*D*H*W] fp32            = Conv3D(1, (1,1,1))(y) = sum_c w1[c]*y[v][c] + b1[0]   (reference unet.py:68, n_labels = 1)
 * so that neither fmri_maxpool3d_2x_fwd nor fmri_conv1x1_fwd has to read y back.  Either output may be NULL (not both).
 * fmri_conv3d_fwd_tail_ok: bit 0 = y_pool available, bit 1 = logits available for this shape (host-side query, no GPU needed). */
int fmri_conv3d_fwd_tail_ok(int C0, int Cout, int N, int D, int H, int W, int dtype);
int fmri_conv3d_fwd_tail(const void* src0, int C0, const void* w, const float* bias, void* y, void* y_pool, const float* w1,
                         const float* b1, float* logits, int N, int D, int H, int W, int Cout, int act, float alpha, int dtype,
                         fmri_stream_t stream);
/* The same for the 2-D models (planar launches: D slices of H x W, bf16): y_pool [N][D][H/2][W/2][Cout] = MaxPooling2D(2) of every slice
 * (reference unet/unet.py:67 behind the encoder block :57-63), logits = the final Conv2D(n_labels = 1, (1, 1)) (unet/unet.py:82). */
int fmri_conv3d_fwd_tail_planar_ok(int C0, int Cout, int N, int D, int H, int W, int dtype);
int fmri_conv3d_fwd_tail_planar(const void* src0, int C0, const void* w, const float* bias, void* y, void* y_pool, const float* w1,
                                const float* b1, float* logits, int N, int D, int H, int W, int Cout, int act, float alpha, int dtype,
                                fmri_stream_t stream);

/* ---- Conv3DBackpropInputV2 (autodiff of unet.py:102): dx = conv(dy, w_dgrad) * (mask > 0).
 * w_dgrad [27][Cin][Cout] is the tap-flipped, transposed copy made by fmri_conv3d_pack_weights. */
int fmri_conv3d_dgrad(const void* dy, int Cout, const void* w_dgrad, const void* mask, void* dx, int N, int D, int H,
                      int W, int Cin, int dtype, int impl, int planar, fmri_stream_t stream);

/* ---- Conv3DBackpropFilterV2 + BiasAddGrad: dw[27][Cout][C0+C1] (fp32) += sum_v x[v+tap][ci]*dy[v][co];
 * db[Cout] (fp32) += sum_v dy[v][co] (db may be NULL).  ACCUMULATES: the caller zeroes dw/db once per step.
 * Same dual-source / fused-upsample input description as fmri_conv3d_fwd.
 * workspace (optional, device, fp32 scratch of at least fmri_conv3d_wgrad_workspace_bytes(...) bytes): the MFMA path then flushes its
 * per-workgroup partial sums with plain stores and reduces them in a second launch (bit-reproducible, faster than the fp32-atomic
 * flush used when workspace == NULL).  fp32 planes on the MFMA path (channel counts in multiples of 32, 3-D) always use the atomic flush. */
int fmri_conv3d_wgrad(const void* src0, int C0, int up0, const void* src1, int C1, const void* dy, float* dw, float* db,
                      int N, int D, int H, int W, int Cout, int dtype, int impl, int planar, void* workspace,
                      int64_t workspace_bytes, fmri_stream_t stream);
/* 1 when fmri_conv3d_wgrad takes a dy whose channel count is a multiple of 32 but not of 64 on the MFMA path (the kd-sharing kernel with its
 * 32 x 32 blocks takes the launch: bf16, 3-D, no fused up-sampling, >= 0.1 TFLOP; the call then ignores any workspace) - otherwise such a
 * launch runs on the generic kernel, and callers that want MFMA speed hand over a zero-extended 64-channel copy of dy. */
int fmri_conv3d_wgrad_cout32_ok(int C0, int C1, int Cout, int N, int D, int H, int W, int dtype, int planar, int up0);
/* bytes of workspace fmri_conv3d_wgrad can use for this shape (0: the shape does not take the slab path) */
int64_t fmri_conv3d_wgrad_workspace_bytes(int C0, int C1, int Cout, int N, int D, int H, int W, int dtype, int planar);

/* ---- Deconvolution3D(filters = C, k 2, s 2) -> concatenate([up, skip]) -> Conv3D(3x3x3) - reference unet.py:132-138, :61, :102 with
 * deconvolution=True - folded into ONE parity-form convolution of the LOW-res tensor (round 3).  Output voxel 2g+p of the transposed conv
 * is Wt[p] x[g] + bt, so tap k of the following conv at output voxel 2g+p reads low-res voxel g + floor((p+k-1)/2) through Wt[(p+k-1) mod 2]:
 * the two low-res neighbours per axis of the nearest-upsample parity form, with pre-MULTIPLIED filters sum_k W3[k] Wt[a(p,k)] per (parity,
 * neighbour) in the w_up_fwd / w_up_dgrad layouts of fmri_conv3d_pack_up_weights (the caller forms them: 216 small GEMMs per layer and step).
 * The transposed conv's bias reaches an output voxel through the in-volume taps only: bias27 [27 = (cd*3+ch)*3+cw][Cout] fp32 is the
 * effective bias per border class of the OUTPUT voxel (c = 0 first voxel of the axis, 1 interior, 2 last).  D,H,W = output dims. */
int fmri_conv3d_upcat_fwd_bias27(const void* src0_low, int C0, const void* src1, int C1, const void* w_up_fwd, const void* w_skip_fwd,
                                 const float* bias27, void* y, int N, int D, int H, int W, int Cout, int act, float alpha, int dtype,
                                 fmri_stream_t stream);
/* weight gradient in parts: the 64 parity-filter gradients stay in dwc [8][8][Cout][C0] fp32 (zeroed here) for the caller to chain through
 * the transposed conv's weights; the skip columns [C0, C0+C1) of dw [27][Cout][C0+C1] and db (= sum of dy) are accumulated as usual. */
int fmri_conv3d_upcat_wgrad_parts(const void* src0_low, int C0, const void* src1, int C1, const void* dy, float* dw, float* db, float* dwc,
                                  int N, int D, int H, int W, int Cout, int dtype, void* workspace, int64_t workspace_bytes,
                                  fmri_stream_t stream);
/* out27 [27][C] fp32 (ACCUMULATED, caller zeroes) += sum of dy [N][D][H][W][C] over the voxels of each border class EXCEPT the interior
 * class 13, which is left untouched (it is the sum over all voxels minus the 26 others): the bias gradient of the folded transposed conv. */
int fmri_border_class_sums(const void* dy, float* out27, int N, int D, int H, int W, int C, int dtype, fmri_stream_t stream);

/* ---- [nearest_up2(src0) | src1] -> Conv3D(3x3x3) without the redundant taps (reference unet.py:132-138 UpSampling3D, :61
 * concatenate, :102 Conv3D).  Output voxel 2g+p of the up-sampled source only sees low-res voxels {g-1,g} (p = 0) or {g,g+1}
 * (p = 1) per axis with pre-summed weights: 8 parity classes x 8 taps on the LOW-res tensor instead of 27 taps on 8x the voxels.
 * Same result as fmri_conv3d_fwd(up0 = 1) up to one extra rounding (to the storage dtype) of the up-sampled channels' partial sum.  3-D; bf16, and
 * fp32 where fmri_conv3d_upcat_ok says so (bit 0; the fp32 weight gradient takes the 27-tap kernel: bit 1 is bf16 only).
 * fmri_conv3d_upcat_ok (D,H,W = OUTPUT dims): bit 0 = forward / input gradients supported, bit 1 = weight gradient too. */
int fmri_conv3d_upcat_ok(int C0, int C1, int Cout, int D, int H, int W, int dtype);
/* w: fp32 master [27][Cout][C0+C1] (up-sampled channels first).  Outputs (any may be NULL): w_up_fwd [8][8][Cout][C0],
 * w_up_dgrad [8][8][C0][Cout], w_skip_fwd [27][Cout][C1], w_skip_dgrad [27][C1][Cout] (tap-flipped). */
int fmri_conv3d_pack_up_weights(const float* w, int C0, int C1, int Cout, void* w_up_fwd, void* w_up_dgrad, void* w_skip_fwd,
                                void* w_skip_dgrad, int dtype, fmri_stream_t stream);
/* y [N][D][H][W][Cout] = act(conv(up2(src0_low [N][D/2][H/2][W/2][C0])) + conv(src1 [N][D][H][W][C1]) + bias) */
int fmri_conv3d_upcat_fwd(const void* src0_low, int C0, const void* src1, int C1, const void* w_up_fwd, const void* w_skip_fwd,
                          const float* bias, void* y, int N, int D, int H, int W, int Cout, int act, float alpha, int dtype,
                          fmri_stream_t stream);
/* dx_low [N][D/2][H/2][W/2][C0] and dx_skip [N][D][H][W][C1] from dy [N][D][H][W][Cout]; mask_* (optional) = the stored
 * post-ReLU tensors of the two producers (gradient zeroed where they are <= 0).  Replaces dgrad-of-concat + UpSampling3D gradient. */
int fmri_conv3d_upcat_dgrad(const void* dy, int Cout, const void* w_up_dgrad, const void* w_skip_dgrad, const void* mask_low,
                            const void* mask_skip, void* dx_low, void* dx_skip, int N, int D, int H, int W, int C0, int C1, int dtype,
                            fmri_stream_t stream);
/* Conv3D(3x3x3, strides (2,2,2), padding 'same') on even D, H, W (reference fetal_net/model/unet3d/isensee2017.py:51: the context pathway's
 * down-sampling convs): y [N][D/2][H/2][W/2][Cout] = bias + sum_t W[t] x[2o + t] (TF pads one plane BEHIND the volume) as ONE gather launch of
 * the kernel above over x [N][D][H][W][Cin].  w_s2_fwd: the 27 taps in 27 of the 64 (parity, block offset) slots of a w_up_dgrad-shaped image
 * [8][8][Cout][Cin] (fmri_hip/strided_parity.py builds it); bias fp32 [Cout] or NULL, added in the fp32 accumulators.  Shapes as
 * fmri_conv3d_upcat_ok(Cout, 0, Cin, D, H, W) bit 0.  Input gradient: fmri_conv3d_upcat_fwd with the scatter image; weight gradient:
 * fmri_conv3d_upcat_wgrad (27 of its 64 slot gradients). */
int fmri_conv3d_stride2_fwd(const void* x, int Cin, const void* w_s2_fwd, const float* bias, void* y, int N, int D, int H, int W, int Cout, int dtype,
                            fmri_stream_t stream);

/* dw [27][Cout][C0+C1] fp32 and db [Cout] (optional) ACCUMULATED, like fmri_conv3d_wgrad(up0 = 1) but with 8 instead of 27 taps of
 * work on the up-sampled channels.  dwc_scratch: 64*Cout*C0 floats of device scratch (overwritten).  workspace: as fmri_conv3d_wgrad
 * for the skip channels (query fmri_conv3d_wgrad_workspace_bytes(C1, 0, Cout, ...)). */
int fmri_conv3d_upcat_wgrad(const void* src0_low, int C0, const void* src1, int C1, const void* dy, float* dw, float* db,
                            float* dwc_scratch, int N, int D, int H, int W, int Cout, int dtype, void* workspace, int64_t workspace_bytes,
                            fmri_stream_t stream);

/* ---- 2-D twins of the parity-form family (reference fetal_net/model/unet/unet.py:60-66 UpSampling2D -> concatenate -> Conv2D, the decoder
 * of unet_model_2d).  Tensors are [S][H][W][C] (S slices, no coupling along S), dims below are OUTPUT dims; 4 parity classes (ph,pw) x 4
 * pre-summed taps on the low-res [S][H/2][W/2][C0] tensor instead of 9 taps on 4x the pixels.  w / dw / w_skip_* keep the 27-tap layout of
 * the planar path (the 3x3 kernel is the centre kd plane); w_up_fwd [4][2][2][Cout][C0], w_up_dgrad [4][2][2][C0][Cout];
 * dwc_scratch: 16*Cout*C0 floats.  Argument meaning otherwise as the fmri_conv3d_upcat_* functions. */
int fmri_conv2d_upcat_ok(int C0, int C1, int Cout, int S, int H, int W, int dtype);
int fmri_conv2d_pack_up_weights(const float* w, int C0, int C1, int Cout, void* w_up_fwd, void* w_up_dgrad, void* w_skip_fwd,
                                void* w_skip_dgrad, int dtype, fmri_stream_t stream);
int fmri_conv2d_upcat_fwd(const void* src0_low, int C0, const void* src1, int C1, const void* w_up_fwd, const void* w_skip_fwd,
                          const float* bias, void* y, int S, int H, int W, int Cout, int act, float alpha, int dtype, fmri_stream_t stream);
int fmri_conv2d_upcat_dgrad(const void* dy, int Cout, const void* w_up_dgrad, const void* w_skip_dgrad, const void* mask_low,
                            const void* mask_skip, void* dx_low, void* dx_skip, int S, int H, int W, int C0, int C1, int dtype,
                            fmri_stream_t stream);
int fmri_conv2d_upcat_wgrad(const void* src0_low, int C0, const void* src1, int C1, const void* dy, float* dw, float* db,
                            float* dwc_scratch, int S, int H, int W, int Cout, int dtype, void* workspace, int64_t workspace_bytes,
                            fmri_stream_t stream);

/* fp32 master filter [27][Cout][Cin] -> w_fwd (dtype, same layout) and w_dgrad (dtype, [26-tap][Cin][Cout]).
 * Either destination may be NULL. */
int fmri_conv3d_pack_weights(const float* w, void* w_fwd, void* w_dgrad, int Cout, int Cin, int dtype,
                             fmri_stream_t stream);
/* Every weight image of a model in ONE launch (the per-layer launches above are launch-bound: 14 of them cost the configs[1] step 0.12 ms).
 * table (device memory): n_layers records of 10 int64:
 *   [0] kind: 0 = fmri_conv3d_pack_weights, 1 = fmri_conv3d_pack_up_weights / fmri_conv2d_pack_up_weights
 *   [1] number of the layer's first workgroup (records ascending; the layer takes the workgroups its own launch would: 27 * ceil(Cout/64) *
 *       ceil(Cin/64), or (64 | planar 16) * ceil(Cout/64) * ceil(C0/64) + 27 * ceil(Cout/64) * ceil(C1/64))
 *   [2] w   [3..6] destinations - kind 0: {w_fwd, w_dgrad, 0, 0}, kind 1: {w_up_fwd, w_up_dgrad, w_skip_fwd, w_skip_dgrad} (0 = not wanted)
 *   [7] Cout   [8] Cin (kind 0) or C0 (kind 1)   [9] kind 1: C1 | planar << 32
 * n_blocks = the sum of the layers' workgroups.  The images are bit-identical to those of the per-layer calls. */
int fmri_pack_weights_batched(const int64_t* table, int n_layers, int n_blocks, int dtype, fmri_stream_t stream);

/* ---- final Conv3D(n_labels,(1,1,1)) — reference unet.py:68.  logits[v][l] (fp32) = sum_c x[v][c]*w[l][c] + b[l]. */
int fmri_conv1x1_fwd(const void* x, const float* w, const float* b, float* logits, int64_t nvox, int C, int L,
                     int dtype, fmri_stream_t stream);
/* dx[v][c] = (x[v][c] > 0 || !relu_mask) ? sum_l dlogits[v][l]*w[l][c] : 0 ; dw[l][c] += sum_v dlogits*x ; db[l] += sum_v dlogits */
int fmri_conv1x1_bwd(const void* x, const float* w, const float* dlogits, void* dx, float* dw, float* db, int64_t nvox,
                     int C, int L, int relu_mask, int dtype, fmri_stream_t stream);

/* ---- Activation('sigmoid') + dice_coefficient_loss + the compiled metrics — reference unet.py:69,81-85, metrics.py:11-32.
 * probs = sigmoid(logits); sums (16 doubles, ACCUMULATED, caller zeroes) =
 *   [0] sum y*p  [1] sum y  [2] sum p  [3] sum (y>.5)(p>.5)  [4] sum (y>.5)  [5] sum (p>.5)  [6] sum (round(p)==y)  [7] n
 *   [8] sum binary cross-entropy (p clipped to [1e-7, 1-1e-7] as K.binary_crossentropy)  [9] sum focal term (alpha .5, gamma 2)
 * y_true is uint8 (reference generator emits uint8 truth) with the same [v][l] indexing as logits. */
int fmri_sigmoid_dice_fwd(const float* logits, const uint8_t* y_true, float* probs, double* sums, int64_t n,
                          fmri_stream_t stream);
/* dlogits = grad_scale * dL/dp * p(1-p),  L = -(2I+s)/(Sy+Sp+s),  dL/dp = -[2y(Sy+Sp+s) - (2I+s)]/(Sy+Sp+s)^2, s = smooth.
 * `sums` are the (possibly all-reduced, global-batch) sums produced above. */
int fmri_sigmoid_dice_bwd(const float* probs, const uint8_t* y_true, const double* sums, float* dlogits, int64_t n,
                          float smooth, float grad_scale, fmri_stream_t stream);

/* The other selectable losses of reference fetal_net/metrics.py (config_utils.py loss table), differentiated from the same sums:
 * kind 0 dice_coefficient_loss, 1 binary_crossentropy_loss (mean), 2 dice_and_xent (param = xent_weight), 3 focal_loss,
 * 4 vod_coefficient_loss, 5 double_dice_loss (param = ratio). */
int fmri_sigmoid_loss_bwd(const float* probs, const uint8_t* y_true, const double* sums, float* dlogits, int64_t n, int kind,
                          float param, float smooth, float grad_scale, fmri_stream_t stream);
/* ---- weighted_dice_coefficient_loss — reference fetal_net/metrics.py:39-55 (values pinned by reference test/test_metrics.py:10-38):
 * Dice per (sample, label) over the voxel axes with smooth = 1e-5, mean over the G = nsamples * L groups.  probs / y_true are indexed
 * [(n * vox + v) * L + l] like the logits.  fwd: gsums [G][3] doubles (zeroed here) = sum y*p, sum y, sum p per group;
 * sums[10] += sum_g (2 I_g + s) / (Sy_g + Sp_g + s), sums[11] += G (both additive over ranks: loss = -sums[10] / sums[11]).
 * bwd: dlogits = grad_scale * dL/dp * p (1 - p) with the groups' own sums and the (possibly all-reduced) group count sums[11]. */
int fmri_weighted_dice_fwd(const float* probs, const uint8_t* y_true, double* gsums, double* sums, int nsamples, int64_t vox, int L,
                           float smooth, fmri_stream_t stream);
int fmri_weighted_dice_bwd(const float* probs, const uint8_t* y_true, const double* gsums, const double* sums, float* dlogits,
                           int nsamples, int64_t vox, int L, float smooth, float grad_scale, fmri_stream_t stream);
/* the same with a per-voxel weight (device, n floats) on the cross-entropy term: sums[8] = sum(weight * xent); gradient kinds 1 and 2.
 * reference metrics.py:72-76 weighted_cross_entropy_loss and :89-95 dice_and_xent_mask (weight = exp(-distance_mask / sigma)) */
int fmri_sigmoid_dice_fwd_weighted(const float* logits, const uint8_t* y_true, const float* weight, float* probs, double* sums, int64_t n,
                                   fmri_stream_t stream);
int fmri_sigmoid_loss_bwd_weighted(const float* probs, const uint8_t* y_true, const float* weight, const double* sums, float* dlogits,
                                   int64_t n, int kind, float param, float smooth, float grad_scale, fmri_stream_t stream);

/* ---- MaxPooling3D(2,2,2) — reference unet.py:51.  D,H,W are the INPUT dims (even). */
int fmri_maxpool3d_2x_fwd(const void* x, void* y, int N, int D, int H, int W, int C, int dtype, int planar,
                          fmri_stream_t stream);
/* MaxPool3DGrad fused with the skip-gradient add and the ReluGrad of the producer:
 *   dx[v][c] = ( (add ? add[v*add_ld + add_off + c] : 0) + (x[v][c] is the first max of its window ? dy[win][c] : 0) ) * (x>0 || !relu_mask) */
int fmri_maxpool3d_2x_bwd(const void* x, const void* dy, const void* add, int add_ld, int add_off, void* dx, int N,
                          int D, int H, int W, int C, int relu_mask, int dtype, int planar, fmri_stream_t stream);

/* ---- UpSampling3D(2) — reference unet.py:138.  Materialising forward (writes channel slice [y_off, y_off+C) of a
 * tensor with y_ld channels); D,H,W are the LOW-resolution dims.  The conv kernels do not need it (fused). */
int fmri_upsample_nearest2x_fwd(const void* x, void* y, int y_ld, int y_off, int N, int D, int H, int W, int C, int dtype,
                                int planar, fmri_stream_t stream);
/* backward: dx[v][c] = (sum over the 2x2x2 children of dy[child*dy_ld + dy_off + c]) * (xmask[v][c] > 0 || !xmask) */
int fmri_upsample_nearest2x_bwd(const void* dy, int dy_ld, int dy_off, const void* xmask, void* dx, int N, int D, int H,
                                int W, int C, int dtype, int planar, fmri_stream_t stream);

/* ---- BatchNormalization(axis=1) / keras-contrib InstanceNormalization(axis=1) fused with the block's activation — reference
 * unet.py:103-115.  x, y: [N][V][C] (V = D*H*W voxels).  per_instance = 0: statistics over all N*V voxels per channel (Keras
 * training-mode batch norm, eps inside the sqrt); 1: per (sample, channel) (instance norm; eps_on_std = 1 reproduces
 * keras-contrib's (x-mean)/(std+eps)).  stats [G][C][3] fp32 = {mean, 1/s, 1/sigma} is written for the backward (G = N or 1);
 * ws = [G][C][2] doubles of scratch: ZERO on first use (allocate it zeroed) - every fmri_norm_* call leaves it zero again (the last
 * kernel that reads the sums clears them: one launch per layer and pass less than clearing in front).  y = act(gamma*(x-mean)/s + beta). */
int fmri_norm_act_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, double* ws, int N, int64_t V,
                      int C, int per_instance, float eps, int eps_on_std, int act, float alpha, int dtype, fmri_stream_t stream);
/* dy is the gradient w.r.t. y (NOT yet multiplied by act'): dz = dy*act'(y); dgamma/dbeta (fp32, ACCUMULATED) ;
 * dx = gamma*[(dz-mean(dz))/s - xhat*mean(dz*xhat)/sigma].  dx may alias dy.  ws: ZERO ON ENTRY (see above), left zero. */
int fmri_norm_act_bwd(const void* x, const void* y, const void* dy, const float* gamma, const float* stats, void* dx, float* dgamma,
                      float* dbeta, double* ws, int N, int64_t V, int C, int per_instance, int act, float alpha, int dtype,
                      fmri_stream_t stream);
/* the same without reading y: the sign of the block's output is recomputed from x, gamma, beta and the stored statistics with the very
 * operations fmri_norm_act_fwd used (so it agrees with the stored y bit for bit) - each of the two HBM-bound backward passes reads two
 * tensors instead of three.  ws: ZERO ON ENTRY, left zero (a caller with uninitialised scratch, or one whose sequence was cut between a
 * summing launch and its reader, gets wrong statistics without an error: zero it again). */
int fmri_norm_act_bwd_x(const void* x, const void* dy, const float* gamma, const float* beta, const float* stats, void* dx, float* dgamma,
                        float* dbeta, double* ws, int N, int64_t V, int C, int per_instance, int act, float alpha, int dtype,
                        fmri_stream_t stream);

/* ---- Normalisation tails of the conv launches (round 3; north-star "fused InstanceNorm+ReLU"; reference create_convolution_block,
 * fetal_net/model/unet3d/unet.py:103-115, isensee2017.py:12: Conv3D -> BatchNormalization | InstanceNormalization -> activation).
 * The conv in front of the normalisation layer sums its own output for the layer's statistics (no reduction pass over the tensor), and
 * the input-gradient launch behind a normalised block applies the activation's derivative and forms the two reductions of the
 * normalisation's backward pass - both in the asynchronous epilogue of the bf16 MFMA kernel, i.e. for launches with more (tile, channel
 * block) pairs than CUs: fmri_conv3d_fwd_ntail_ok(C0, C1, Cout, ...) != 0.  Elsewhere the caller keeps fmri_conv3d_fwd +
 * fmri_norm_act_fwd / fmri_norm_act_bwd_x (same results up to the order of summation).  ws: fmri_norm_tail_ws_doubles(G, C) doubles,
 * zero on first use and left zero by every call like the scratch of fmri_norm_act_fwd: the totals [G][C][2] in front (what the *_pre
 * functions read and clear), behind them one block per workgroup of the persistent launch - every workgroup sums into its own, a small
 * kernel folds and clears them (1,024 waves adding to the same 128 addresses with device-scope atomics cost more than the reduction pass
 * the tail replaces). */
int64_t fmri_norm_tail_ws_doubles(int G, int C);
int fmri_conv3d_fwd_ntail_ok(int C0, int C1, int Cout, int N, int D, int H, int W, int dtype);
/* fmri_conv3d_fwd (3-D, MFMA path) + ws[g][c] = {sum y, sum y^2} over the bf16 values as stored */
int fmri_conv3d_fwd_stats(const void* src0, int C0, int up0, const void* src1, int C1, const void* w, const float* bias, void* y, int N, int D,
                          int H, int W, int Cout, int act, float alpha, double* ws, int per_instance, int dtype, fmri_stream_t stream);
/* fmri_conv3d_upcat_fwd + the same sums (needs fmri_conv3d_fwd_ntail_ok(C1, 0, Cout, ...): the skip launch finishes and sums the output) */
int fmri_conv3d_upcat_fwd_stats(const void* src0_low, int C0, const void* src1, int C1, const void* w_up_fwd, const void* w_skip_fwd,
                                const float* bias, void* y, int N, int D, int H, int W, int Cout, int act, float alpha, double* ws,
                                int per_instance, int dtype, fmri_stream_t stream);
/* fmri_norm_act_fwd without its reduction pass: ws already holds {sum x, sum x^2} (per_instance 0 | 1) */
int fmri_norm_act_fwd_pre(const void* x, const float* gamma, const float* beta, void* y, float* stats, double* ws, int N, int64_t V, int C,
                          int per_instance, float eps, int eps_on_std, int act, float alpha, int dtype, fmri_stream_t stream);
/* Keras BatchNormalization moving statistics after a training forward (batch statistics in stats[0][C][3], M = elements per channel):
 * moving = momentum * moving + (1 - momentum) * {mean, var * M / (M - 1 - eps)} */
int fmri_norm_moving_update(const float* stats, float* moving_mean, float* moving_var, int C, double M, float momentum, float eps,
                            fmri_stream_t stream);
/* nss[g][c] = {scale, shift} with z = fma(x, scale, shift) the normalised pre-activation - the apply pass's own operations */
int fmri_norm_scale_shift(const float* stats, const float* gamma, const float* beta, float* nss, int G, int C, fmri_stream_t stream);
/* fmri_conv3d_dgrad (3-D, MFMA path; needs fmri_conv3d_fwd_ntail_ok(Cout, 0, Cin, ...)) whose output feeds a normalised block's backward:
 * dz = conv_dgrad(dy) * act'(z(x)), x = that block's conv output, nss its {scale, shift}; ws[g][c] = {sum dz, sum dz * x} */
int fmri_conv3d_dgrad_norm(const void* dy, int Cout, const void* w_dgrad, const void* x, const float* nss, void* dz, int N, int D, int H, int W,
                           int Cin, int act, float alpha, double* ws, int per_instance, int dtype, fmri_stream_t stream);
/* the rest of the normalisation's backward pass from dz and those sums: dgamma / dbeta (ACCUMULATED) and dx (may alias dz) */
int fmri_norm_act_bwd_pre(const void* x, const void* dz, const float* gamma, const float* stats, void* dx, float* dgamma, float* dbeta,
                          double* ws, int N, int64_t V, int C, int per_instance, int dtype, fmri_stream_t stream);

/* ---- Deconvolution3D(filters, kernel_size=(2,2,2), strides=(2,2,2)) — reference unet.py:135.  x [N][D][H][W][Cin] ->
 * y [N][2D][2H][2W][Cout]; w [8 taps = ad*4+ah*2+aw][Cout][Cin] (dtype); planar: (1,2,2) taps 0..3 and D unchanged. */
int fmri_deconv3d_k2s2_fwd(const void* x, const void* w, const float* b, void* y, int N, int D, int H, int W, int Cin, int Cout,
                           int dtype, int planar, fmri_stream_t stream);
/* dy = channel slice [dy_off, dy_off+Cout) of a tensor with dy_ld channels; dx (optional, masked by xmask > 0), dw [8][Cout][Cin]
 * and db [Cout] fp32 ACCUMULATED (either may be NULL). */
int fmri_deconv3d_k2s2_bwd(const void* x, const void* w, const void* dy, int dy_ld, int dy_off, const void* xmask, void* dx, float* dw,
                           float* db, int N, int D, int H, int W, int Cin, int Cout, int dtype, int planar, fmri_stream_t stream);

/* ---- Conv3D with kernel 1x1x1 (any Cin -> Cout) or 3x3x3 stride 2, 'same' (TensorFlow padding: even extent 0 before / 1
 * after) — reference isensee2017.py:51 (strides=(2,2,2)), :95-98 (1x1x1 localisation conv), :66 (segmentation heads).
 * x [N][D][H][W][Cin], w [k^3][Cout][Cin] (dtype), y [N][ceil(D/s)][ceil(H/s)][ceil(W/s)][Cout]. */
int fmri_conv3d_direct_fwd(const void* x, const void* w, const float* bias, void* y, int N, int D, int H, int W, int Cin, int Cout,
                           int ksize, int stride, int act, float alpha, int dtype, fmri_stream_t stream);
/* dx (optional) = input gradient; dw [k^3][Cout][Cin], db [Cout] fp32 ACCUMULATED (optional). */
int fmri_conv3d_direct_bwd(const void* x, const void* w, const void* dy, void* dx, float* dw, float* db, int N, int D, int H, int W,
                           int Cin, int Cout, int ksize, int stride, int dtype, fmri_stream_t stream);
/* 2-D twins — reference model/unet/isensee.py:49 (strides=(2,2)), :96 (kernel=(1,1)), :59 (segmentation heads): x [1][S][H][W][Cin] =
 * S slices stacked along the planar axis, which the stride leaves alone; y [1][S][ceil(H/s)][ceil(W/s)][Cout].  The filter image keeps
 * the k^3 layout with the 2-D kernel in its centre kd plane; the other planes are never read and their gradient is never written. */
int fmri_conv2d_direct_fwd(const void* x, const void* w, const float* bias, void* y, int S, int H, int W, int Cin, int Cout, int ksize,
                           int stride, int act, float alpha, int dtype, fmri_stream_t stream);
int fmri_conv2d_direct_bwd(const void* x, const void* w, const void* dy, void* dx, float* dw, float* db, int S, int H, int W, int Cin,
                           int Cout, int ksize, int stride, int dtype, fmri_stream_t stream);
/* y = a + b (residual Add, reference isensee2017.py:55; gradient fan-in of multiply-consumed tensors) */
int fmri_add(const void* a, const void* b, void* y, int64_t n, int dtype, fmri_stream_t stream);
/* dx = dy * act'(y): ReluGrad / LeakyRelu gradient from the stored post-activation tensor (dx may alias dy) */
int fmri_act_bwd(const void* y, const void* dy, void* dx, int act, float alpha, int64_t n, int dtype, fmri_stream_t stream);
/* dst[v][c] (+)= src[v*ld + off + c] — channel slice of a wider tensor (splitting a concat gradient; accumulate != 0 adds) */
int fmri_slice_channels(const void* src, int ld, int off, void* dst, int C, int64_t nvox, int accumulate, int dtype,
                        fmri_stream_t stream);
/* y[n][v][c] = x[n][v][c] * scale[n][c] — SpatialDropout3D (reference isensee2017.py:109): the host draws the 0 | 1/(1-p) mask */
int fmri_channel_scale(const void* x, const float* scale, void* y, int N, int64_t V, int C, int dtype, fmri_stream_t stream);

/* ---- bit-reproducible gradients (round 3).  The default step adds workgroups' partial sums to the gradient buffer with fp32 atomics, in
 * arrival order.  After fmri_set_deterministic(G, shadow, n) - G the fp32 gradient buffer every dw / db pointer of the following calls
 * points into, shadow n zeroed int64 - those partial sums are rounded to 2^-40 fixed point and added with 64-bit integer atomics to
 * shadow[i] instead (integer adds commute exactly); fmri_deterministic_finish adds shadow * 2^-40 to G and clears it (call it once per
 * backward pass, after every gradient kernel, before the optimizer).  The metric sums of fmri_sigmoid_dice_fwd meet the same way (2^-20).
 * Process-wide state (one engine at a time); (NULL, NULL, 0) switches it off.  Covered: the 3x3x3 / first-layer / 1x1x1 / direct
 * weight and bias gradients of the plain U-Net step; NOT covered: the parity-form weight gradient's scratch (fmri_conv3d_upcat_wgrad*:
 * use fmri_conv3d_wgrad with up0), the normalisation statistics, fmri_weighted_dice_fwd, fmri_border_class_sums. */
int fmri_set_deterministic(float* grad_base, void* shadow_i64, int64_t n);
int fmri_deterministic_finish(float* grad_base, void* shadow_i64, int64_t n, fmri_stream_t stream);

/* ---- Keras Adam.get_updates — reference unet.py:85.  lr_t = lr*sqrt(1-b2^t)/(1-b1^t) is computed by the host.
 * g is multiplied by grad_scale first.  p -= lr_t * m/(sqrt(v)+eps). One launch over the flat parameter buffer. */
int fmri_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr_t, float beta1, float beta2,
                   float eps, float grad_scale, fmri_stream_t stream);

/* ---- sliding-window overlap-add — reference prediction.py:98-114 (batch_iterator/get_patch_from_3d_data), :188-193, :210.
 * vol fp32 [X][Y][Z] (already padded by the host as prediction.py:138-146); idx int32 [B][3] device; tiles out (dtype)
 * [B][px][py][pz] (C = 1, NDHWC == NCDHW). */
int fmri_tile_gather(const float* vol, int X, int Y, int Z, const int32_t* idx, int B, int px, int py, int pz, void* tiles,
                     int dtype, fmri_stream_t stream);
/* acc (double) [X][Y][Z][C] += pred[b][px][py][pz][C] (fp32); cnt (int32) [X][Y][Z] += 1 */
int fmri_tile_scatter_accumulate(const float* pred, const int32_t* idx, int B, int px, int py, int pz, int C, double* acc,
                                 int32_t* cnt, int X, int Y, int Z, fmri_stream_t stream);
/* out (double) [n][C] = acc / cnt ; *bad (int32, accumulated) counts voxels with cnt == 0 (reference asserts none) */
int fmri_tile_finalize(const double* acc, const int32_t* cnt, double* out, int32_t* bad, int64_t nvox, int C,
                       fmri_stream_t stream);

/* ---- device-side patch sampler + intensity augmentation.  Replaces, per training patch, the host chain
 * reference fetal_net/generator.py:246-328 (add_data / extract_patch) -> augment.py:222-377 (augment_data) ->
 * utils/utils.py:100-113 (interpolate_affine_range = scipy.ndimage.map_coordinates, mode='constant').
 * out[(i*ny + j)*out_ld + k] = sample of vol [X][Y][Z] at  affine . (x0+i, y0+j, z0+k, 1)   for i<nx, j<ny, k<nz.
 * affine: 12 doubles, row-major 3x4, HOST memory (read at enqueue).  order 0 = nearest (floor(c+0.5)), 1 = trilinear; a source
 * point outside [0, n-1] on any axis gives cval.  vol_dtype F32|U8, out_dtype F32|BF16 (U8 -> U8|F32|BF16).  out_ld >= nz lets a
 * patch land in a channel slice of a wider [..][C] tensor (2-D models: previous-slice truth appended to the slice stack). */
int fmri_affine_sample(const void* vol, int vol_dtype, int X, int Y, int Z, const double* affine, int x0, int y0, int z0, int nx, int ny,
                       int nz, int order, float cval, void* out, int out_dtype, int out_ld, fmri_stream_t stream);
/* out2 (device, 2 floats) = {min, max} of x[0..n) — the image range rescale_intensity(out_range='image') and MinMaxScaler need */
int fmri_minmax(const void* x, int64_t n, int dtype, float* out2, fmri_stream_t stream);
/* contrast != 0: x = clip(x, lo, hi) rescaled from [lo, hi] to [stats[0], stats[1]] (skimage rescale_intensity, reference
 * augment.py:125-128); then x *= mult (augment.py:351-352).  stats = fmri_minmax of x BEFORE the call (device). In place. */
int fmri_rescale_intensity(void* x, int64_t n, int dtype, const float* stats, int contrast, float lo, float hi, float mult,
                           fmri_stream_t stream);
/* MinMaxScaler((0,1)) -> + noise*sigma (kind 0, gaussian) or + x*noise*sigma (kind 1, speckle) -> clip [0,1] -> inverse scaling
 * (reference augment.py:99-110).  noise: n fp32 N(0,1) draws on the device; stats = fmri_minmax of x before the call. In place. */
int fmri_noise_augment(void* x, int64_t n, int dtype, const float* stats, const float* noise, int kind, float sigma,
                       fmri_stream_t stream);
/* shot noise (reference augment.py:87-94): MinMaxScaler((0,1)) -> floor(x * 1023) / 1023 -> skimage random_noise('poisson', clip=True) ->
 * inverse scaling, in three phases around the caller's Poisson draw: 0 = mark the occupied quantisation levels in present (int32 [1024],
 * zeroed by the caller); 1 = rates[t] = q[t] * vals with vals = 2 ** ceil(log2(number of occupied levels)) (skimage's rule); 2 = x[t] from
 * draws[t] ~ Poisson(rates[t]) (fp32, device).  stats = fmri_minmax of x before phase 0. */
int fmri_shot_noise_step(void* x, int64_t n, int dtype, const float* stats, int* present, float* rates, const float* draws, int phase,
                         fmri_stream_t stream);
/* imgaug ElasticTransformation as the reference applies it (fetal_net/augment.py:149-170; in the DEFAULT config, fetal/config_utils.py:104-107):
 * dst[i][j][c] = src[:, :, c] sampled at (i - d0[i][j], j - d1[i][j]) - one in-plane displacement field for every slice c and for image, truth,
 * previous-slice truth and mask alike; order 1 (image: bilinear) or 0 (labels: nearest, floor(c + 0.5)), mode 'nearest' (coordinates clamped) -
 * the scipy branch of imgaug 0.4.0's `_map_coordinates` (its cv2.remap branch, taken for float images where cv2 is importable, quantises
 * coordinates to 1/32 pixel; parity unpinned, see oracle/augment_oracle.py).
 * src, dst: [X][Y] rows of src_ld / dst_ld elements (>= C), dtype FMRI_F32 or FMRI_U8; d0, d1 fp32 [X][Y].  src != dst. */
int fmri_elastic_warp(const void* src, int dtype, int X, int Y, int C, int src_ld, const float* d0, const float* d1, int order, void* dst,
                      int dst_ld, fmri_stream_t stream);
/* imgaug PiecewiseAffine on the reference's 2 x 2 grid (fetal_net/augment.py:131-146; commented out in the default config): output pixel (i, j) of
 * every slice reads the source at tri_t . (i, j, 1), t = 0 where j * X <= i * Y (the triangle below the diagonal (0,0) - (X,Y) of skimage's Delaunay
 * triangulation of the four corners), else t = 1; tri: 2 x (2 x 3) fp64 on the HOST, rows = (source row, source column).  order 1 / 0, scipy
 * map_coordinates mode 'constant' with cval 0 (imgaug's defaults).  dtype FMRI_F32 or FMRI_U8; src != dst. */
int fmri_piecewise_affine2(const void* src, int dtype, int X, int Y, int C, int src_ld, const double* tri, int order, void* dst, int dst_ld,
                           fmri_stream_t stream);
/* imgaug CoarseDropout as the reference applies it (fetal_net/augment.py:116-120, :373-375; DEFAULT config, config_utils.py:109-113): a voxel
 * whose cell of the low-resolution grid keep[hs][ws][kc] (uint8; kc = C: one grid per slice = per_channel, or 1) is 0 takes the patch minimum
 * (0 in the reference's [0, 255] scaling); the grid is enlarged by nearest neighbour, source index = min(floor(i * hs / X), hs - 1).
 * stats = fmri_minmax of x before the call.  In place; dtype FMRI_F32 or FMRI_BF16. */
int fmri_coarse_dropout(void* x, int dtype, int X, int Y, int C, int ld, const uint8_t* keep, int hs, int ws, int kc, const float* stats,
                        fmri_stream_t stream);
/* ---- the same intensity steps with the random draws made IN the kernel and over the B patches of a batch in ONE launch.
 * Draws: Philox4x32-10 keyed by `seed`, counter = (element or grid-cell index, seqs[b], rejection round) - the result depends on (seed,
 * seqs[b]) only, not on the launch geometry or on B; seqs[b] == 0 means "patch b skips this step".  Min / max chained: each call reads
 * stats[b] = {min, max} of patch b as it is and - when it rewrites the patch - leaves the min / max of the NEW values there for the next
 * step.  Patch b: x + b * stride elements; stats: device float [B][2]; ws: device int32 [B][FMRI_AUG_WS_INTS], zeroed ONCE by the caller
 * (every call leaves it zeroed; one ws per stream of calls).  seqs, params, grids, alphas: HOST arrays, read at enqueue.
 * A training batch of the reference's default config (fetal/config_utils.py:81-123: contrast, shot noise, speckle / gaussian noise with
 * probability 1/2 each, elastic transform, coarse dropout; reference fetal_net/augment.py:344-375) is 13 launches for the whole batch
 * instead of ~35 per patch with the draws taken from torch's generator (tools/r05/prof_generator.py).  The fed-draw forms above stay: the
 * parity tests hand them the oracle's own draws. */
#define FMRI_AUG_WS_INTS 1568
/* fmri_affine_sample for B patches: patch b reads vols[b] (extent dims[3b..], all of vol_dtype) at affines[12b..] . (corners[3b..] + (i, j, k), 1)
 * with outside value cvals[b] and lands at out + b * out_stride elements.  vols: HOST array of device pointers. */
int fmri_affine_sample_batch(int B, const void* const* vols, const int* dims, const double* affines, const int* corners, const float* cvals,
                             int vol_dtype, int nx, int ny, int nz, int order, void* out, int out_dtype, int out_ld, int64_t out_stride,
                             fmri_stream_t stream);
/* stats[b] = {min, max} of patch b: fmri_minmax in ONE launch for all patches (the last workgroup of a patch decodes and re-arms ws[b]) */
int fmri_minmax_ws_batch(const void* x, int64_t n, int64_t stride, int B, int dtype, float* stats, int* ws, fmri_stream_t stream);
/* fmri_rescale_intensity (reference augment.py:125-128, :351-352) per patch; params[b] = {mode, lo, hi, mult}: mode 0 skip, 1 multiply by
 * mult only, 2 contrast to [lo, hi] then multiply.  Updates stats[b]. */
int fmri_rescale_intensity_ws_batch(void* x, int64_t n, int64_t stride, int B, int dtype, float* stats, int* ws, const float* params,
                                    fmri_stream_t stream);
/* fmri_noise_augment with N(0,1) draws by Box-Muller; kind 0 gaussian, 1 speckle (reference augment.py:99-110). In place; updates stats[b]. */
int fmri_noise_rng_batch(void* x, int64_t n, int64_t stride, int B, int dtype, float* stats, int* ws, int kind, float sigma, uint64_t seed,
                         const uint32_t* seqs, fmri_stream_t stream);
/* fmri_shot_noise_step phases 0-2 (reference augment.py:87-94) in two launches, the Poisson draws as numpy's legacy sampler makes them
 * (lam < 10: product of uniforms; else Hoermann's PTRS) - what np.random.poisson runs under skimage's random_noise. Updates stats[b]. */
int fmri_shot_noise_rng_batch(void* x, int64_t n, int64_t stride, int B, int dtype, float* stats, int* ws, uint64_t seed, const uint32_t* seqs,
                              fmri_stream_t stream);
/* imgaug ElasticTransformation's displacement fields (reference augment.py:149-170 -> imgaug 0.4.0 _generate_shift_maps) in one launch:
 * uniform(-1, 1) noise on the image padded by k on every side, blurred with the k-tap kernel `weights` (device, fp64 [k], normalised; k odd,
 * <= 31) along both axes, times alphas[b], padding cropped.  d: device fp32 [B][2][X][Y]; d[b][0] = shift along axis 0 (imgaug's dy), d[b][1]
 * = along axis 1 (dx); a patch with seqs[b] == 0 gets zero fields.  Noise pixel p of the padded (2, X + 2k, Y + 2k) grid (block 0 = dx)
 * = 2 u - 1 with u = (word (p & 3) of Philox(p >> 2, 0, seqs[b], 1) >> 8) / 2^24. */
int fmri_elastic_fields_rng_batch(float* d, int X, int Y, int k, const double* weights, const float* alphas, uint64_t seed, const uint32_t* seqs,
                                  int B, fmri_stream_t stream);
/* fmri_elastic_warp for B patches with the fields of fmri_elastic_fields_rng_batch: patch b = src + b * src_stride -> dst + b * dst_stride */
int fmri_elastic_warp_batch(const void* src, int dtype, int X, int Y, int C, int src_ld, int64_t src_stride, const float* d, int order, void* dst,
                            int dst_ld, int64_t dst_stride, int B, fmri_stream_t stream);
/* fmri_coarse_dropout with the keep grid drawn in the kernel: cell (si, sj[, c]) of patch b's grids[2b] x grids[2b+1] (x kc) grid is dropped
 * when its uniform draw is < rate (imgaug CoarseDropout(p=rate); reference augment.py:116-120).  stats[b] = min / max of the patch; not
 * rewritten (the dropped voxels take the minimum). */
int fmri_coarse_dropout_rng_batch(void* x, int dtype, int X, int Y, int C, int ld, int64_t stride, int B, const int* grids, int kc, float rate,
                                  const float* stats, uint64_t seed, const uint32_t* seqs, fmri_stream_t stream);
/* one axis of skimage.filters.gaussian on an fp32 patch [X][Y][Z] (reference augment.py:113-114): weights fp64 [2*radius+1] on the device,
 * sums in fp64 in scipy's order, result rounded to fp32; mode 0 = 'reflect', 1 = 'nearest' (skimage's default).  src != dst. */
int fmri_correlate1d_f32(const float* src, float* dst, int X, int Y, int Z, int axis, const double* weights, int radius, int mode,
                         fmri_stream_t stream);

/* ---- plumbing: dtype casts used around the boundary (fp32 <-> bf16), n elements */
int fmri_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, fmri_stream_t stream);

/* ---- post-processing of a predicted probability volume [X][Y][Z] (reference fetal_net/postprocess.py:7-19, used by
 * prod/predict_nifti2.py:77-95): scipy.ndimage.gaussian_filter -> "> threshold" -> binary_fill_holes -> largest connected component.
 * fmri_correlate1d_f64: one axis of the separable gaussian, mode 'reflect', weights [2*radius+1] fp64 on the device (the caller computes
 *   them as scipy does); sums in scipy's order, so three calls (axis 0, 1, 2) reproduce gaussian_filter bit for bit.  src != dst.
 * fmri_fill_holes_step / fmri_largest_component_step: phase 0 = initialise, phase 1 = `sweeps` propagation sweeps (6-connectivity;
 *   *changed, a device int the caller zeroes, is set while the region still grows: repeat phase 1 until it stays 0), phase 2 = write the
 *   result mask (uint8 0/1).  Largest component: ties go to the component whose first voxel comes first in C order (scipy's numbering);
 *   counts = int32 [X*Y*Z + 1] scratch, best = 8-byte scratch. */
int fmri_correlate1d_f64(const double* src, double* dst, int X, int Y, int Z, int axis, const double* weights, int radius,
                         fmri_stream_t stream);
int fmri_threshold_f64(const double* src, uint8_t* dst, int64_t n, double threshold, fmri_stream_t stream);
int fmri_fill_holes_step(const uint8_t* mask, uint8_t* reached, uint8_t* out, int X, int Y, int Z, int phase, int sweeps, int* changed,
                         fmri_stream_t stream);
int fmri_largest_component_step(const uint8_t* mask, int32_t* labels, int32_t* counts, unsigned long long* best, uint8_t* out, int X, int Y,
                                int Z, int phase, int sweeps, int* changed, fmri_stream_t stream);

/* ---- PatchGAN discriminator head and the adversarial coupling.  Reference
 * fetal_net/model/discriminator/all_dis_3d.py:11-72 (conv blocks of the segmentation path's layer kinds + AveragePooling3D,
 * GlobalAveragePooling3D, Dense(128, LeakyReLU) x fc_layers, Dense(1, 'sigmoid'), loss binary_crossentropy, metric 'mae') and
 * fetal/experiments/train_adv.py:92-125, :165-180 (discriminator inputs; generator trained through the frozen discriminator).
 * fmri_avgpool3d_2x: x [N][D][H][W][C] -> y [N][D/2][H/2][W/2][C] ('valid': floor, an odd trailing plane is ignored / gets a zero
 *   gradient); planar: window 1x2x2.  D,H,W are the INPUT dims in both directions.
 * fmri_global_avgpool: x [N][V][C] (dtype) <-> y [N][C] fp32.
 * fmri_dense: y[n][m] = act(b[m] + sum_k x[n][k] * w[k][m]), w the Keras kernel [K][M]; bwd takes the OUTPUT y for the activation's
 *   derivative, ACCUMULATES into dw / db (nullable) and writes dx (nullable).
 * fmri_sigmoid_bce: probs = sigmoid(logits); sums[0] += sum_i BCE_i with Keras' clip of p to [1e-7, 1-1e-7], sums[1] += sum |p - t|,
 *   sums[2] += n (the caller zeroes sums); bwd: dlogits = scale * (p - t), 0 where the clip is active.  target is a FLOAT (soft labels).
 * fmri_sigmoid_chain: dlogits (=|+=) scale * dprobs[v * ld + l] * p * (1 - p) for the generator's probs [nvox][n_labels] (fp32);
 *   dprobs (dtype) = the leading channels of the discriminator's input gradient.
 * fmri_discriminator_input: out [nvox][out_ld] (zero-filled past the used channels) = merge ? [x * s, x * (1 - s)] (mul_merge_maps,
 *   numpy channel broadcasting: C == 1, n_labels == 1 or C == n_labels) : [s, x] (Concatenate(axis=1)([segs, inputs])), s = probs fp32. */
int fmri_avgpool3d_2x_fwd(const void* x, void* y, int N, int D, int H, int W, int C, int dtype, int planar, fmri_stream_t stream);
int fmri_avgpool3d_2x_bwd(const void* dy, void* dx, int N, int D, int H, int W, int C, int dtype, int planar, fmri_stream_t stream);
int fmri_global_avgpool_fwd(const void* x, float* y, int N, int64_t V, int C, int dtype, fmri_stream_t stream);
int fmri_global_avgpool_bwd(const float* dy, void* dx, int N, int64_t V, int C, int dtype, fmri_stream_t stream);
int fmri_dense_fwd(const float* x, const float* w, const float* b, float* y, int N, int K, int M, int act, float alpha, fmri_stream_t stream);
int fmri_dense_bwd(const float* x, const float* w, const float* y, const float* dy, float* dx, float* dw, float* db, int N, int K, int M,
                   int act, float alpha, fmri_stream_t stream);
int fmri_sigmoid_bce_fwd(const float* logits, const float* target, float* probs, double* sums, int64_t n, fmri_stream_t stream);
int fmri_sigmoid_bce_bwd(const float* probs, const float* target, float* dlogits, int64_t n, float scale, fmri_stream_t stream);
int fmri_sigmoid_chain(const float* probs, const void* dprobs, int dprobs_ld, int n_labels, float* dlogits, int64_t nvox, float scale,
                       int accumulate, int dtype, fmri_stream_t stream);
int fmri_discriminator_input(const float* probs, int n_labels, const void* x, int C, int x_dtype, void* out, int out_ld, int out_dtype,
                             int64_t nvox, int merge, fmri_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FMRI_HIP_H */
