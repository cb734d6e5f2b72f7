// Kernels of the PatchGAN discriminator head and of the adversarial coupling (gfx950).
//
// Reference: fetal_net/model/discriminator/all_dis_3d.py:11-72 (conv_block = [Conv3D -> InstanceNormalization -> LeakyReLU] x 2 with a
// SpatialDropout3D in between, then AveragePooling3D(); after the last block GlobalAveragePooling3D -> Dense(128, LeakyReLU) x fc_layers
// -> Dense(1, 'sigmoid'); loss binary_crossentropy on soft labels, metric 'mae') and fetal/experiments/train_adv.py:173-180 (the
// generator is trained through the frozen discriminator: dL/dprobs of the segmentation comes back from the discriminator's input).
// The convolutions / normalisations / dropout are the kernels of the segmentation path; this file adds what only the discriminator has:
//   average pooling 2x2x2 (2-D: 1x2x2), global average pooling, the tiny dense layers, sigmoid + binary cross-entropy on float targets,
//   and the chain rule through the generator's sigmoid for a gradient that arrives on its probabilities.
// All of it is HBM- or latency-bound element work: one pass over the tensor, vector loads along the channel axis (NDHWC).
#include "common.h"

static __device__ __forceinline__ void decode_vox(int64_t v, int D, int H, int W, int& n, int& d, int& h, int& w) {
    w = (int)(v % W); v /= W;
    h = (int)(v % H); v /= H;
    d = (int)(v % D);
    n = (int)(v / D);
}

// ------------------------------------------------------------------------------------------------ AveragePooling3D(2) / AveragePooling2D(2)
// 'valid' pooling: output dims floor(d / 2); an odd trailing plane is not read (Keras) and gets a zero gradient.
template <typename T, int VEC>
__global__ void k_avgpool_fwd(const T* __restrict__ x, T* __restrict__ y, int N, int D, int H, int W, int C, int pd) {
    const int Do = pd ? D >> 1 : D, Ho = H >> 1, Wo = W >> 1, CG = C / VEC;
    const int64_t total = (int64_t)N * Do * Ho * Wo * CG;
    const int nt = 4 << pd;
    const float inv = 1.f / (float)nt;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cg = (int)(i % CG), n, d_o, ho, wo;
        decode_vox(i / CG, Do, Ho, Wo, n, d_o, ho, wo);
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (t >= nt) break;
            const int dd = (d_o << pd) + (t >> 2), hh = 2 * ho + ((t >> 1) & 1), ww = 2 * wo + (t & 1);
            float xv[VEC];
            ldv<T, VEC>(x + ((((int64_t)n * D + dd) * H + hh) * W + ww) * C + cg * VEC, xv);
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] += xv[k];
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] *= inv;
        stv<T, VEC>(y + (i / CG) * C + cg * VEC, acc);
    }
}
// one thread per INPUT voxel: dx = dy[parent] / window, 0 in an odd trailing plane
template <typename T, int VEC>
__global__ void k_avgpool_bwd(const T* __restrict__ dy, T* __restrict__ dx, int N, int D, int H, int W, int C, int pd) {
    const int Do = pd ? D >> 1 : D, Ho = H >> 1, Wo = W >> 1, CG = C / VEC;
    const int64_t total = (int64_t)N * D * H * W * CG;
    const float inv = 1.f / (float)(4 << pd);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cg = (int)(i % CG), n, d, h, w;
        decode_vox(i / CG, D, H, W, n, d, h, w);
        const int d_o = pd ? d >> 1 : d, ho = h >> 1, wo = w >> 1;
        float g[VEC];
        if (d_o < Do && ho < Ho && wo < Wo) {
            ldv<T, VEC>(dy + ((((int64_t)n * Do + d_o) * Ho + ho) * Wo + wo) * C + cg * VEC, g);
#pragma unroll
            for (int k = 0; k < VEC; ++k) g[k] *= inv;
        } else {
#pragma unroll
            for (int k = 0; k < VEC; ++k) g[k] = 0.f;
        }
        stv<T, VEC>(dx + (i / CG) * C + cg * VEC, g);
    }
}

extern "C" int fmri_avgpool3d_2x_fwd(const void* x, void* y, int N, int D, int H, int W, int C, int dtype, int planar, fmri_stream_t stream) {
    const int pd = planar ? 0 : 1;
    if (N <= 0 || C <= 0 || D < 1 || H < 2 || W < 2 || (pd && D < 2) || !x || !y) return FMRI_E_SHAPE;
    const int vec = pick_vec(C);
    const int grid = grid_for((int64_t)N * (pd ? D / 2 : D) * (H / 2) * (W / 2) * (C / vec));
    hipStream_t s = as_stream(stream);
    if (dtype == FMRI_F32) LAUNCH_TV(k_avgpool_fwd, float, vec, grid, 256, s, (const float*)x, (float*)y, N, D, H, W, C, pd);
    else if (dtype == FMRI_BF16) LAUNCH_TV(k_avgpool_fwd, bf16_t, vec, grid, 256, s, (const bf16_t*)x, (bf16_t*)y, N, D, H, W, C, pd);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
extern "C" int fmri_avgpool3d_2x_bwd(const void* dy, void* dx, int N, int D, int H, int W, int C, int dtype, int planar, fmri_stream_t stream) {
    const int pd = planar ? 0 : 1;
    if (N <= 0 || C <= 0 || D < 1 || H < 2 || W < 2 || (pd && D < 2) || !dy || !dx) return FMRI_E_SHAPE;
    const int vec = pick_vec(C);
    const int grid = grid_for((int64_t)N * D * H * W * (C / vec));
    hipStream_t s = as_stream(stream);
    if (dtype == FMRI_F32) LAUNCH_TV(k_avgpool_bwd, float, vec, grid, 256, s, (const float*)dy, (float*)dx, N, D, H, W, C, pd);
    else if (dtype == FMRI_BF16) LAUNCH_TV(k_avgpool_bwd, bf16_t, vec, grid, 256, s, (const bf16_t*)dy, (bf16_t*)dx, N, D, H, W, C, pd);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

// ------------------------------------------------------------------------------------------------ GlobalAveragePooling3D / 2D
// x [N][V][C] -> y [N][C] fp32.  One workgroup per (sample, 64-channel slab): 64 channel lanes x 4 voxel lanes, fixed summation order
// (lane-strided partial sums, then the 4 partials in order) so that the result does not depend on the launch.
template <typename T>
__global__ void k_gap_fwd(const T* __restrict__ x, float* __restrict__ y, int64_t V, int C) {
    const int n = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    __shared__ float red[4][64];
    float acc = 0.f;
    if (c < C)
        for (int64_t v = part; v < V; v += 4) acc += to_f<T>(x[((int64_t)n * V + v) * C + c]);
    red[part][threadIdx.x & 63] = acc;
    __syncthreads();
    if (part == 0 && c < C) y[(int64_t)n * C + c] = (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) / (float)V;
}
template <typename T, int VEC>
__global__ void k_gap_bwd(const float* __restrict__ dy, T* __restrict__ dx, int N, int64_t V, int C) {
    const int CG = C / VEC;
    const int64_t total = (int64_t)N * V * CG;
    const float inv = 1.f / (float)V;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cg = (int)(i % CG);
        const int64_t n = (i / CG) / V;
        float g[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) g[k] = dy[n * C + cg * VEC + k] * inv;
        stv<T, VEC>(dx + (i / CG) * C + cg * VEC, g);
    }
}
extern "C" int fmri_global_avgpool_fwd(const void* x, float* y, int N, int64_t V, int C, int dtype, fmri_stream_t stream) {
    if (N <= 0 || V <= 0 || C <= 0 || !x || !y) return FMRI_E_SHAPE;
    dim3 grid((C + 63) / 64, N);
    hipStream_t s = as_stream(stream);
    if (dtype == FMRI_F32) k_gap_fwd<float><<<grid, 256, 0, s>>>((const float*)x, y, V, C);
    else if (dtype == FMRI_BF16) k_gap_fwd<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)x, y, V, C);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
extern "C" int fmri_global_avgpool_bwd(const float* dy, void* dx, int N, int64_t V, int C, int dtype, fmri_stream_t stream) {
    if (N <= 0 || V <= 0 || C <= 0 || !dy || !dx) return FMRI_E_SHAPE;
    const int vec = pick_vec(C);
    const int grid = grid_for((int64_t)N * V * (C / vec));
    hipStream_t s = as_stream(stream);
    if (dtype == FMRI_F32) LAUNCH_TV(k_gap_bwd, float, vec, grid, 256, s, dy, (float*)dx, N, V, C);
    else if (dtype == FMRI_BF16) LAUNCH_TV(k_gap_bwd, bf16_t, vec, grid, 256, s, dy, (bf16_t*)dx, N, V, C);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

// ------------------------------------------------------------------------------------------------ Dense (K -> M), fp32, tiny
// w is the Keras kernel [K][M]; act = FMRI_ACT_NONE | FMRI_ACT_LEAKY (Dense(128, activation=LeakyReLU()), all_dis_3d.py:42).
// Sizes are (batch <= a few dozen) x (<= 128) x (<= 128): latency-bound, one thread per output element, sequential sums.
static __device__ __forceinline__ float dense_act(float z, int act, float alpha) {
    if (act == FMRI_ACT_RELU) return z > 0.f ? z : 0.f;
    if (act == FMRI_ACT_LEAKY) return z > 0.f ? z : alpha * z;
    return z;
}
static __device__ __forceinline__ float dense_act_grad(float y, int act, float alpha) {      // from the OUTPUT (sign(y) = sign(z) for alpha > 0)
    if (act == FMRI_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (act == FMRI_ACT_LEAKY) return y > 0.f ? 1.f : alpha;
    return 1.f;
}
__global__ void k_dense_fwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ y, int K,
                            int M, int act, float alpha) {
    const int n = blockIdx.x;
    for (int m = threadIdx.x; m < M; m += blockDim.x) {
        float acc = b ? b[m] : 0.f;
        for (int k = 0; k < K; ++k) acc = fmaf(x[(int64_t)n * K + k], w[(int64_t)k * M + m], acc);
        y[(int64_t)n * M + m] = dense_act(acc, act, alpha);
    }
}
// block b < K: dw[b][:] += sum_n x[n][b] * dz[n][:];  block K: db[:] += sum_n dz[n][:];  blocks K+1 .. K+N: dx[n][:] = dz[n][:] . w^T
__global__ void k_dense_bwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ y, const float* __restrict__ dy,
                            float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db, int N, int K, int M, int act, float alpha) {
    const int blk = blockIdx.x;
    if (blk < K) {
        if (!dw) return;
        for (int m = threadIdx.x; m < M; m += blockDim.x) {
            float acc = 0.f;
            for (int n = 0; n < N; ++n)
                acc = fmaf(x[(int64_t)n * K + blk], dy[(int64_t)n * M + m] * dense_act_grad(y[(int64_t)n * M + m], act, alpha), acc);
            dw[(int64_t)blk * M + m] += acc;
        }
    } else if (blk == K) {
        if (!db) return;
        for (int m = threadIdx.x; m < M; m += blockDim.x) {
            float acc = 0.f;
            for (int n = 0; n < N; ++n) acc += dy[(int64_t)n * M + m] * dense_act_grad(y[(int64_t)n * M + m], act, alpha);
            db[m] += acc;
        }
    } else {
        if (!dx) return;
        const int n = blk - K - 1;
        for (int k = threadIdx.x; k < K; k += blockDim.x) {
            float acc = 0.f;
            for (int m = 0; m < M; ++m)
                acc = fmaf(dy[(int64_t)n * M + m] * dense_act_grad(y[(int64_t)n * M + m], act, alpha), w[(int64_t)k * M + m], acc);
            dx[(int64_t)n * K + k] = acc;
        }
    }
}
extern "C" int fmri_dense_fwd(const float* x, const float* w, const float* b, float* y, int N, int K, int M, int act, float alpha,
                              fmri_stream_t stream) {
    if (N <= 0 || K <= 0 || M <= 0 || !x || !w || !y) return FMRI_E_SHAPE;
    k_dense_fwd<<<N, 128, 0, as_stream(stream)>>>(x, w, b, y, K, M, act, alpha);
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
extern "C" int fmri_dense_bwd(const float* x, const float* w, const float* y, const float* dy, float* dx, float* dw, float* db, int N, int K,
                              int M, int act, float alpha, fmri_stream_t stream) {
    if (N <= 0 || K <= 0 || M <= 0 || !x || !w || !y || !dy) return FMRI_E_SHAPE;
    k_dense_bwd<<<K + 1 + N, 128, 0, as_stream(stream)>>>(x, w, y, dy, dx, dw, db, N, K, M, act, alpha);
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

// ------------------------------------------------------------------------------------------------ sigmoid + binary cross-entropy, float targets
// Keras 2.2 K.binary_crossentropy on probabilities: p clipped to [1e-7, 1 - 1e-7] before the logs; the discriminator's targets are soft
// labels drawn in [0.9, 1] / [0, 0.1] (train_adv.py:113-115), so the target is a float.  sums[0] += sum of the element losses,
// sums[1] += sum |p - t| (metric 'mae'), sums[2] += n.  One workgroup: n is batch-sized, the order of the sum is fixed.
__global__ void k_sigmoid_bce_fwd(const float* __restrict__ logits, const float* __restrict__ target, float* __restrict__ probs,
                                  double* __restrict__ sums, int64_t n) {
    double s0 = 0, s1 = 0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const float p = 1.f / (1.f + expf(-logits[i])), t = target[i];
        probs[i] = p;
        const float pc = fminf(fmaxf(p, 1e-7f), 1.f - 1e-7f);
        s0 += (double)(-(t * logf(pc) + (1.f - t) * logf(1.f - pc)));
        s1 += (double)fabsf(p - t);
    }
    __shared__ double red[2][256];
    red[0][threadIdx.x] = s0;
    red[1][threadIdx.x] = s1;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            red[0][threadIdx.x] += red[0][threadIdx.x + o];
            red[1][threadIdx.x] += red[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sums[0] += red[0][0];
        sums[1] += red[1][0];
        sums[2] += (double)n;
    }
}
// d(mean BCE)/dlogit = scale * (p - t) (scale carries 1/n and the loss weight); zero where the clip is active (TF clip_by_value)
__global__ void k_sigmoid_bce_bwd(const float* __restrict__ probs, const float* __restrict__ target, float* __restrict__ dl, int64_t n, float scale) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float p = probs[i];
        dl[i] = (p > 1e-7f && p < 1.f - 1e-7f) ? scale * (p - target[i]) : 0.f;
    }
}
extern "C" int fmri_sigmoid_bce_fwd(const float* logits, const float* target, float* probs, double* sums, int64_t n, fmri_stream_t stream) {
    if (n <= 0 || !logits || !target || !probs || !sums) return FMRI_E_SHAPE;
    k_sigmoid_bce_fwd<<<1, 256, 0, as_stream(stream)>>>(logits, target, probs, sums, n);
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
extern "C" int fmri_sigmoid_bce_bwd(const float* probs, const float* target, float* dlogits, int64_t n, float scale, fmri_stream_t stream) {
    if (n <= 0 || !probs || !target || !dlogits) return FMRI_E_SHAPE;
    k_sigmoid_bce_bwd<<<grid_for(n), 256, 0, as_stream(stream)>>>(probs, target, dlogits, n, scale);
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

// ------------------------------------------------------------------------------------------------ chain rule through the generator's sigmoid
// The combined model (train_adv.py:173-180) adds gd_loss_ratio * BCE(D(concat(segs, x)), valid) to the segmentation loss: its gradient
// arrives on the generator's probabilities.  dlogits (+)= scale * dprobs * p * (1 - p); dprobs has row stride `ld` (it is the first
// n_labels channels of the discriminator's input gradient) and element type fp32 or bf16.
template <typename T>
__global__ void k_sigmoid_chain(const float* __restrict__ probs, const T* __restrict__ dprobs, int ld, int L, float* __restrict__ dl, int64_t nvox,
                                float scale, int accumulate) {
    const int64_t total = nvox * L;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t v = i / L;
        const int l = (int)(i - v * L);
        const float p = probs[i];
        const float g = scale * to_f<T>(dprobs[v * ld + l]) * p * (1.f - p);
        dl[i] = accumulate ? dl[i] + g : g;
    }
}
extern "C" int fmri_sigmoid_chain(const float* probs, const void* dprobs, int dprobs_ld, int n_labels, float* dlogits, int64_t nvox, float scale,
                                  int accumulate, int dtype, fmri_stream_t stream) {
    if (nvox <= 0 || n_labels <= 0 || dprobs_ld < n_labels || !probs || !dprobs || !dlogits) return FMRI_E_SHAPE;
    const int grid = grid_for(nvox * n_labels);
    hipStream_t s = as_stream(stream);
    if (dtype == FMRI_F32) k_sigmoid_chain<float><<<grid, 256, 0, s>>>(probs, (const float*)dprobs, dprobs_ld, n_labels, dlogits, nvox, scale, accumulate);
    else if (dtype == FMRI_BF16) k_sigmoid_chain<bf16_t><<<grid, 256, 0, s>>>(probs, (const bf16_t*)dprobs, dprobs_ld, n_labels, dlogits, nvox, scale, accumulate);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

// ------------------------------------------------------------------------------------------------ discriminator input assembly
// d_in[v][0 .. L)     = probs[v][:]            (generator path: Concatenate(axis=1)([segs, inputs]), train_adv.py:175)
// d_in[v][L .. L + C) = x[v][:]
// d_in[v][L + C .. ld) = 0                     (channel padding of the bf16 engine)
// With `merge` = 1 the mul-merge maps of train_adv.py:92-95 are written instead: [x * s (C x L maps), x * (1 - s)], s = probs.
template <typename TX, typename TO>
__global__ void k_dis_input(const float* __restrict__ probs, int L, const TX* __restrict__ x, int C, TO* __restrict__ out, int ld, int64_t nvox,
                            int merge) {
    const int64_t total = nvox * ld;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t v = i / ld;
        const int c = (int)(i - v * ld);
        float r = 0.f;
        if (!merge) {
            if (c < L) r = probs[v * L + c];
            else if (c < L + C) r = to_f<TX>(x[v * C + (c - L)]);
        } else {
            const int P = C > L ? C : L;           // numpy broadcast of r[:, C] * s[:, L] along the channel axis (C == 1, L == 1 or C == L)
            if (c < 2 * P) {
                const int j = c < P ? c : c - P;
                const float s = probs[v * L + (L == 1 ? 0 : j)], xv = to_f<TX>(x[v * C + (C == 1 ? 0 : j)]);
                r = c < P ? xv * s : xv * (1.f - s);
            }
        }
        out[i] = from_f<TO>(r);
    }
}
extern "C" int fmri_discriminator_input(const float* probs, int n_labels, const void* x, int C, int x_dtype, void* out, int out_ld, int out_dtype,
                                        int64_t nvox, int merge, fmri_stream_t stream) {
    if (nvox <= 0 || n_labels <= 0 || C <= 0 || !probs || !x || !out) return FMRI_E_SHAPE;
    if (merge && !(C == 1 || n_labels == 1 || C == n_labels)) return FMRI_E_SHAPE;
    const int need = merge ? 2 * (C > n_labels ? C : n_labels) : n_labels + C;
    if (out_ld < need) return FMRI_E_SHAPE;
    const int grid = grid_for(nvox * out_ld);
    hipStream_t s = as_stream(stream);
#define DIS_IN(TX, TO) k_dis_input<TX, TO><<<grid, 256, 0, s>>>(probs, n_labels, (const TX*)x, C, (TO*)out, out_ld, nvox, merge)
    if (x_dtype == FMRI_F32 && out_dtype == FMRI_F32) DIS_IN(float, float);
    else if (x_dtype == FMRI_F32 && out_dtype == FMRI_BF16) DIS_IN(float, bf16_t);
    else if (x_dtype == FMRI_BF16 && out_dtype == FMRI_BF16) DIS_IN(bf16_t, bf16_t);
    else if (x_dtype == FMRI_BF16 && out_dtype == FMRI_F32) DIS_IN(bf16_t, float);
    else return FMRI_E_DTYPE;
#undef DIS_IN
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
