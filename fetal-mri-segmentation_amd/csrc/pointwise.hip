// HBM-bound kernels of the U-Net step: pooling, up-sampling, final 1x1x1 conv, sigmoid+Dice, Adam, weight packing,
// sliding-window tile gather / overlap-add, casts.  All channels-last, vectorised where the channel count allows.
#include <cstdlib>
#include "common.h"

FMRI_DET_TU(pointwise)
static __device__ unsigned long long g_isums[16];      // metric sums of the deterministic mode, 2^-20 fixed point

// ------------------------------------------------------------------------------------------------ version / errors
extern "C" int fmri_version(void) { return 100; }
extern "C" const char* fmri_error_string(int code) {
    switch (code) {
        case FMRI_OK: return "ok";
        case FMRI_E_SHAPE: return "unsupported shape";
        case FMRI_E_ALIGN: return "misaligned pointer";
        case FMRI_E_ARCH: return "unsupported architecture";
        case FMRI_E_LAUNCH: return "kernel launch failed";
        case FMRI_E_DTYPE: return "unsupported dtype";
        default: return "unknown error";
    }
}

// ------------------------------------------------------------------------------------------------ shader-clock stamps (measurement aid)
// One (s_memtime, s_memrealtime) pair per XCD: s_memtime ticks with the shader clock, s_memrealtime at a constant 100 MHz, so two stamps
// bracket a region's average shader clock: d(memtime) / d(memrealtime) x 0.1 GHz (MI355X guide, 'DVFS give-back' item 6).  The counters
// are per XCD, so a workgroup writes the slot of the XCD it runs on (HW_REG_XCC_ID); 64 workgroups reach all eight.
__global__ void k_clock_stamp(unsigned long long* out) {
    if (threadIdx.x != 0) return;
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u;       // XCC_ID[3:0], hwreg 20 on gfx942 / gfx950
    const unsigned long long t = __builtin_amdgcn_s_memtime(), r = __builtin_amdgcn_s_memrealtime();
    out[2 * xcc] = t;
    out[2 * xcc + 1] = r;
}
extern "C" int fmri_clock_stamp(unsigned long long* out16, fmri_stream_t stream) {
    if (!out16) return FMRI_E_SHAPE;
    k_clock_stamp<<<64, 64, 0, as_stream(stream)>>>(out16);
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

__device__ __forceinline__ void decode_vox(int64_t v, int Do, int Ho, int Wo, int& n, int& d, int& h, int& w) {
    w = (int)(v % Wo); v /= Wo;
    h = (int)(v % Ho); v /= Ho;
    d = (int)(v % Do);
    n = (int)(v / Do);
}

// ------------------------------------------------------------------------------------------------ max-pool 2x2x2
// One thread per (pooled voxel, channel group of VEC).  Reference: MaxPooling3D(pool_size) at unet3d/unet.py:51.
template <typename T, int VEC>
__global__ void k_maxpool_fwd(const T* __restrict__ x, T* __restrict__ y, int N, int D, int H, int W, int C, int pd) {
    const int Do = D >> pd, Ho = H >> 1, Wo = W >> 1, CG = C / VEC;      // pd = 0: planar (2-D slices), window 1x2x2
    const int64_t total = (int64_t)N * Do * Ho * Wo * CG;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cg = (int)(i % CG), n, d_o, ho, wo;
        decode_vox(i / CG, Do, Ho, Wo, n, d_o, ho, wo);
        float m[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) m[k] = -INFINITY;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (t >= (4 << pd)) break;
            int dd = (d_o << pd) + (t >> 2), hh = 2 * ho + ((t >> 1) & 1), ww = 2 * wo + (t & 1);
            float xv[VEC];
            ldv<T, VEC>(x + ((((int64_t)n * D + dd) * H + hh) * W + ww) * C + cg * VEC, xv);
#pragma unroll
            for (int k = 0; k < VEC; ++k) m[k] = fmaxf(m[k], xv[k]);
        }
        stv<T, VEC>(y + ((((int64_t)n * Do + d_o) * Ho + ho) * Wo + wo) * C + cg * VEC, m);
    }
}

// backward: one thread per (pooled voxel, channel group): recompute the window max, route dy to the FIRST max in
// (d,h,w) scan order, add the skip gradient, apply the producer's ReLU mask, write all 8 children.
template <typename T, int VEC>
__global__ void __launch_bounds__(256) k_maxpool_bwd(const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ add, int add_ld,
                              int add_off, T* __restrict__ dx, int N, int D, int H, int W, int C, int relu_mask, int pd) {
    const int Do = D >> pd, Ho = H >> 1, Wo = W >> 1, CG = C / VEC;
    const int64_t total = (int64_t)N * Do * Ho * Wo * CG;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cg = (int)(i % CG), n, d_o, ho, wo;
        decode_vox(i / CG, Do, Ho, Wo, n, d_o, ho, wo);
        float xv[8][VEC], m[VEC], g[VEC];
        int64_t off[8];
#pragma unroll
        for (int k = 0; k < VEC; ++k) m[k] = -INFINITY;
        const int nt = 4 << pd;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (t >= nt) break;
            int dd = (d_o << pd) + (t >> 2), hh = 2 * ho + ((t >> 1) & 1), ww = 2 * wo + (t & 1);
            off[t] = (((int64_t)n * D + dd) * H + hh) * W + ww;
            ldv<T, VEC>(x + off[t] * C + cg * VEC, xv[t]);
#pragma unroll
            for (int k = 0; k < VEC; ++k) m[k] = fmaxf(m[k], xv[t][k]);
        }
        ldv<T, VEC>(dy + ((((int64_t)n * Do + d_o) * Ho + ho) * Wo + wo) * C + cg * VEC, g);
        bool taken[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) taken[k] = false;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (t >= nt) break;
            float r[VEC], a[VEC];
            if (add) ldv<T, VEC>(add + off[t] * add_ld + add_off + cg * VEC, a);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                float rr = 0.f;
                if (!taken[k] && xv[t][k] == m[k]) { rr = g[k]; taken[k] = true; }
                if (add) rr += a[k];
                if (relu_mask && !(xv[t][k] > 0.f)) rr = 0.f;
                r[k] = rr;
            }
            stv<T, VEC>(dx + off[t] * C + cg * VEC, r);
        }
    }
}

extern "C" int fmri_maxpool3d_2x_fwd(const void* x, void* y, int N, int D, int H, int W, int C, int dtype, int planar,
                                     fmri_stream_t stream) {
    const int pd = planar ? 0 : 1;
    if (N <= 0 || C <= 0 || D < 1 || (H & 1) || (W & 1) || (pd && (D & 1))) return FMRI_E_SHAPE;
    int vec = pick_vec(C);
    int grid = grid_for((int64_t)N * (D >> pd) * (H / 2) * (W / 2) * (C / vec));
    hipStream_t s = as_stream(stream);
    if (dtype == FMRI_F32) LAUNCH_TV(k_maxpool_fwd, float, vec, grid, 256, s, (const float*)x, (float*)y, N, D, H, W, C, pd);
    else if (dtype == FMRI_BF16) LAUNCH_TV(k_maxpool_fwd, bf16_t, vec, grid, 256, s, (const bf16_t*)x, (bf16_t*)y, N, D, H, W, C, pd);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

extern "C" int fmri_maxpool3d_2x_bwd(const void* x, const void* dy, const void* add, int add_ld, int add_off, void* dx,
                                     int N, int D, int H, int W, int C, int relu_mask, int dtype, int planar, fmri_stream_t stream) {
    const int pd = planar ? 0 : 1;
    if (N <= 0 || C <= 0 || D < 1 || (H & 1) || (W & 1) || (pd && (D & 1))) return FMRI_E_SHAPE;
    int vec = pick_vec(C);
    if (add) { while (vec > 1 && ((add_ld % vec) || (add_off % vec))) vec >>= 1; }
    int grid = grid_for((int64_t)N * (D >> pd) * (H / 2) * (W / 2) * (C / vec));
    hipStream_t s = as_stream(stream);
    if (dtype == FMRI_F32)
        LAUNCH_TV(k_maxpool_bwd, float, vec, grid, 256, s, (const float*)x, (const float*)dy, (const float*)add, add_ld,
                  add_off, (float*)dx, N, D, H, W, C, relu_mask, pd);
    else if (dtype == FMRI_BF16)
        LAUNCH_TV(k_maxpool_bwd, bf16_t, vec, grid, 256, s, (const bf16_t*)x, (const bf16_t*)dy, (const bf16_t*)add, add_ld,
                  add_off, (bf16_t*)dx, N, D, H, W, C, relu_mask, pd);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

// ------------------------------------------------------------------------------------------------ nearest up-sampling x2
// Reference: UpSampling3D(size=pool_size) at unet3d/unet.py:138.  D,H,W = low-resolution dims.
template <typename T, int VEC>
__global__ void k_upsample_fwd(const T* __restrict__ x, T* __restrict__ y, int y_ld, int y_off, int N, int D, int H, int W,
                               int C, int pd) {
    const int CG = C / VEC;
    const int64_t total = (int64_t)N * D * H * W * CG;
    const int D2 = D << pd, H2 = 2 * H, W2 = 2 * W;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cg = (int)(i % CG), n, d, h, w;
        decode_vox(i / CG, D, H, W, n, d, h, w);
        float v[VEC];
        ldv<T, VEC>(x + (i / CG) * C + cg * VEC, v);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (t >= (4 << pd)) break;
            int64_t o = (((int64_t)n * D2 + (d << pd) + (t >> 2)) * H2 + 2 * h + ((t >> 1) & 1)) * W2 + 2 * w + (t & 1);
            stv<T, VEC>(y + o * y_ld + y_off + cg * VEC, v);
        }
    }
}
template <typename T, int VEC>
__global__ void k_upsample_bwd(const T* __restrict__ dy, int dy_ld, int dy_off, const T* __restrict__ xmask,
                               T* __restrict__ dx, int N, int D, int H, int W, int C, int pd) {
    const int CG = C / VEC;
    const int64_t total = (int64_t)N * D * H * W * CG;
    const int D2 = D << pd, H2 = 2 * H, W2 = 2 * W;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int cg = (int)(i % CG), n, d, h, w;
        decode_vox(i / CG, D, H, W, n, d, h, w);
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (t >= (4 << pd)) break;
            int64_t o = (((int64_t)n * D2 + (d << pd) + (t >> 2)) * H2 + 2 * h + ((t >> 1) & 1)) * W2 + 2 * w + (t & 1);
            float g[VEC];
            ldv<T, VEC>(dy + o * dy_ld + dy_off + cg * VEC, g);
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] += g[k];
        }
        if (xmask) {
            float m[VEC];
            ldv<T, VEC>(xmask + (i / CG) * C + cg * VEC, m);
#pragma unroll
            for (int k = 0; k < VEC; ++k) if (!(m[k] > 0.f)) acc[k] = 0.f;
        }
        stv<T, VEC>(dx + (i / CG) * C + cg * VEC, acc);
    }
}

extern "C" int fmri_upsample_nearest2x_fwd(const void* x, void* y, int y_ld, int y_off, int N, int D, int H, int W, int C,
                                           int dtype, int planar, fmri_stream_t stream) {
    const int pd = planar ? 0 : 1;
    if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || y_ld < y_off + C) return FMRI_E_SHAPE;
    int vec = pick_vec(C);
    while (vec > 1 && ((y_ld % vec) || (y_off % vec))) vec >>= 1;
    int grid = grid_for((int64_t)N * D * H * W * (C / vec));
    hipStream_t s = as_stream(stream);
    if (dtype == FMRI_F32) LAUNCH_TV(k_upsample_fwd, float, vec, grid, 256, s, (const float*)x, (float*)y, y_ld, y_off, N, D, H, W, C, pd);
    else if (dtype == FMRI_BF16) LAUNCH_TV(k_upsample_fwd, bf16_t, vec, grid, 256, s, (const bf16_t*)x, (bf16_t*)y, y_ld, y_off, N, D, H, W, C, pd);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
extern "C" int fmri_upsample_nearest2x_bwd(const void* dy, int dy_ld, int dy_off, const void* xmask, void* dx, int N, int D,
                                           int H, int W, int C, int dtype, int planar, fmri_stream_t stream) {
    const int pd = planar ? 0 : 1;
    if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || dy_ld < dy_off + C) return FMRI_E_SHAPE;
    int vec = pick_vec(C);
    while (vec > 1 && ((dy_ld % vec) || (dy_off % vec))) vec >>= 1;
    int grid = grid_for((int64_t)N * D * H * W * (C / vec));
    hipStream_t s = as_stream(stream);
    if (dtype == FMRI_F32)
        LAUNCH_TV(k_upsample_bwd, float, vec, grid, 256, s, (const float*)dy, dy_ld, dy_off, (const float*)xmask, (float*)dx, N, D, H, W, C, pd);
    else if (dtype == FMRI_BF16)
        LAUNCH_TV(k_upsample_bwd, bf16_t, vec, grid, 256, s, (const bf16_t*)dy, dy_ld, dy_off, (const bf16_t*)xmask, (bf16_t*)dx, N, D, H, W, C, pd);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

// ------------------------------------------------------------------------------------------------ final 1x1x1 conv
// Reference: Conv3D(n_labels,(1,1,1)) at unet3d/unet.py:68.  AI ~ 1 flop/B: one pass over x, C small (<= 256).
// One thread per voxel, channel loop vectorised; weights broadcast from LDS.
template <typename T, int VEC>
__global__ void k_conv1x1_fwd(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                              float* __restrict__ logits, int64_t nvox, int C, int L) {
    extern __shared__ __attribute__((aligned(16))) float sw[];  // [L][C]
    for (int i = threadIdx.x; i < L * C; i += blockDim.x) sw[i] = w[i];
    __syncthreads();
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < nvox; v += (int64_t)gridDim.x * blockDim.x) {
        for (int l = 0; l < L; ++l) {
            float acc = b ? b[l] : 0.f;
            for (int c = 0; c < C; c += VEC) {
                float xv[VEC];
                ldv<T, VEC>(x + v * C + c, xv);
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc = fmaf(xv[k], sw[l * C + c + k], acc);
            }
            logits[v * L + l] = acc;
        }
    }
}
// backward: dx (masked), and block-reduced dw/db with one atomic per (block, output).
template <typename T, int VEC>
__global__ void k_conv1x1_bwd(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ dl,
                              T* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db, int64_t nvox, int C, int L,
                              int relu_mask) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // [L][C] weights, then [L][C+1] accumulators
    float* sw = sm;
    float* sacc = sm + L * C;
    for (int i = threadIdx.x; i < L * C; i += blockDim.x) sw[i] = w[i];
    for (int i = threadIdx.x; i < L * (C + 1); i += blockDim.x) sacc[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int64_t v0 = blockIdx.x * (int64_t)blockDim.x; v0 < nvox; v0 += (int64_t)gridDim.x * blockDim.x) {
        int64_t v = v0 + threadIdx.x;
        bool ok = v < nvox;
        for (int c = 0; c < C; c += VEC) {
            float xv[VEC], gx[VEC];
            if (ok) ldv<T, VEC>(x + v * C + c, xv);
            else {
#pragma unroll
                for (int k = 0; k < VEC; ++k) xv[k] = 0.f;
            }
#pragma unroll
            for (int k = 0; k < VEC; ++k) gx[k] = 0.f;
            for (int l = 0; l < L; ++l) {
                float g = ok ? dl[v * L + l] : 0.f;
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    gx[k] = fmaf(g, sw[l * C + c + k], gx[k]);
                    float p = g * xv[k];  // dw contribution: wave-reduce then one LDS atomic
                    for (int o = 32; o > 0; o >>= 1) p += __shfl_down(p, o);
                    if (lane == 0) atomicAdd(&sacc[l * (C + 1) + c + k], p);
                }
            }
            if (ok && dx) {
                if (relu_mask) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) if (!(xv[k] > 0.f)) gx[k] = 0.f;
                }
                stv<T, VEC>(dx + v * C + c, gx);
            }
        }
        for (int l = 0; l < L; ++l) {
            float g = ok ? dl[v * L + l] : 0.f;
            for (int o = 32; o > 0; o >>= 1) g += __shfl_down(g, o);
            if (lane == 0) atomicAdd(&sacc[l * (C + 1) + C], g);
        }
    }
    __syncthreads();
    const FmriDetCfg dc = g_det_cfg;
    for (int i = threadIdx.x; i < L * (C + 1); i += blockDim.x) {
        int l = i / (C + 1), c = i % (C + 1);
        if (c < C) fmri_grad_add(dc, &dw[l * C + c], sacc[i]);
        else if (db) fmri_grad_add(dc, &db[l], sacc[i]);
    }
}


// ---- fast path: LPV = C/8 lanes cooperate on one voxel (16-B loads, whole 128-B lines per voxel group), L <= 4.
template <typename T, int LPV>
__global__ void __launch_bounds__(256)
k_conv1x1_fwd_v2(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ logits,
                 int64_t nvox, int C, int L) {
    const int sub = threadIdx.x % LPV;                 // which 8-channel group of the voxel
    float wr[4][8];
#pragma unroll
    for (int l = 0; l < 4; ++l)
#pragma unroll
        for (int k = 0; k < 8; ++k) wr[l][k] = l < L ? w[l * C + sub * 8 + k] : 0.f;
    const int64_t vpb = blockDim.x / LPV;
    for (int64_t v = blockIdx.x * vpb + threadIdx.x / LPV; v < nvox + (vpb - 1); v += (int64_t)gridDim.x * vpb) {
        const bool ok = v < nvox;                      // keep whole waves in the shuffles
        float xv[8];
        if (ok) ldv<T, 8>(x + v * C + sub * 8, xv);
        else {
#pragma unroll
            for (int k = 0; k < 8; ++k) xv[k] = 0.f;
        }
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            if (l < L) {
                float a = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) a = fmaf(xv[k], wr[l][k], a);
#pragma unroll
                for (int o = LPV / 2; o > 0; o >>= 1) a += __shfl_xor(a, o);
                if (ok && sub == 0) logits[v * L + l] = a + (b ? b[l] : 0.f);
            }
        }
    }
}
template <typename T, int LPV>
__global__ void __launch_bounds__(256)
k_conv1x1_bwd_v2(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ dl, T* __restrict__ dx,
                 float* __restrict__ dw, float* __restrict__ db, int64_t nvox, int C, int L, int relu_mask) {
    // [L][C+1] sums of the workgroup in 2^-40 fixed point: 64-bit integer adds give the same bits in any order (fp32 LDS atomics from 32
    // voxel lanes per element did not: the workgroup's partial sums - and with them dw of the last layer - differed from run to run)
    extern __shared__ __attribute__((aligned(16))) unsigned long long sacc[];
    for (int i = threadIdx.x; i < L * (C + 1); i += blockDim.x) sacc[i] = 0ull;
    __syncthreads();
    const int sub = threadIdx.x % LPV;
    float wr[4][8], aw[4][8], ab[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        ab[l] = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { wr[l][k] = l < L ? w[l * C + sub * 8 + k] : 0.f; aw[l][k] = 0.f; }
    }
    const int64_t vpb = blockDim.x / LPV;
    const int64_t stride = (int64_t)gridDim.x * vpb;
    // four voxels per trip: their loads are issued together (HBM-bound kernel: one dependent load per trip left the memory pipe half empty)
    for (int64_t v0 = blockIdx.x * vpb + threadIdx.x / LPV; v0 < nvox; v0 += 4 * stride) {
        float xv[4][8];
        float gl[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t v = v0 + u * stride;
            if (v < nvox) {
                ldv<T, 8>(x + v * C + sub * 8, xv[u]);
#pragma unroll
                for (int l = 0; l < 4; ++l) gl[u][l] = l < L ? dl[v * L + l] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t v = v0 + u * stride;
            if (v >= nvox) break;
            float gx[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) gx[k] = 0.f;
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                if (l < L) {
                    const float g = gl[u][l];
                    ab[l] += g;
#pragma unroll
                    for (int k = 0; k < 8; ++k) { gx[k] = fmaf(g, wr[l][k], gx[k]); aw[l][k] = fmaf(g, xv[u][k], aw[l][k]); }
                }
            }
            if (dx) {
                if (relu_mask) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) if (!(xv[u][k] > 0.f)) gx[k] = 0.f;
                }
                stv<T, 8>(dx + v * C + sub * 8, gx);
            }
        }
    }
    for (int l = 0; l < L; ++l) {
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(&sacc[l * (C + 1) + sub * 8 + k], (unsigned long long)__float2ll_rn(aw[l][k] * FMRI_DET_SCALE));
        if (sub == 0) atomicAdd(&sacc[l * (C + 1) + C], (unsigned long long)__float2ll_rn(ab[l] * FMRI_DET_SCALE));
    }
    __syncthreads();
    const FmriDetCfg dc = g_det_cfg;
    for (int i = threadIdx.x; i < L * (C + 1); i += blockDim.x) {
        int l = i / (C + 1), c = i % (C + 1);
        const float v = (float)((double)(long long)sacc[i] * (1.0 / (double)FMRI_DET_SCALE));
        if (c < C) fmri_grad_add(dc, &dw[l * C + c], v);
        else if (db) fmri_grad_add(dc, &db[l], v);
    }
}
static int lpv_of(int C, int L) {
    if ((C % 8) || L > 4) return 0;
    int lpv = C / 8;
    if (lpv > 64 || (lpv & (lpv - 1))) return 0;
    return lpv;
}
#define LAUNCH_LPV(KERN, T, lpv, grid, sh, s, ...)                              \
    switch (lpv) {                                                              \
        case 1: KERN<T, 1><<<grid, 256, sh, s>>>(__VA_ARGS__); break;           \
        case 2: KERN<T, 2><<<grid, 256, sh, s>>>(__VA_ARGS__); break;           \
        case 4: KERN<T, 4><<<grid, 256, sh, s>>>(__VA_ARGS__); break;           \
        case 8: KERN<T, 8><<<grid, 256, sh, s>>>(__VA_ARGS__); break;           \
        case 16: KERN<T, 16><<<grid, 256, sh, s>>>(__VA_ARGS__); break;         \
        case 32: KERN<T, 32><<<grid, 256, sh, s>>>(__VA_ARGS__); break;         \
        default: KERN<T, 64><<<grid, 256, sh, s>>>(__VA_ARGS__); break;         \
    }

extern "C" int fmri_conv1x1_fwd(const void* x, const float* w, const float* b, float* logits, int64_t nvox, int C, int L,
                                int dtype, fmri_stream_t stream) {
    if (nvox <= 0 || C <= 0 || L <= 0 || (size_t)L * C * 4 > 64 * 1024) return FMRI_E_SHAPE;
    int vec = pick_vec(C);
    int grid = grid_for(nvox, 256, 256 * 8);
    hipStream_t s = as_stream(stream);
    size_t sh = (size_t)L * C * 4;
    if (int lpv = lpv_of(C, L)) {
        int g2 = grid_for(nvox * lpv, 256, 256 * 8);
        if (dtype == FMRI_F32) { LAUNCH_LPV(k_conv1x1_fwd_v2, float, lpv, g2, 0, s, (const float*)x, w, b, logits, nvox, C, L) }
        else if (dtype == FMRI_BF16) { LAUNCH_LPV(k_conv1x1_fwd_v2, bf16_t, lpv, g2, 0, s, (const bf16_t*)x, w, b, logits, nvox, C, L) }
        else return FMRI_E_DTYPE;
        FMRI_LAUNCH_CHECK();
        return FMRI_OK;
    }
    if (dtype == FMRI_F32) {
        switch (vec) {
            case 8: k_conv1x1_fwd<float, 8><<<grid, 256, sh, s>>>((const float*)x, w, b, logits, nvox, C, L); break;
            case 4: k_conv1x1_fwd<float, 4><<<grid, 256, sh, s>>>((const float*)x, w, b, logits, nvox, C, L); break;
            case 2: k_conv1x1_fwd<float, 2><<<grid, 256, sh, s>>>((const float*)x, w, b, logits, nvox, C, L); break;
            default: k_conv1x1_fwd<float, 1><<<grid, 256, sh, s>>>((const float*)x, w, b, logits, nvox, C, L); break;
        }
    } else if (dtype == FMRI_BF16) {
        switch (vec) {
            case 8: k_conv1x1_fwd<bf16_t, 8><<<grid, 256, sh, s>>>((const bf16_t*)x, w, b, logits, nvox, C, L); break;
            case 4: k_conv1x1_fwd<bf16_t, 4><<<grid, 256, sh, s>>>((const bf16_t*)x, w, b, logits, nvox, C, L); break;
            case 2: k_conv1x1_fwd<bf16_t, 2><<<grid, 256, sh, s>>>((const bf16_t*)x, w, b, logits, nvox, C, L); break;
            default: k_conv1x1_fwd<bf16_t, 1><<<grid, 256, sh, s>>>((const bf16_t*)x, w, b, logits, nvox, C, L); break;
        }
    } else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

extern "C" int fmri_conv1x1_bwd(const void* x, const float* w, const float* dlogits, void* dx, float* dw, float* db,
                                int64_t nvox, int C, int L, int relu_mask, int dtype, fmri_stream_t stream) {
    if (nvox <= 0 || C <= 0 || L <= 0 || (size_t)L * (2 * C + 1) * 4 > 64 * 1024) return FMRI_E_SHAPE;
    int vec = pick_vec(C, 4);
    int grid = grid_for(nvox, 256, 256 * 4);
    hipStream_t s = as_stream(stream);
    size_t sh = (size_t)L * (2 * C + 1) * 4;
    if (int lpv = lpv_of(C, L)) {
        // two workgroups per CU (256 CUs): every workgroup ends with L x (C + 1) LDS + global atomics and starts with an LDS clear, and the
        // streaming part wants few, long-running workgroups - measured on the benchmark's 4 x 64 x 128 x 128 x 64 tensor (ms per launch):
        // 256 workgroups 0.39, 512 0.22, 768 0.29, 1024 0.23, 2048 0.26, 4096 (the former cap) 0.31, 8192 0.35
        int g2 = grid_for(nvox * lpv, 256, 512);
        size_t sh2 = (size_t)L * (C + 1) * 8;
        if (dtype == FMRI_F32) { LAUNCH_LPV(k_conv1x1_bwd_v2, float, lpv, g2, sh2, s, (const float*)x, w, dlogits, (float*)dx, dw, db, nvox, C, L, relu_mask) }
        else if (dtype == FMRI_BF16) { LAUNCH_LPV(k_conv1x1_bwd_v2, bf16_t, lpv, g2, sh2, s, (const bf16_t*)x, w, dlogits, (bf16_t*)dx, dw, db, nvox, C, L, relu_mask) }
        else return FMRI_E_DTYPE;
        FMRI_LAUNCH_CHECK();
        return FMRI_OK;
    }
    if (dtype == FMRI_F32) {
        switch (vec) {
            case 4: k_conv1x1_bwd<float, 4><<<grid, 256, sh, s>>>((const float*)x, w, dlogits, (float*)dx, dw, db, nvox, C, L, relu_mask); break;
            case 2: k_conv1x1_bwd<float, 2><<<grid, 256, sh, s>>>((const float*)x, w, dlogits, (float*)dx, dw, db, nvox, C, L, relu_mask); break;
            default: k_conv1x1_bwd<float, 1><<<grid, 256, sh, s>>>((const float*)x, w, dlogits, (float*)dx, dw, db, nvox, C, L, relu_mask); break;
        }
    } else if (dtype == FMRI_BF16) {
        switch (vec) {
            case 4: k_conv1x1_bwd<bf16_t, 4><<<grid, 256, sh, s>>>((const bf16_t*)x, w, dlogits, (bf16_t*)dx, dw, db, nvox, C, L, relu_mask); break;
            case 2: k_conv1x1_bwd<bf16_t, 2><<<grid, 256, sh, s>>>((const bf16_t*)x, w, dlogits, (bf16_t*)dx, dw, db, nvox, C, L, relu_mask); break;
            default: k_conv1x1_bwd<bf16_t, 1><<<grid, 256, sh, s>>>((const bf16_t*)x, w, dlogits, (bf16_t*)dx, dw, db, nvox, C, L, relu_mask); break;
        }
    } else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

// ------------------------------------------------------------------------------------------------ sigmoid + Dice
// Reference: Activation('sigmoid') unet.py:69; dice_coefficient metrics.py:11-15; vod_coefficient :18-28;
// Keras 'binary_accuracy' (unet.py:81).  Block-reduce in double, one atomic per block and sum.
__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}
// VEC = 4: 16-byte loads of logits / weights / probabilities, 4-byte loads of the labels (the launcher checks alignment and n % 4); at most 512
// workgroups, each ending in nine double atomics on one 128-byte line (4 x 64x128x128 logits, per launch: scalar loads + 1,024 workgroups
// 27.6 us; this form with 128 / 256 / 512 / 1,024 workgroups 38.6 / 25.0 / 20.5 / 22.8 us - profiles/r06_dice_kernel_ab.log; the same sums).
template <int VEC>
__global__ void __launch_bounds__(256) k_sigmoid_dice_fwd(const float* __restrict__ logits, const uint8_t* __restrict__ y, const float* __restrict__ wgt,
                                   float* __restrict__ probs, double* __restrict__ sums, int64_t n) {
    double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto one = [&](float z, float t, float w, float& p_out) {
        float p = 1.f / (1.f + __expf(-z));
        p_out = p;
        {   // binary cross-entropy with Keras' clipping, and the reference focal term (metrics.py:80-87: alpha .5, gamma 2)
            const float pc = fminf(fmaxf(p, 1e-7f), 1.f - 1e-7f);
            s[7] += (double)(w * (t > 0.5f ? -__logf(pc) : -__logf(1.f - pc)));   // weight_mask * xent (metrics.py:72-76)
            s[8] += (double)(t > 0.5f ? -0.5f * (1.f - p) * (1.f - p) * __logf(p) : -0.5f * p * p * __logf(1.f - p));
        }
        s[0] += (double)(t * p);
        s[1] += (double)t;
        s[2] += (double)p;
        float tb = t > 0.5f ? 1.f : 0.f, pb = p > 0.5f ? 1.f : 0.f;
        s[3] += tb * pb;
        s[4] += tb;
        s[5] += pb;
        s[6] += (rintf(p) == t) ? 1.0 : 0.0;
    };
    const int64_t nv = n / VEC;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
        if constexpr (VEC == 4) {
            const float4 z = reinterpret_cast<const float4*>(logits)[i];
            const uchar4 t = reinterpret_cast<const uchar4*>(y)[i];
            float4 w = make_float4(1.f, 1.f, 1.f, 1.f), p;
            if (wgt) w = reinterpret_cast<const float4*>(wgt)[i];
            one(z.x, (float)t.x, w.x, p.x);
            one(z.y, (float)t.y, w.y, p.y);
            one(z.z, (float)t.z, w.z, p.z);
            one(z.w, (float)t.w, w.w, p.w);
            if (probs) reinterpret_cast<float4*>(probs)[i] = p;
        } else {
            float p;
            one(logits[i], (float)y[i], wgt ? wgt[i] : 1.f, p);
            if (probs) probs[i] = p;
        }
    }
    __shared__ double red[9][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        double r = wave_sum(s[k]);
        if (lane == 0) red[k][wv] = r;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        double r = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        const int k = threadIdx.x < 7 ? threadIdx.x : threadIdx.x + 1;            // [8] = sum xent, [9] = sum focal
        // deterministic mode: the workgroups' sums meet as 2^-20 fixed-point integers (g_isums), k_isums_finish adds them to `sums`
        if (g_det_cfg.base != nullptr) atomicAdd(&g_isums[k], (unsigned long long)__double2ll_rn(r * 1048576.0));
        else atomicAdd(&sums[k], r);
    }
    if (blockIdx.x == 0 && threadIdx.x == 9) atomicAdd(&sums[7], (double)n);
}
__global__ void k_isums_finish(double* __restrict__ sums) {
    const int k = threadIdx.x;
    if (k < 16) {
        const long long v = (long long)g_isums[k];
        if (v) sums[k] += (double)v * (1.0 / 1048576.0);
        g_isums[k] = 0ull;
    }
}
// gradient += shadow * 2^-40, shadow = 0 (see FmriDetCfg in common.h)
__global__ void k_det_finish(float* __restrict__ g, unsigned long long* __restrict__ shadow, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const long long v = (long long)shadow[i];
        if (v) {
            g[i] += (float)((double)v * (1.0 / (double)FMRI_DET_SCALE));
            shadow[i] = 0ull;
        }
    }
}
__global__ void k_sigmoid_dice_bwd(const float* __restrict__ probs, const uint8_t* __restrict__ y, const double* __restrict__ sums,
                                   float* __restrict__ dl, int64_t n, float smooth, float grad_scale) {
    const double I = sums[0], den = sums[1] + sums[2] + (double)smooth;
    const float a = (float)(2.0 / den);                       // coefficient of y
    const float c = (float)((2.0 * I + (double)smooth) / (den * den));
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float p = probs[i], t = (float)y[i];
        float dLdp = -(a * t - c);
        dl[i] = grad_scale * dLdp * p * (1.f - p);
    }
}
// general loss gradient w.r.t. the logits from the (global) sums: kind 0 dice, 1 mean binary cross-entropy, 2 dice + w*xent,
// 3 focal (alpha .5, gamma 2), 4 vod (smoothed IoU on probabilities), 5 double dice (-dice(y,p) + r*dice(1-y,p))
__global__ void k_sigmoid_loss_bwd(const float* __restrict__ probs, const uint8_t* __restrict__ y, const float* __restrict__ wgt,
                                   const double* __restrict__ sums, float* __restrict__ dl, int64_t n, int kind, float p0, float smooth,
                                   float grad_scale) {
    const double I = sums[0], Sy = sums[1], Sp = sums[2], nn = sums[7];
    const double den = Sy + Sp + smooth;
    const float a = (float)(2.0 / den), c = (float)((2.0 * I + smooth) / (den * den));
    const double U = Sy + Sp - I + smooth;
    const float vu = (float)(1.0 / U), vc = (float)((I + smooth) / (U * U));
    const double I2 = Sp - I, den2 = (nn - Sy) + Sp + smooth;
    const float a2 = (float)(2.0 / den2), c2 = (float)((2.0 * I2 + smooth) / (den2 * den2));
    const float invn = (float)(1.0 / nn);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float p = probs[i], t = (float)y[i];
        const float sg = p * (1.f - p);
        float g;                                               // dL/dlogit
        if (kind == 0) g = -(a * t - c) * sg;
        else if (kind == 1) g = (wgt ? wgt[i] : 1.f) * (p - t) * invn;
        else if (kind == 2) g = -(a * t - c) * sg + p0 * (wgt ? wgt[i] : 1.f) * (p - t) * invn;
        else if (kind == 3) {
            float dp;
            if (t > 0.5f) dp = -0.5f * (-2.f * (1.f - p) * __logf(p) + (1.f - p) * (1.f - p) / p);
            else dp = -0.5f * (2.f * p * __logf(1.f - p) - p * p / (1.f - p));
            g = dp * sg;
        } else if (kind == 4) g = -(t * vu - (1.f - t) * vc) * sg;
        else g = (-(a * t - c) + p0 * (a2 * (1.f - t) - c2)) * sg;
        dl[i] = grad_scale * g;
    }
}
extern "C" int fmri_sigmoid_loss_bwd(const float* probs, const uint8_t* y_true, const double* sums, float* dlogits, int64_t n, int kind,
                                     float param, float smooth, float grad_scale, fmri_stream_t stream) {
    if (n <= 0 || kind < 0 || kind > 5) return FMRI_E_SHAPE;
    k_sigmoid_loss_bwd<<<grid_for(n, 256, 2048), 256, 0, as_stream(stream)>>>(probs, y_true, nullptr, sums, dlogits, n, kind, param, smooth, grad_scale);
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
// ---- weighted_dice_coefficient_loss (reference metrics.py:39-55): Dice per (sample, label) over the voxel axes, smooth 1e-5, MEAN over the
// (sample, label) groups - the only loss whose values the reference's own tests pin (test/test_metrics.py:10-38).
// gsums [G = nsamples * L][3] doubles: sum y*p, sum y, sum p of group g = n * L + l (element (n, v, l) sits at (n * vox + v) * L + l)
__global__ void k_wdice_sums(const float* __restrict__ probs, const uint8_t* __restrict__ y, double* __restrict__ gsums, int64_t vox, int L) {
    const int n = blockIdx.y;
    const int64_t per = vox * L;
    const float* const pp = probs + (int64_t)n * per;
    const uint8_t* const yy = y + (int64_t)n * per;
    __shared__ double red[3][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int l = 0; l < L; ++l) {
        double a = 0, b = 0, c = 0;
        for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < vox; v += (int64_t)gridDim.x * blockDim.x) {
            const float p = pp[v * L + l], t = (float)yy[v * L + l];
            a += (double)(t * p);
            b += (double)t;
            c += (double)p;
        }
        a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
        if (lane == 0) { red[0][wv] = a; red[1][wv] = b; red[2][wv] = c; }
        __syncthreads();
        if (threadIdx.x < 3) atomicAdd(&gsums[((int64_t)n * L + l) * 3 + threadIdx.x], red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
        __syncthreads();
    }
}
// sums[10] += sum_g (2 I_g + s) / (Sy_g + Sp_g + s), sums[11] += G: additive over ranks, loss = -sums[10] / sums[11]
__global__ void k_wdice_finish(const double* __restrict__ gsums, double* __restrict__ sums, int G, float smooth) {
    double acc = 0;
    for (int g = threadIdx.x; g < G; g += blockDim.x)
        acc += (2.0 * gsums[3 * g] + (double)smooth) / (gsums[3 * g + 1] + gsums[3 * g + 2] + (double)smooth);
    acc = wave_sum(acc);
    if (threadIdx.x == 0) {
        atomicAdd(&sums[10], acc);
        atomicAdd(&sums[11], (double)G);
    }
}
// dL/dlogit of -mean_g dice_g: group g's elements get -(1 / G_total) [2 y den_g - (2 I_g + s)] / den_g^2 * p (1 - p); G_total = sums[11]
__global__ void k_wdice_bwd(const float* __restrict__ probs, const uint8_t* __restrict__ y, const double* __restrict__ gsums,
                            const double* __restrict__ sums, float* __restrict__ dl, int64_t vox, int L, float smooth, float grad_scale) {
    const int n = blockIdx.y;
    const int64_t per = vox * L, base = (int64_t)n * per;
    const double invg = 1.0 / sums[11];
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < per; i += (int64_t)gridDim.x * blockDim.x) {
        const int l = (int)(i % L);
        const double* const gs = gsums + ((int64_t)n * L + l) * 3;
        const double den = gs[1] + gs[2] + (double)smooth;
        const float a = (float)(2.0 / den * invg), c = (float)((2.0 * gs[0] + (double)smooth) / (den * den) * invg);
        const float p = probs[base + i], t = (float)y[base + i];
        dl[base + i] = grad_scale * -(a * t - c) * p * (1.f - p);
    }
}
extern "C" int fmri_weighted_dice_fwd(const float* probs, const uint8_t* y_true, double* gsums, double* sums, int nsamples, int64_t vox, int L,
                                      float smooth, fmri_stream_t stream) {
    if (!probs || !y_true || !gsums || !sums || nsamples <= 0 || vox <= 0 || L <= 0 || nsamples > 65535) return FMRI_E_SHAPE;
    hipStream_t s = as_stream(stream);
    const int G = nsamples * L;
    if (hipMemsetAsync(gsums, 0, (size_t)G * 3 * sizeof(double), s) != hipSuccess) return FMRI_E_LAUNCH;
    int bx = (int)((vox + 256 * 16 - 1) / (256 * 16));
    if (bx > 256) bx = 256;
    k_wdice_sums<<<dim3(bx, nsamples), 256, 0, s>>>(probs, y_true, gsums, vox, L);
    k_wdice_finish<<<1, 64, 0, s>>>(gsums, sums, G, smooth);
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
extern "C" int fmri_weighted_dice_bwd(const float* probs, const uint8_t* y_true, const double* gsums, const double* sums, float* dlogits,
                                      int nsamples, int64_t vox, int L, float smooth, float grad_scale, fmri_stream_t stream) {
    if (!probs || !y_true || !gsums || !sums || !dlogits || nsamples <= 0 || vox <= 0 || L <= 0 || nsamples > 65535) return FMRI_E_SHAPE;
    int bx = (int)((vox * L + 256 * 8 - 1) / (256 * 8));
    if (bx > 1024) bx = 1024;
    k_wdice_bwd<<<dim3(bx, nsamples), 256, 0, as_stream(stream)>>>(probs, y_true, gsums, sums, dlogits, vox, L, smooth, grad_scale);
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
// the same two passes with a per-voxel weight on the cross-entropy term (reference metrics.py:72-76 weighted_cross_entropy_loss,
// :89-95 dice_and_xent_mask: weight = exp(-distance_mask / sigma), computed by the caller); kinds 1 and 2 use it
extern "C" int fmri_sigmoid_dice_fwd_weighted(const float* logits, const uint8_t* y_true, const float* weight, float* probs, double* sums,
                                              int64_t n, fmri_stream_t stream) {
    if (n <= 0 || !weight) return FMRI_E_SHAPE;
    {
        const bool v4 = n % 4 == 0 && !((((uintptr_t)logits) | ((uintptr_t)probs) | ((uintptr_t)weight)) & 15) && !(((uintptr_t)y_true) & 3);
        if (v4) k_sigmoid_dice_fwd<4><<<grid_for(n / 4, 256, 512), 256, 0, as_stream(stream)>>>(logits, y_true, weight, probs, sums, n);
        else k_sigmoid_dice_fwd<1><<<grid_for(n, 256, 512), 256, 0, as_stream(stream)>>>(logits, y_true, weight, probs, sums, n);
    }
    if (h_det_on) k_isums_finish<<<1, 64, 0, as_stream(stream)>>>(sums);
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
extern "C" int fmri_sigmoid_loss_bwd_weighted(const float* probs, const uint8_t* y_true, const float* weight, const double* sums,
                                              float* dlogits, int64_t n, int kind, float param, float smooth, float grad_scale,
                                              fmri_stream_t stream) {
    if (n <= 0 || !weight || (kind != 1 && kind != 2)) return FMRI_E_SHAPE;
    k_sigmoid_loss_bwd<<<grid_for(n, 256, 2048), 256, 0, as_stream(stream)>>>(probs, y_true, weight, sums, dlogits, n, kind, param, smooth, grad_scale);
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

extern "C" int fmri_sigmoid_dice_fwd(const float* logits, const uint8_t* y_true, float* probs, double* sums, int64_t n,
                                     fmri_stream_t stream) {
    if (n <= 0) return FMRI_E_SHAPE;
    {
        const bool v4 = n % 4 == 0 && !((((uintptr_t)logits) | ((uintptr_t)probs) | ((uintptr_t)logits)) & 15) && !(((uintptr_t)y_true) & 3);
        if (v4) k_sigmoid_dice_fwd<4><<<grid_for(n / 4, 256, 512), 256, 0, as_stream(stream)>>>(logits, y_true, nullptr, probs, sums, n);
        else k_sigmoid_dice_fwd<1><<<grid_for(n, 256, 512), 256, 0, as_stream(stream)>>>(logits, y_true, nullptr, probs, sums, n);
    }
    if (h_det_on) k_isums_finish<<<1, 64, 0, as_stream(stream)>>>(sums);
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
extern "C" int fmri_sigmoid_dice_bwd(const float* probs, const uint8_t* y_true, const double* sums, float* dlogits, int64_t n,
                                     float smooth, float grad_scale, fmri_stream_t stream) {
    if (n <= 0) return FMRI_E_SHAPE;
    k_sigmoid_dice_bwd<<<grid_for(n, 256, 2048), 256, 0, as_stream(stream)>>>(probs, y_true, sums, dlogits, n, smooth, grad_scale);
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

// ------------------------------------------------------------------------------------------------ Keras Adam
// Reference: Adam(lr=initial_learning_rate) at unet3d/unet.py:85; Keras 2.2 update rule (epsilon outside the sqrt).
__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                       int64_t n, float lr_t, float b1, float b2, float eps, float gs) {
    const int64_t n4 = n >> 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 pp = ((float4*)p)[i], gg = ((const float4*)g)[i], mm = ((float4*)m)[i], vv = ((float4*)v)[i];
        float* pa = (float*)&pp; float* ga = (float*)&gg; float* ma = (float*)&mm; float* va = (float*)&vv;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float gr = ga[k] * gs;
            ma[k] = b1 * ma[k] + (1.f - b1) * gr;
            va[k] = b2 * va[k] + (1.f - b2) * gr * gr;
            pa[k] -= lr_t * ma[k] / (sqrtf(va[k]) + eps);
        }
        ((float4*)p)[i] = pp; ((float4*)m)[i] = mm; ((float4*)v)[i] = vv;
    }
    for (int64_t i = (n4 << 2) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gr = g[i] * gs;
        float mi = b1 * m[i] + (1.f - b1) * gr, vi = b2 * v[i] + (1.f - b2) * gr * gr;
        m[i] = mi; v[i] = vi;
        p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}
extern "C" int fmri_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr_t, float beta1, float beta2,
                              float eps, float grad_scale, fmri_stream_t stream) {
    if (n <= 0) return FMRI_E_SHAPE;
    if ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) return FMRI_E_ALIGN;
    k_adam<<<grid_for(n / 4 + 1, 256, 2048), 256, 0, as_stream(stream)>>>(p, g, m, v, n, lr_t, beta1, beta2, eps, grad_scale);
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

// ------------------------------------------------------------------------------------------------ weight packing
// fp32 master [27][Cout][Cin] -> compute-dtype copies: forward layout and the tap-flipped transposed dgrad layout.
template <typename T>
__global__ void __launch_bounds__(256) k_pack_weights(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd, int Cout, int Cin) {
    __shared__ T tile[64][PACK_PITCH(T)];
    pack_plain_block<T>(blockIdx.x, w, wf, wd, Cout, Cin, tile);
}
extern "C" int fmri_conv3d_pack_weights(const float* w, void* w_fwd, void* w_dgrad, int Cout, int Cin, int dtype,
                                        fmri_stream_t stream) {
    if (Cout <= 0 || Cin <= 0) return FMRI_E_SHAPE;
    const int grid = 27 * ((Cout + 63) / 64) * ((Cin + 63) / 64);
    if (dtype == FMRI_F32) k_pack_weights<float><<<grid, 256, 0, as_stream(stream)>>>(w, (float*)w_fwd, (float*)w_dgrad, Cout, Cin);
    else if (dtype == FMRI_BF16) k_pack_weights<bf16_t><<<grid, 256, 0, as_stream(stream)>>>(w, (bf16_t*)w_fwd, (bf16_t*)w_dgrad, Cout, Cin);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

// ------------------------------------------------------------------------------------------------ sliding-window tiles
// Reference: batch_iterator / get_patch_from_3d_data (prediction.py:98-114, utils/patches.py:57-72) and the
// overlap-add loop prediction.py:188-193 and the final division :210.
template <typename T>
__global__ void k_tile_gather(const float* __restrict__ vol, int X, int Y, int Z, const int32_t* __restrict__ idx, int B, int px,
                              int py, int pz, T* __restrict__ tiles) {
    const int64_t per = (int64_t)px * py * pz, total = per * B;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int b = (int)(i / per);
        int64_t r = i % per;
        int z = (int)(r % pz); r /= pz;
        int y = (int)(r % py);
        int x = (int)(r / py);
        int gx = idx[3 * b] + x, gy = idx[3 * b + 1] + y, gz = idx[3 * b + 2] + z;
        // out-of-bound corners replicate the edge (utils/patches.py:75-91, np.pad mode="edge")
        gx = min(max(gx, 0), X - 1); gy = min(max(gy, 0), Y - 1); gz = min(max(gz, 0), Z - 1);
        tiles[i] = from_f<T>(vol[((int64_t)gx * Y + gy) * Z + gz]);
    }
}
__global__ void k_tile_scatter(const float* __restrict__ pred, const int32_t* __restrict__ idx, int B, int px, int py, int pz,
                               int C, double* __restrict__ acc, int32_t* __restrict__ cnt, int X, int Y, int Z) {
    const int64_t per = (int64_t)px * py * pz, total = per * B;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int b = (int)(i / per);
        int64_t r = i % per;
        int z = (int)(r % pz); r /= pz;
        int y = (int)(r % py);
        int x = (int)(r / py);
        int gx = idx[3 * b] + x, gy = idx[3 * b + 1] + y, gz = idx[3 * b + 2] + z;
        if (gx < 0 || gy < 0 || gz < 0 || gx >= X || gy >= Y || gz >= Z) continue;
        int64_t o = ((int64_t)gx * Y + gy) * Z + gz;
        for (int c = 0; c < C; ++c) atomicAdd(&acc[o * C + c], (double)pred[i * C + c]);
        atomicAdd(&cnt[o], 1);
    }
}
__global__ void k_tile_finalize(const double* __restrict__ acc, const int32_t* __restrict__ cnt, double* __restrict__ out,
                                int32_t* __restrict__ bad, int64_t nvox, int C) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nvox; i += (int64_t)gridDim.x * blockDim.x) {
        int c = cnt[i];
        if (c <= 0) { atomicAdd(bad, 1); c = 1; }
        for (int k = 0; k < C; ++k) out[i * C + k] = acc[i * C + k] / (double)c;
    }
}
extern "C" int fmri_tile_gather(const float* vol, int X, int Y, int Z, const int32_t* idx, int B, int px, int py, int pz,
                                void* tiles, int dtype, fmri_stream_t stream) {
    if (B <= 0 || px <= 0 || py <= 0 || pz <= 0 || X <= 0 || Y <= 0 || Z <= 0) return FMRI_E_SHAPE;
    int grid = grid_for((int64_t)B * px * py * pz, 256, 4096);
    if (dtype == FMRI_F32) k_tile_gather<float><<<grid, 256, 0, as_stream(stream)>>>(vol, X, Y, Z, idx, B, px, py, pz, (float*)tiles);
    else if (dtype == FMRI_BF16) k_tile_gather<bf16_t><<<grid, 256, 0, as_stream(stream)>>>(vol, X, Y, Z, idx, B, px, py, pz, (bf16_t*)tiles);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
extern "C" int fmri_tile_scatter_accumulate(const float* pred, const int32_t* idx, int B, int px, int py, int pz, int C,
                                            double* acc, int32_t* cnt, int X, int Y, int Z, fmri_stream_t stream) {
    if (B <= 0 || px <= 0 || py <= 0 || pz <= 0 || C <= 0) return FMRI_E_SHAPE;
    k_tile_scatter<<<grid_for((int64_t)B * px * py * pz, 256, 4096), 256, 0, as_stream(stream)>>>(pred, idx, B, px, py, pz, C, acc, cnt, X, Y, Z);
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
extern "C" int fmri_tile_finalize(const double* acc, const int32_t* cnt, double* out, int32_t* bad, int64_t nvox, int C,
                                  fmri_stream_t stream) {
    if (nvox <= 0 || C <= 0) return FMRI_E_SHAPE;
    k_tile_finalize<<<grid_for(nvox, 256, 4096), 256, 0, as_stream(stream)>>>(acc, cnt, out, bad, nvox, C);
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

// ------------------------------------------------------------------------------------------------ casts
template <typename S, typename Dd>
__global__ void k_cast(const S* __restrict__ s, Dd* __restrict__ d, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        d[i] = from_f<Dd>(to_f<S>(s[i]));
}
extern "C" int fmri_cast(const void* src, int sd, void* dst, int dd, int64_t n, fmri_stream_t stream) {
    if (n <= 0) return FMRI_E_SHAPE;
    int grid = grid_for(n, 256, 4096);
    hipStream_t s = as_stream(stream);
    if (sd == FMRI_F32 && dd == FMRI_BF16) k_cast<float, bf16_t><<<grid, 256, 0, s>>>((const float*)src, (bf16_t*)dst, n);
    else if (sd == FMRI_BF16 && dd == FMRI_F32) k_cast<bf16_t, float><<<grid, 256, 0, s>>>((const bf16_t*)src, (float*)dst, n);
    else if (sd == FMRI_F32 && dd == FMRI_F32) k_cast<float, float><<<grid, 256, 0, s>>>((const float*)src, (float*)dst, n);
    else if (sd == FMRI_BF16 && dd == FMRI_BF16) k_cast<bf16_t, bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)src, (bf16_t*)dst, n);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

// ---- deterministic mode (common.h, FmriDetCfg)
int fmri_det_set_mfma(const FmriDetCfg&);
int fmri_det_set_first(const FmriDetCfg&);
int fmri_det_set_generic(const FmriDetCfg&);
int fmri_det_set_direct(const FmriDetCfg&);
extern "C" int fmri_set_deterministic(float* grad_base, void* shadow_i64, int64_t n) {
    if ((grad_base == nullptr) != (shadow_i64 == nullptr) || (grad_base && n <= 0)) return FMRI_E_SHAPE;
    const FmriDetCfg c{grad_base, (unsigned long long*)shadow_i64, grad_base ? (long long)n : 0};
    int rc = fmri_det_set_pointwise(c);
    if (!rc) rc = fmri_det_set_mfma(c);
    if (!rc) rc = fmri_det_set_first(c);
    if (!rc) rc = fmri_det_set_generic(c);
    if (!rc) rc = fmri_det_set_direct(c);
    return rc;
}
extern "C" int fmri_deterministic_finish(float* grad_base, void* shadow_i64, int64_t n, fmri_stream_t stream) {
    if (!grad_base || !shadow_i64 || n <= 0) return FMRI_E_SHAPE;
    k_det_finish<<<grid_for(n, 256, 2048), 256, 0, as_stream(stream)>>>(grad_base, (unsigned long long*)shadow_i64, n);
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
