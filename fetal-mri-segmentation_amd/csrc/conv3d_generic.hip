// Generic 3x3x3 'same' Conv3D (forward / input-grad via flipped weights / weight-grad) for ANY channel count and
// fp32 or bf16 storage, plain fp32 VALU arithmetic.  This is the parity-mode path (fp32, bit-faithful fmaf chains) and
// the fallback for shapes the MFMA kernels do not cover (Cin = 1 first layer, base-8 configs, odd spatial sizes).
// Reference ops: Conv3D / Conv3DBackpropFilterV2 emitted for create_convolution_block (unet3d/unet.py:89-115).
#include "common.h"

FMRI_DET_TU(generic)

namespace {

constexpr int TD = 2, TH = 8, TW = 16;            // output tile = 256 voxels = one thread each
constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;
constexpr int HVOX = HD * HH * HW;                // 720 halo voxels
constexpr int CK = 8;                             // input channels staged per pass
constexpr int COB = 16;                           // output channels per block

struct Src {
    const void* p0; const void* p1; int C0, C1, up0, dsh, planar;   // dsh: the fused x2 also doubles D (3-D) or not (2-D slices)
};

// value of concat channel c at full-res voxel (n,d,h,w); zero outside the volume ('same' zero padding)
template <typename T>
__device__ __forceinline__ float load_in(const Src& s, int n, int d, int h, int w, int c, int D, int H, int W) {
    if ((unsigned)d >= (unsigned)D || (unsigned)h >= (unsigned)H || (unsigned)w >= (unsigned)W) return 0.f;
    if (c < s.C0) {
        if (s.up0) {
            int64_t o = ((((int64_t)n * (D >> s.dsh) + (d >> s.dsh)) * (H >> 1) + (h >> 1)) * (W >> 1) + (w >> 1)) * s.C0 + c;
            return to_f<T>(((const T*)s.p0)[o]);
        }
        return to_f<T>(((const T*)s.p0)[((((int64_t)n * D + d) * H + h) * W + w) * s.C0 + c]);
    }
    return to_f<T>(((const T*)s.p1)[((((int64_t)n * D + d) * H + h) * W + w) * s.C1 + (c - s.C0)]);
}

__device__ __forceinline__ void tile_origin(int tile, int D, int H, int W, int& n, int& d0, int& h0, int& w0) {
    const int tw = (W + TW - 1) / TW, th = (H + TH - 1) / TH, td = (D + TD - 1) / TD;
    w0 = (tile % tw) * TW; tile /= tw;
    h0 = (tile % th) * TH; tile /= th;
    d0 = (tile % td) * TD;
    n = tile / td;
}

template <typename T>
__global__ void __launch_bounds__(256)
k_conv_fwd_generic(Src s, const T* __restrict__ wt, const float* __restrict__ bias, const T* __restrict__ mask,
                   T* __restrict__ y, int N, int D, int H, int W, int Cout, int act, float alpha) {
    __shared__ float sx[CK][HVOX];          // 23,040 B
    __shared__ float sw[27][CK][COB];       // 13,824 B
    const int Cin = s.C0 + s.C1;
    int n, d0, h0, w0;
    tile_origin(blockIdx.x, D, H, W, n, d0, h0, w0);
    const int co0 = blockIdx.y * COB;
    const int t = threadIdx.x;
    const int lw = t & 15, lh = (t >> 4) & 7, ld = t >> 7;
    float acc[COB];
#pragma unroll
    for (int k = 0; k < COB; ++k) acc[k] = 0.f;
    for (int c0 = 0; c0 < Cin; c0 += CK) {
        for (int i = t; i < CK * HVOX; i += 256) {
            int ci = i / HVOX, hv = i % HVOX;
            int hw_ = hv % HW, hh_ = (hv / HW) % HH, hd_ = hv / (HW * HH);
            float v = 0.f;
            if (c0 + ci < Cin) v = load_in<T>(s, n, d0 + hd_ - 1, h0 + hh_ - 1, w0 + hw_ - 1, c0 + ci, D, H, W);
            sx[ci][hv] = v;
        }
        for (int i = t; i < 27 * CK * COB; i += 256) {
            int co = i % COB, ci = (i / COB) % CK, tap = i / (COB * CK);
            float v = 0.f;
            if (c0 + ci < Cin && co0 + co < Cout) v = to_f<T>(wt[((int64_t)tap * Cout + co0 + co) * Cin + c0 + ci]);
            sw[tap][ci][co] = v;
        }
        __syncthreads();
#pragma unroll 1
        for (int tap = s.planar ? 9 : 0; tap < (s.planar ? 18 : 27); ++tap) {   // planar (2-D slices): centre kd plane only
            const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
            const int hv = ((ld + kd) * HH + (lh + kh)) * HW + lw + kw;
#pragma unroll
            for (int ci = 0; ci < CK; ++ci) {
                const float xv = sx[ci][hv];
#pragma unroll
                for (int co = 0; co < COB; ++co) acc[co] = fmaf(xv, sw[tap][ci][co], acc[co]);
            }
        }
        __syncthreads();
    }
    const int d = d0 + ld, h = h0 + lh, w = w0 + lw;
    if (d < D && h < H && w < W) {
        const int64_t vo = (((int64_t)n * D + d) * H + h) * W + w;
        for (int co = 0; co < COB && co0 + co < Cout; ++co) {
            float v = acc[co] + (bias ? bias[co0 + co] : 0.f);
            if (act == FMRI_ACT_RELU) v = fmaxf(v, 0.f);
            else if (act == FMRI_ACT_LEAKY) v = v > 0.f ? v : alpha * v;
            if (mask && !(to_f<T>(mask[vo * Cout + co0 + co]) > 0.f)) v = 0.f;
            y[vo * Cout + co0 + co] = from_f<T>(v);
        }
    }
}

// weight gradient: each block walks a strided set of voxel tiles for one (Cin chunk, Cout block), keeps its
// 27*COB*CK partial sums in registers and issues one atomic per output at the end.
constexpr int WG_OUT = 27 * COB * CK;                    // 3456 outputs per (chunk, block)
constexpr int WG_PER_T = (WG_OUT + 255) / 256;           // 14
template <typename T>
__global__ void __launch_bounds__(256)
k_conv_wgrad_generic(Src s, const T* __restrict__ dy, float* __restrict__ dw, float* __restrict__ db, int N, int D, int H, int W,
                     int Cout, int ntiles) {
    __shared__ float sx[CK][HVOX + 1];
    __shared__ float sg[COB][TD * TH * TW + 1];
    const int Cin = s.C0 + s.C1;
    const int c0 = blockIdx.y * CK, co0 = blockIdx.z * COB;
    const int t = threadIdx.x;
    float acc[WG_PER_T];
#pragma unroll
    for (int j = 0; j < WG_PER_T; ++j) acc[j] = 0.f;
    float bacc = 0.f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int n, d0, h0, w0;
        tile_origin(tile, D, H, W, n, d0, h0, w0);
        for (int i = t; i < CK * HVOX; i += 256) {
            int ci = i / HVOX, hv = i % HVOX;
            int hw_ = hv % HW, hh_ = (hv / HW) % HH, hd_ = hv / (HW * HH);
            float v = 0.f;
            if (c0 + ci < Cin) v = load_in<T>(s, n, d0 + hd_ - 1, h0 + hh_ - 1, w0 + hw_ - 1, c0 + ci, D, H, W);
            sx[ci][hv] = v;
        }
        for (int i = t; i < COB * 256; i += 256) {
            int co = i % COB, v = i / COB;
            int lw = v & 15, lh = (v >> 4) & 7, ld = v >> 7;
            int d = d0 + ld, h = h0 + lh, w = w0 + lw;
            float g = 0.f;
            if (d < D && h < H && w < W && co0 + co < Cout)
                g = to_f<T>(dy[((((int64_t)n * D + d) * H + h) * W + w) * Cout + co0 + co]);
            sg[co][v] = g;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < WG_PER_T; ++j) {
            const int o = t + 256 * j;
            if (o < WG_OUT) {
                const int ci = o % CK, co = (o / CK) % COB, tap = o / (CK * COB);
                const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
                if (s.planar && kd != 1) continue;
                float a = acc[j];
                for (int v = 0; v < 256; ++v) {
                    const int lw = v & 15, lh = (v >> 4) & 7, ld = v >> 7;
                    a = fmaf(sx[ci][((ld + kd) * HH + lh + kh) * HW + lw + kw], sg[co][v], a);
                }
                acc[j] = a;
            }
        }
        if (db && blockIdx.y == 0 && t < COB) {
            float b = 0.f;
            for (int v = 0; v < 256; ++v) b += sg[t][v];
            bacc += b;
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < WG_PER_T; ++j) {
        const int o = t + 256 * j;
        if (o < WG_OUT) {
            const int ci = o % CK, co = (o / CK) % COB, tap = o / (CK * COB);
            if (c0 + ci < Cin && co0 + co < Cout) fmri_grad_add(g_det_cfg, &dw[((int64_t)tap * Cout + co0 + co) * Cin + c0 + ci], acc[j]);
        }
    }
    if (db && blockIdx.y == 0 && t < COB && co0 + t < Cout) fmri_grad_add(g_det_cfg, &db[co0 + t], bacc);
}

int ntiles_of(int N, int D, int H, int W) {
    return N * ((D + TD - 1) / TD) * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
}

}  // namespace

// internal entry points used by the dispatcher in conv3d_api.hip
int conv3d_fwd_generic(const void* src0, int C0, int up0, int planar, const void* src1, int C1, const void* w, const float* bias,
                       const void* mask, void* y, int N, int D, int H, int W, int Cout, int act, float alpha, int dtype,
                       hipStream_t s) {
    Src sr{src0, src1, C0, C1, up0, planar ? 0 : 1, planar};
    dim3 grid(ntiles_of(N, D, H, W), (Cout + COB - 1) / COB);
    if (dtype == FMRI_F32)
        k_conv_fwd_generic<float><<<grid, 256, 0, s>>>(sr, (const float*)w, bias, (const float*)mask, (float*)y, N, D, H, W, Cout, act, alpha);
    else if (dtype == FMRI_BF16)
        k_conv_fwd_generic<bf16_t><<<grid, 256, 0, s>>>(sr, (const bf16_t*)w, bias, (const bf16_t*)mask, (bf16_t*)y, N, D, H, W, Cout, act, alpha);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

int conv3d_wgrad_generic(const void* src0, int C0, int up0, int planar, const void* src1, int C1, const void* dy, float* dw, float* db,
                         int N, int D, int H, int W, int Cout, int dtype, hipStream_t s) {
    Src sr{src0, src1, C0, C1, up0, planar ? 0 : 1, planar};
    const int Cin = C0 + C1;
    const int nt = ntiles_of(N, D, H, W);
    const int cy = (Cin + CK - 1) / CK, cz = (Cout + COB - 1) / COB;
    int gx = 2048 / (cy * cz);
    if (gx < 1) gx = 1;
    if (gx > nt) gx = nt;
    dim3 grid(gx, cy, cz);
    if (dtype == FMRI_F32)
        k_conv_wgrad_generic<float><<<grid, 256, 0, s>>>(sr, (const float*)dy, dw, db, N, D, H, W, Cout, nt);
    else if (dtype == FMRI_BF16)
        k_conv_wgrad_generic<bf16_t><<<grid, 256, 0, s>>>(sr, (const bf16_t*)dy, dw, db, N, D, H, W, Cout, nt);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
