// Direct (one thread per output element) convolutions for the shapes the tiled kernels do not cover: kernel 1x1x1 with any
// Cin->Cout and kernel 3x3x3 with stride 2, TensorFlow 'same' padding (even extent: 0 before / 1 after; odd: 1 / 1).
// Used by the Isensee topology (reference fetal_net/model/unet3d/isensee2017.py:51 stride-2 blocks, :95-98 1x1x1 localisation,
// :66 segmentation heads).  Correctness-first VALU kernels, fp32 accumulate.
#include "common.h"

FMRI_DET_TU(direct)

namespace {

struct Geo {
    int N, D, H, W, Do, Ho, Wo, Cin, Cout, k, s, pd, ph, pw;   // p* = zeros in front of each axis
    int planar;       // 2-D slices stacked along D: only the centre kd plane of the filter exists, D is neither padded nor strided
};

__device__ __forceinline__ void out_coords(int64_t v, const Geo& g, int& n, int& d, int& h, int& w) {
    w = (int)(v % g.Wo); v /= g.Wo;
    h = (int)(v % g.Ho); v /= g.Ho;
    d = (int)(v % g.Do);
    n = (int)(v / g.Do);
}

template <typename T>
__global__ void k_direct_fwd(const T* __restrict__ x, const T* __restrict__ wt, const float* __restrict__ bias, T* __restrict__ y, Geo g,
                             int act, float alpha) {
    const int64_t total = (int64_t)g.N * g.Do * g.Ho * g.Wo * g.Cout;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % g.Cout);
        int n, d, h, w;
        out_coords(i / g.Cout, g, n, d, h, w);
        float acc = bias ? bias[co] : 0.f;
        for (int kd = 0; kd < g.k; ++kd) {
            if (g.planar && kd != g.k / 2) continue;
            const int id = g.planar ? d : d * g.s + kd - g.pd;
            if ((unsigned)id >= (unsigned)g.D) continue;
            for (int kh = 0; kh < g.k; ++kh) {
                const int ih = h * g.s + kh - g.ph;
                if ((unsigned)ih >= (unsigned)g.H) continue;
                for (int kw = 0; kw < g.k; ++kw) {
                    const int iw = w * g.s + kw - g.pw;
                    if ((unsigned)iw >= (unsigned)g.W) continue;
                    const T* xp = x + ((((int64_t)n * g.D + id) * g.H + ih) * g.W + iw) * g.Cin;
                    const T* wp = wt + ((int64_t)((kd * g.k + kh) * g.k + kw) * g.Cout + co) * g.Cin;
                    for (int ci = 0; ci < g.Cin; ++ci) acc = fmaf(to_f<T>(xp[ci]), to_f<T>(wp[ci]), acc);
                }
            }
        }
        if (act == FMRI_ACT_RELU) acc = fmaxf(acc, 0.f);
        else if (act == FMRI_ACT_LEAKY) acc = acc > 0.f ? acc : alpha * acc;
        y[i] = from_f<T>(acc);
    }
}

// dx[n,i,ci] = sum_{taps, o : o*s + tap - p = i} sum_co dy[o][co] * w[tap][co][ci]
template <typename T>
__global__ void k_direct_dgrad(const T* __restrict__ dy, const T* __restrict__ wt, T* __restrict__ dx, Geo g) {
    const int64_t total = (int64_t)g.N * g.D * g.H * g.W * g.Cin;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % g.Cin);
        int64_t v = i / g.Cin;
        const int iw = (int)(v % g.W); v /= g.W;
        const int ih = (int)(v % g.H); v /= g.H;
        const int id = (int)(v % g.D);
        const int n = (int)(v / g.D);
        float acc = 0.f;
        for (int kd = 0; kd < g.k; ++kd) {
            if (g.planar && kd != g.k / 2) continue;
            const int td = id + g.pd - kd;
            if (!g.planar && (td < 0 || (td % g.s) || td / g.s >= g.Do)) continue;
            const int od = g.planar ? id : td / g.s;
            for (int kh = 0; kh < g.k; ++kh) {
                const int th = ih + g.ph - kh;
                if (th < 0 || (th % g.s) || th / g.s >= g.Ho) continue;
                for (int kw = 0; kw < g.k; ++kw) {
                    const int tw = iw + g.pw - kw;
                    if (tw < 0 || (tw % g.s) || tw / g.s >= g.Wo) continue;
                    const T* gp = dy + ((((int64_t)n * g.Do + od) * g.Ho + th / g.s) * g.Wo + tw / g.s) * g.Cout;
                    const T* wp = wt + (int64_t)((kd * g.k + kh) * g.k + kw) * g.Cout * g.Cin + ci;
                    for (int co = 0; co < g.Cout; ++co) acc = fmaf(to_f<T>(gp[co]), to_f<T>(wp[(int64_t)co * g.Cin]), acc);
                }
            }
        }
        dx[i] = from_f<T>(acc);
    }
}

// dw[tap][co][ci] += sum_o dy[o][co] * x[o*s+tap-p][ci] ; db[co] += sum_o dy[o][co].  Block = (tap, co, split); threads over ci.
template <typename T>
__global__ void k_direct_wgrad(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ dw, float* __restrict__ db, Geo g,
                               int nsplit) {
    const int tap = blockIdx.x, co = blockIdx.y, sp = blockIdx.z;
    const int kd = tap / (g.k * g.k), kh = (tap / g.k) % g.k, kw = tap % g.k;
    if (g.planar && kd != g.k / 2) return;                      // dead filter planes of the 2-D slices: their gradient stays zero
    const int tap_b = g.planar ? (g.k / 2) * g.k * g.k : 0;     // the tap whose workgroups also sum the bias gradient
    const int64_t nout = (int64_t)g.N * g.Do * g.Ho * g.Wo;
    const int64_t v0 = nout * sp / nsplit, v1 = nout * (sp + 1) / nsplit;
    float bsum = 0.f;
    for (int ci0 = 0; ci0 < g.Cin; ci0 += blockDim.x) {
        const int ci = ci0 + threadIdx.x;
        float acc = 0.f;
        for (int64_t v = v0; v < v1; ++v) {
            int n, d, h, w;
            out_coords(v, g, n, d, h, w);
            const float gv = to_f<T>(dy[v * g.Cout + co]);
            if (ci0 == 0 && threadIdx.x == 0 && tap == tap_b) bsum += gv;
            const int id = g.planar ? d : d * g.s + kd - g.pd, ih = h * g.s + kh - g.ph, iw = w * g.s + kw - g.pw;
            if ((unsigned)id >= (unsigned)g.D || (unsigned)ih >= (unsigned)g.H || (unsigned)iw >= (unsigned)g.W) continue;
            if (ci < g.Cin) acc = fmaf(gv, to_f<T>(x[((((int64_t)n * g.D + id) * g.H + ih) * g.W + iw) * g.Cin + ci]), acc);
        }
        if (ci < g.Cin) fmri_grad_add(g_det_cfg, &dw[((int64_t)tap * g.Cout + co) * g.Cin + ci], acc);
    }
    if (db && threadIdx.x == 0 && tap == tap_b) fmri_grad_add(g_det_cfg, &db[co], bsum);
}

template <typename T>
__global__ void k_add(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = from_f<T>(to_f<T>(a[i]) + to_f<T>(b[i]));
}
// y[v][c] = x[v][c] * scale[n(v)][c]   (SpatialDropout3D: whole channels of a sample are dropped; scale = 0 or 1/(1-p))
template <typename T>
__global__ void k_channel_scale(const T* __restrict__ x, const float* __restrict__ scale, T* __restrict__ y, int64_t V, int C, int64_t total) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t n = (i / C) / V;
        y[i] = from_f<T>(to_f<T>(x[i]) * scale[n * C + c]);
    }
}

// dst[v][c] (+)= src[v*ld + off + c]: channel slice of a wider tensor (split of a concat gradient, gradient fan-in)
template <typename T>
__global__ void k_slice(const T* __restrict__ src, int ld, int off, T* __restrict__ dst, int C, int64_t total, int accumulate) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t v = i / C;
        float val = to_f<T>(src[v * ld + off + c]);
        if (accumulate) val += to_f<T>(dst[i]);
        dst[i] = from_f<T>(val);
    }
}

// dx = dy * act'(y)  (ReLU / LeakyReLU derivative from the stored post-activation tensor)
template <typename T>
__global__ void k_act_bwd(const T* __restrict__ y, const T* __restrict__ dy, T* __restrict__ dx, int act, float alpha, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float d = to_f<T>(dy[i]);
        const float yy = to_f<T>(y[i]);
        if (act == FMRI_ACT_RELU) d = yy > 0.f ? d : 0.f;
        else if (act == FMRI_ACT_LEAKY) d = yy > 0.f ? d : alpha * d;
        dx[i] = from_f<T>(d);
    }
}

bool make_geo(Geo& g, int N, int D, int H, int W, int Cin, int Cout, int k, int s, int planar) {
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (k != 1 && k != 3) || (s != 1 && s != 2)) return false;
    auto pad_before = [&](int n) { int out = (n + s - 1) / s; int tot = (out - 1) * s + k - n; if (tot < 0) tot = 0; return tot / 2; };
    g.N = N; g.D = D; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.k = k; g.s = s;
    g.Do = planar ? D : (D + s - 1) / s; g.Ho = (H + s - 1) / s; g.Wo = (W + s - 1) / s;
    g.pd = planar ? 0 : pad_before(D); g.ph = pad_before(H); g.pw = pad_before(W);
    g.planar = planar ? 1 : 0;
    return true;
}

}  // namespace

static int direct_fwd(const void* x, const void* w, const float* bias, void* y, int N, int D, int H, int W, int Cin, int Cout, int ksize,
                      int stride, int act, float alpha, int dtype, int planar, fmri_stream_t stream) {
    Geo g;
    if (!make_geo(g, N, D, H, W, Cin, Cout, ksize, stride, planar)) return FMRI_E_SHAPE;
    const int grid = grid_for((int64_t)N * g.Do * g.Ho * g.Wo * Cout, 256, 16384);
    hipStream_t s = as_stream(stream);
    if (dtype == FMRI_F32) k_direct_fwd<float><<<grid, 256, 0, s>>>((const float*)x, (const float*)w, bias, (float*)y, g, act, alpha);
    else if (dtype == FMRI_BF16) k_direct_fwd<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)x, (const bf16_t*)w, bias, (bf16_t*)y, g, act, alpha);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

static int direct_bwd(const void* x, const void* w, const void* dy, void* dx, float* dw, float* db, int N, int D, int H, int W, int Cin,
                      int Cout, int ksize, int stride, int dtype, int planar, fmri_stream_t stream) {
    Geo g;
    if (!make_geo(g, N, D, H, W, Cin, Cout, ksize, stride, planar)) return FMRI_E_SHAPE;
    hipStream_t s = as_stream(stream);
    const int grid = grid_for((int64_t)N * D * H * W * Cin, 256, 16384);
    const int nsplit = 32;
    dim3 gw(ksize * ksize * ksize, Cout, nsplit);
    const int bt = Cin >= 256 ? 256 : (Cin >= 128 ? 128 : 64);
    if (dtype == FMRI_F32) {
        if (dx) k_direct_dgrad<float><<<grid, 256, 0, s>>>((const float*)dy, (const float*)w, (float*)dx, g);
        if (dw) k_direct_wgrad<float><<<gw, bt, 0, s>>>((const float*)x, (const float*)dy, dw, db, g, nsplit);
    } else if (dtype == FMRI_BF16) {
        if (dx) k_direct_dgrad<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)dy, (const bf16_t*)w, (bf16_t*)dx, g);
        if (dw) k_direct_wgrad<bf16_t><<<gw, bt, 0, s>>>((const bf16_t*)x, (const bf16_t*)dy, dw, db, g, nsplit);
    } else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

extern "C" int fmri_conv3d_direct_fwd(const void* x, const void* w, const float* bias, void* y, int N, int D, int H, int W, int Cin,
                                      int Cout, int ksize, int stride, int act, float alpha, int dtype, fmri_stream_t stream) {
    return direct_fwd(x, w, bias, y, N, D, H, W, Cin, Cout, ksize, stride, act, alpha, dtype, 0, stream);
}
extern "C" int fmri_conv3d_direct_bwd(const void* x, const void* w, const void* dy, void* dx, float* dw, float* db, int N, int D, int H,
                                      int W, int Cin, int Cout, int ksize, int stride, int dtype, fmri_stream_t stream) {
    return direct_bwd(x, w, dy, dx, dw, db, N, D, H, W, Cin, Cout, ksize, stride, dtype, 0, stream);
}
// 2-D twins (reference model/unet/isensee.py:49 strides=(2,2), :96 kernel=(1,1), :59 heads): S slices stacked along D, untouched by the
// stride; the filter image keeps the 27-tap (or 1-tap) layout with the 2-D kernel in its centre kd plane
extern "C" int fmri_conv2d_direct_fwd(const void* x, const void* w, const float* bias, void* y, int S, int H, int W, int Cin, int Cout,
                                      int ksize, int stride, int act, float alpha, int dtype, fmri_stream_t stream) {
    return direct_fwd(x, w, bias, y, 1, S, H, W, Cin, Cout, ksize, stride, act, alpha, dtype, 1, stream);
}
extern "C" int fmri_conv2d_direct_bwd(const void* x, const void* w, const void* dy, void* dx, float* dw, float* db, int S, int H, int W,
                                      int Cin, int Cout, int ksize, int stride, int dtype, fmri_stream_t stream) {
    return direct_bwd(x, w, dy, dx, dw, db, 1, S, H, W, Cin, Cout, ksize, stride, dtype, 1, stream);
}

extern "C" int fmri_add(const void* a, const void* b, void* y, int64_t n, int dtype, fmri_stream_t stream) {
    if (n <= 0) return FMRI_E_SHAPE;
    const int grid = grid_for(n, 256, 4096);
    if (dtype == FMRI_F32) k_add<float><<<grid, 256, 0, as_stream(stream)>>>((const float*)a, (const float*)b, (float*)y, n);
    else if (dtype == FMRI_BF16) k_add<bf16_t><<<grid, 256, 0, as_stream(stream)>>>((const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y, n);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

extern "C" int fmri_channel_scale(const void* x, const float* scale, void* y, int N, int64_t V, int C, int dtype, fmri_stream_t stream) {
    if (N <= 0 || V <= 0 || C <= 0) return FMRI_E_SHAPE;
    const int64_t total = (int64_t)N * V * C;
    const int grid = grid_for(total, 256, 4096);
    if (dtype == FMRI_F32) k_channel_scale<float><<<grid, 256, 0, as_stream(stream)>>>((const float*)x, scale, (float*)y, V, C, total);
    else if (dtype == FMRI_BF16) k_channel_scale<bf16_t><<<grid, 256, 0, as_stream(stream)>>>((const bf16_t*)x, scale, (bf16_t*)y, V, C, total);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

extern "C" int fmri_slice_channels(const void* src, int ld, int off, void* dst, int C, int64_t nvox, int accumulate, int dtype,
                                   fmri_stream_t stream) {
    if (nvox <= 0 || C <= 0 || ld < off + C) return FMRI_E_SHAPE;
    const int64_t total = nvox * C;
    const int grid = grid_for(total, 256, 4096);
    if (dtype == FMRI_F32) k_slice<float><<<grid, 256, 0, as_stream(stream)>>>((const float*)src, ld, off, (float*)dst, C, total, accumulate);
    else if (dtype == FMRI_BF16) k_slice<bf16_t><<<grid, 256, 0, as_stream(stream)>>>((const bf16_t*)src, ld, off, (bf16_t*)dst, C, total, accumulate);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

extern "C" int fmri_act_bwd(const void* y, const void* dy, void* dx, int act, float alpha, int64_t n, int dtype, fmri_stream_t stream) {
    if (n <= 0) return FMRI_E_SHAPE;
    const int grid = grid_for(n, 256, 4096);
    if (dtype == FMRI_F32) k_act_bwd<float><<<grid, 256, 0, as_stream(stream)>>>((const float*)y, (const float*)dy, (float*)dx, act, alpha, n);
    else if (dtype == FMRI_BF16) k_act_bwd<bf16_t><<<grid, 256, 0, as_stream(stream)>>>((const bf16_t*)y, (const bf16_t*)dy, (bf16_t*)dx, act, alpha, n);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
