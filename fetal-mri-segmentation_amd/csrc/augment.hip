// Device-side patch sampler and intensity augmentation: the reference builds every training patch on the host
// (fetal_net/generator.py:246-328 -> augment.py:222-377 -> utils/utils.py:100-113, scipy map_coordinates); here the volumes stay in
// HBM and a patch is one gather launch plus a few elementwise passes.  All of these are HBM-bound byte movers: one thread per output
// voxel, z (the contiguous axis of the reference's (X,Y,Z) volumes) fastest, coordinates in fp64 so that nearest-neighbour label
// sampling rounds exactly like the fp64 host code.
#include "common.h"

namespace {

struct Affine34 { double a[12]; };     // row-major 3x4: source = A . (i, j, k, 1)

template <typename T> __device__ __forceinline__ float ld_any(const T* p, int64_t i);
template <> __device__ __forceinline__ float ld_any<float>(const float* p, int64_t i) { return p[i]; }
template <> __device__ __forceinline__ float ld_any<uint8_t>(const uint8_t* p, int64_t i) { return (float)p[i]; }
template <typename T> __device__ __forceinline__ void st_any(T* p, int64_t i, float v);
template <> __device__ __forceinline__ void st_any<float>(float* p, int64_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void st_any<bf16_t>(bf16_t* p, int64_t i, float v) { p[i] = f2bf(v); }
template <> __device__ __forceinline__ void st_any<uint8_t>(uint8_t* p, int64_t i, float v) { p[i] = (uint8_t)v; }

// scipy.ndimage.map_coordinates(mode='constant'): a point outside [0, n-1] on any axis takes cval (no interpolation beyond the
// edges); inside, order 0 picks floor(c + 0.5) and order 1 is trilinear on the enclosing cell.
constexpr int AFF_MAXB = 16;
struct AffineB {                        // per patch of a launch: the volume, its extent, the patch corner, the outside value, the affine
    const void* vol[AFF_MAXB];
    int X[AFF_MAXB], Y[AFF_MAXB], Z[AFF_MAXB], x0[AFF_MAXB], y0[AFF_MAXB], z0[AFF_MAXB];
    float cval[AFF_MAXB];
    Affine34 A[AFF_MAXB];
};
template <typename TI, typename TO>
__global__ void k_affine_sample(AffineB P, int nx, int ny, int nz, int order, TO* __restrict__ out, int ld, int64_t out_stride) {
    const int pb = blockIdx.y;
    const TI* __restrict__ vol = static_cast<const TI*>(P.vol[pb]);
    const int X = P.X[pb], Y = P.Y[pb], Z = P.Z[pb], x0 = P.x0[pb], y0 = P.y0[pb], z0 = P.z0[pb];
    const float cval = P.cval[pb];
    const Affine34& A = P.A[pb];
    out += pb * out_stride;
    const int64_t total = (int64_t)nx * ny * nz;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(t % nz);
        const int64_t q = t / nz;
        const int j = (int)(q % ny), i = (int)(q / ny);
        const double pi = x0 + i, pj = y0 + j, pk = z0 + k;
        const double cx = A.a[0] * pi + A.a[1] * pj + A.a[2] * pk + A.a[3];
        const double cy = A.a[4] * pi + A.a[5] * pj + A.a[6] * pk + A.a[7];
        const double cz = A.a[8] * pi + A.a[9] * pj + A.a[10] * pk + A.a[11];
        float v = cval;
        if (cx >= 0.0 && cx <= (double)(X - 1) && cy >= 0.0 && cy <= (double)(Y - 1) && cz >= 0.0 && cz <= (double)(Z - 1)) {
            if (order == 0) {
                const int ix = (int)floor(cx + 0.5), iy = (int)floor(cy + 0.5), iz = (int)floor(cz + 0.5);
                v = ld_any<TI>(vol, ((int64_t)ix * Y + iy) * Z + iz);
            } else {
                const int ix = (int)floor(cx), iy = (int)floor(cy), iz = (int)floor(cz);
                const double fx = cx - ix, fy = cy - iy, fz = cz - iz;
                const int jx = min(ix + 1, X - 1), jy = min(iy + 1, Y - 1), jz = min(iz + 1, Z - 1);
                auto at = [&](int a, int b, int c) { return (double)ld_any<TI>(vol, ((int64_t)a * Y + b) * Z + c); };
                const double c00 = at(ix, iy, iz) * (1 - fz) + at(ix, iy, jz) * fz;
                const double c01 = at(ix, jy, iz) * (1 - fz) + at(ix, jy, jz) * fz;
                const double c10 = at(jx, iy, iz) * (1 - fz) + at(jx, iy, jz) * fz;
                const double c11 = at(jx, jy, iz) * (1 - fz) + at(jx, jy, jz) * fz;
                v = (float)((c00 * (1 - fy) + c01 * fy) * (1 - fx) + (c10 * (1 - fy) + c11 * fy) * fx);
            }
        }
        st_any<TO>(out, q * ld + k, v);
    }
}

// order-preserving float <-> uint key, so that min / max are plain unsigned atomics
__device__ __forceinline__ unsigned f2key(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

__global__ void k_minmax_init(unsigned* mm) {
    mm[0] = 0xffffffffu;
    mm[1] = 0u;
}
// 16-byte loads, wave shuffle, then ONE atomic pair per workgroup (4 waves through LDS) and at most 256 workgroups: the round-2 form
// (4 B per thread and iteration, one atomic pair per WAVE of up to 1,024 workgroups: 8,192 device-scope atomics on two addresses) took
// 100 us per 64x128x128 patch next to the training step's kernels - 0.8 ms of an augmented batch of four (rocprofv3, tools/trace_fit.sh)
template <typename T>
__global__ void __launch_bounds__(256) k_minmax(const T* __restrict__ x, int64_t n, unsigned* mm) {
    constexpr int V = 16 / sizeof(T);
    float lo = INFINITY, hi = -INFINITY;
    const int64_t nv = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) ? n / V : 0;      // vector part (aligned base), the tail goes scalar
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nv; t += (int64_t)gridDim.x * blockDim.x) {
        const uint4 q = reinterpret_cast<const uint4*>(x)[t];
        const T* e = reinterpret_cast<const T*>(&q);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const float v = to_f<T>(e[i]);
            lo = fminf(lo, v);
            hi = fmaxf(hi, v);
        }
    }
    for (int64_t t = nv * V + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const float v = to_f<T>(x[t]);
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    __shared__ float slo[4], shi[4];
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        lo = fminf(fminf(slo[0], slo[1]), fminf(slo[2], slo[3]));
        hi = fmaxf(fmaxf(shi[0], shi[1]), fmaxf(shi[2], shi[3]));
        if (lo <= hi) {
            atomicMin(&mm[0], f2key(lo));
            atomicMax(&mm[1], f2key(hi));
        }
    }
}
__global__ void k_minmax_decode(unsigned* mm) {
    const float lo = key2f(mm[0]), hi = key2f(mm[1]);
    reinterpret_cast<float*>(mm)[0] = lo;
    reinterpret_cast<float*>(mm)[1] = hi;
}

// skimage.exposure.rescale_intensity(x, in_range=(lo, hi), out_range='image') then x *= mult (reference augment.py:125-128, :348-352)
template <typename T>
__global__ void k_rescale(T* __restrict__ x, int64_t n, const float* __restrict__ stats, int contrast, float lo, float hi, float mult) {
    const float omin = stats[0], omax = stats[1];
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        float v = to_f<T>(x[t]);
        if (contrast) {
            v = fminf(fmaxf(v, lo), hi);
            if (lo != hi) v = (v - lo) / (hi - lo) * (omax - omin) + omin;
            else v = fminf(fmaxf(v, omin), omax);
        }
        x[t] = from_f<T>(v * mult);
    }
}

// MinMaxScaler((0,1)) -> skimage.util.random_noise(mode = 'gaussian' | 'speckle', clip=True, var=sigma^2) -> inverse scaling
// (reference augment.py:99-110); `noise` holds the N(0,1) draws
template <typename T>
__global__ void k_noise(T* __restrict__ x, int64_t n, const float* __restrict__ stats, const float* __restrict__ noise, int kind, float sigma) {
    const float dmin = stats[0], rng = stats[1] - stats[0];
    const float scale = 1.f / (rng != 0.f ? rng : 1.f), mn = -dmin * scale;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        float s = fminf(fmaxf(to_f<T>(x[t]) * scale + mn, 0.f), 1.f);
        const float e = noise[t] * sigma;
        s = kind == 0 ? s + e : s + s * e;
        s = fminf(fmaxf(s, 0.f), 1.f);
        x[t] = from_f<T>((s - mn) / scale);
    }
}

// shot noise (reference augment.py:87-94): MinMaxScaler((0,1)) -> floor(x * 1023) / 1023 -> skimage random_noise('poisson', clip=True) ->
// inverse scaling.  skimage: vals = 2 ** ceil(log2(number of distinct values)); out = poisson(image * vals) / vals.  Three passes:
//   phase 0  mark which of the 1024 quantisation levels occur                               (present[1024], zeroed by the caller)
//   phase 1  rates[t] = q[t] * vals  - the caller draws Poisson(rates) (torch.poisson; the draws are the only random part)
//   phase 2  x[t] = (clip(draws[t] / vals, 0, 1) - mn) / scale
__device__ __forceinline__ float shot_vals(const int* __restrict__ present) {
    int cnt = 0;
    for (int i = 0; i < 1024; ++i) cnt += present[i] != 0;
    int v = 1;
    while (v < cnt) v <<= 1;                       // 2 ** ceil(log2(cnt))
    return (float)v;
}
template <typename T>
__global__ void k_shot_noise(T* __restrict__ x, int64_t n, const float* __restrict__ stats, int* __restrict__ present, float* __restrict__ rates,
                             const float* __restrict__ draws, int phase) {
    const float dmin = stats[0], rng = stats[1] - stats[0];
    const float scale = 1.f / (rng != 0.f ? rng : 1.f), mn = -dmin * scale;
    __shared__ float vals_s;
    if (phase > 0) {
        if (threadIdx.x == 0) vals_s = shot_vals(present);
        __syncthreads();
    }
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        if (phase == 2) {
            const float s = fminf(fmaxf(draws[t] / vals_s, 0.f), 1.f);
            x[t] = from_f<T>((s - mn) / scale);
            continue;
        }
        const float s = fminf(fmaxf(to_f<T>(x[t]) * scale + mn, 0.f), 1.f);
        const float level = floorf(s * 1023.f);
        if (phase == 0) present[(int)level] = 1;
        else rates[t] = level / 1023.f * vals_s;
    }
}

// ---- imgaug's two augmenters of the reference's DEFAULT config (fetal/config_utils.py:104-112): ElasticTransformation and CoarseDropout,
// applied by reference fetal_net/augment.py:116-120 (coarse dropout) and :149-170 (elastic transform).  imgaug sees a patch [X][Y][Z] as an
// image of height X, width Y with Z channels: one in-plane displacement field / one low-resolution mask grid for all slices.
//
// Elastic: dst[i][j][c] = src[:, :, c] at (i - d0[i][j], j - d1[i][j]) (imgaug 0.4.0 `_map_coordinates`: x_shifted = x + (-1) * dx),
// scipy map_coordinates order 0 (floor(c + 0.5)) or 1 (bilinear), mode 'nearest' (coordinates clamped to the image); fp64 coordinates.
// This is `_map_coordinates`' SCIPY branch.  For float32 / float64 images at order 0 / 1 imgaug 0.4.0 prefers its cv2 branch
// (cv2.convertMaps to CV_16SC2 + cv2.remap: coordinates quantised to 1/32 pixel, fixed-point INTER_LINEAR weights, cvRound for order 0,
// BORDER_REPLICATE) and keeps scipy for the dtypes cv2 rejects - cv2 is installed nowhere here, so the fallback branch is the one
// restated.  With the reference's alpha <= 5, sigma = 10 the displacement is ~0.1 pixel: the two branches differ by at most 1/64 pixel
// of coordinate.  Parity unpinned either way (oracle/augment_oracle.py).
template <typename T>
__global__ void k_elastic_warp(const T* __restrict__ src, int X, int Y, int C, int src_ld, const float* __restrict__ d0, const float* __restrict__ d1,
                               int order, T* __restrict__ dst, int dst_ld, int64_t src_stride, int64_t dst_stride, int64_t d_stride) {
    src += blockIdx.y * src_stride;                             // blockIdx.y = patch of the batch
    dst += blockIdx.y * dst_stride;
    d0 += blockIdx.y * d_stride;
    d1 += blockIdx.y * d_stride;
    const int64_t total = (int64_t)X * Y * C;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % C);
        const int64_t ij = t / C;
        const int j = (int)(ij % Y), i = (int)(ij / Y);
        const double ci = (double)i - (double)d0[ij], cj = (double)j - (double)d1[ij];
        float v;
        if (order == 0) {
            const int si = min(max((int)floor(ci + 0.5), 0), X - 1), sj = min(max((int)floor(cj + 0.5), 0), Y - 1);
            v = ld_any<T>(src, ((int64_t)si * Y + sj) * src_ld + c);
        } else {
            const double fi = floor(ci), fj = floor(cj);
            const double ti = ci - fi, tj = cj - fj;
            const int i0 = min(max((int)fi, 0), X - 1), i1 = min(max((int)fi + 1, 0), X - 1);
            const int j0 = min(max((int)fj, 0), Y - 1), j1 = min(max((int)fj + 1, 0), Y - 1);
            const double v00 = ld_any<T>(src, ((int64_t)i0 * Y + j0) * src_ld + c), v01 = ld_any<T>(src, ((int64_t)i0 * Y + j1) * src_ld + c);
            const double v10 = ld_any<T>(src, ((int64_t)i1 * Y + j0) * src_ld + c), v11 = ld_any<T>(src, ((int64_t)i1 * Y + j1) * src_ld + c);
            v = (float)((1.0 - ti) * ((1.0 - tj) * v00 + tj * v01) + ti * ((1.0 - tj) * v10 + tj * v11));
        }
        st_any<T>(dst, ij * dst_ld + c, v);
    }
}

// imgaug PiecewiseAffine with the reference's 2 x 2 grid (fetal_net/augment.py:131-146: nb_rows = nb_cols = 2): the four image corners move, skimage's
// PiecewiseAffineTransform triangulates the SOURCE corners (scipy Delaunay: the diagonal (0,0) - (h,w); simplices {p3,p2,p0} and {p1,p3,p0}) and
// warp() reads output pixel (i, j) from tri(i, j) . (i, j, 1): triangle 0 where j * X <= i * Y (below the diagonal), else triangle 1.  Sampling as
// k_affine_sample (scipy map_coordinates mode 'constant', cval 0: no interpolation beyond the edges), per slice c.  m: 2 x (2 x 3) fp64.
struct Pw2 { double m[12]; };
template <typename T>
__global__ void k_piecewise_affine2(const T* __restrict__ src, int X, int Y, int C, int src_ld, Pw2 P, int order, T* __restrict__ dst, int dst_ld) {
    const int64_t total = (int64_t)X * Y * C;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % C);
        const int64_t ij = t / C;
        const int j = (int)(ij % Y), i = (int)(ij / Y);
        const double* m = P.m + (((int64_t)j * X <= (int64_t)i * Y) ? 0 : 6);
        const double ci = m[0] * i + m[1] * j + m[2], cj = m[3] * i + m[4] * j + m[5];
        float v = 0.f;
        if (ci >= 0.0 && ci <= (double)(X - 1) && cj >= 0.0 && cj <= (double)(Y - 1)) {
            if (order == 0) {
                v = ld_any<T>(src, ((int64_t)floor(ci + 0.5) * Y + (int64_t)floor(cj + 0.5)) * src_ld + c);
            } else {
                const int i0 = (int)floor(ci), j0 = (int)floor(cj);
                const double ti = ci - i0, tj = cj - j0;
                const int i1 = min(i0 + 1, X - 1), j1 = min(j0 + 1, Y - 1);
                const double v00 = ld_any<T>(src, ((int64_t)i0 * Y + j0) * src_ld + c), v01 = ld_any<T>(src, ((int64_t)i0 * Y + j1) * src_ld + c);
                const double v10 = ld_any<T>(src, ((int64_t)i1 * Y + j0) * src_ld + c), v11 = ld_any<T>(src, ((int64_t)i1 * Y + j1) * src_ld + c);
                v = (float)((1.0 - ti) * ((1.0 - tj) * v00 + tj * v01) + ti * ((1.0 - tj) * v10 + tj * v11));
            }
        }
        st_any<T>(dst, ij * dst_ld + c, v);
    }
}

// Coarse dropout: x[i][j][c] keeps its value where the low-resolution mask keep[hs][ws][kc] (kc = C: one grid per slice, or 1), enlarged to
// X x Y by nearest neighbour the way cv2.resize(INTER_NEAREST) does it (source index = min(floor(dst * hs / X), hs - 1)), is 1; a dropped
// voxel becomes 0 in the reference's [0, 255] min-max scaling, i.e. the patch's minimum (stats[0] = fmri_minmax of x before the call).
template <typename T>
__global__ void k_coarse_dropout(T* __restrict__ x, int X, int Y, int C, int ld, const uint8_t* __restrict__ keep, int hs, int ws, int kc,
                                 const float* __restrict__ stats) {
    const int64_t total = (int64_t)X * Y * C;
    const double fi = (double)hs / X, fj = (double)ws / Y;
    const float lo = stats[0];
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % C);
        const int64_t ij = t / C;
        const int j = (int)(ij % Y), i = (int)(ij / Y);
        const int si = min((int)floor(i * fi), hs - 1), sj = min((int)floor(j * fj), ws - 1);
        if (!keep[((int64_t)si * ws + sj) * kc + (kc == 1 ? 0 : c)]) st_any<T>(x, ij * ld + c, lo);
    }
}


// ---- the intensity chain with the random draws made IN the kernels (round 5).  The forms above take the draws from the
// caller (tests feed the oracle's own); a training patch does not need that, and the round trip through torch's generator made a patch
// ~35 launches: a fill, torch.poisson (121 us per 64x128x128 patch), torch.randn per noise kind, three launches per min / max.  Here:
//   * Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11), keyed by the sampler's seed; the
//     counter is (element or cell index, call number, rejection round): the draws depend on neither the grid nor the launch order.
//   * every kernel that rewrites the image also reduces the min / max of what it wrote and the LAST workgroup to finish (a ticket) decodes it
//     into `stats` and re-arms the workspace - the next step of the chain finds its MinMaxScaler range without a pass of its own.
// Workspace `ws` (int32 x FMRI_AUG_WS_INTS, zeroed ONCE by the caller; every kernel leaves it zeroed): [2] ticket of the min / max
// reduction, [3] ticket of the level table, [32 + 32 s] ~key(min) and [33 + 32 s] key(max) of slot s < 16, [544 .. 1568) the 1,024 level
// marks of the shot noise.
constexpr int AUG_WS_SLOTS = 32, AUG_WS_PRESENT = 544;
constexpr int AUG_BLOCKS = 256;       // one atomic pair + one ticket per workgroup: 768 device-scope atomics per launch
constexpr int AUG_THREADS = 1024;     // ... so the parallelism of the kernels that draw comes from 16 waves per workgroup
// One launch covers the patches of a batch (blockIdx.y = patch; patch b's image at x + b * stride, its range at stats + 2 b, its workspace
// at ws + b * AUG_WS_INTS): a 64x128x128 patch is a 5 us kernel of which 4 us are launch ramp, and the step's persistent workgroups leave
// a generator kernel nowhere to hide - what a batch costs the training step is the NUMBER of its launches.  Per-patch parameters by value.
constexpr int AUG_MAXB = 16, AUG_WS_INTS = 1568;
struct SeqB { unsigned v[AUG_MAXB]; };                       // the call number of the patch's draws; 0 = the patch skips this step
struct RescaleB { int mode[AUG_MAXB]; float lo[AUG_MAXB], hi[AUG_MAXB], mult[AUG_MAXB]; };      // mode 0 skip, 1 multiply, 2 contrast + multiply
struct GridB { int hs[AUG_MAXB], ws[AUG_MAXB]; };
struct AlphaB { float v[AUG_MAXB]; };

struct Philox {
    unsigned k0, k1;
    __device__ __forceinline__ uint4 operator()(unsigned c0, unsigned c1, unsigned c2, unsigned c3) const {
        unsigned a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const unsigned h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
            const unsigned h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
            c0 = h1 ^ c1 ^ a;
            c1 = l1;
            c2 = h0 ^ c3 ^ b;
            c3 = l0;
            a += 0x9E3779B9u;
            b += 0xBB67AE85u;
        }
        return make_uint4(c0, c1, c2, c3);
    }
};
__device__ __forceinline__ float u01_open_low(unsigned r) { return (float)((r >> 8) + 1u) * (1.f / 16777216.f); }    // (0, 1]
__device__ __forceinline__ float u01(unsigned r) { return (float)(r >> 8) * (1.f / 16777216.f); }                    // [0, 1)

// workgroup reduction of (lo, hi) -> one atomic pair on one of 16 slot pairs (128 B apart: atomics on ONE address serialise in its L2
// channel, ~12 ns each); the last workgroup (ticket) folds the slots into stats = {min, max} and leaves slots and ticket zeroed
__device__ __forceinline__ void emit_minmax(float lo, float hi, int* __restrict__ ws, float* __restrict__ stats) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    __shared__ float slo[16], shi[16];
    __shared__ int last;
    unsigned* w = reinterpret_cast<unsigned*>(ws);
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) {
            lo = fminf(lo, slo[i]);
            hi = fmaxf(hi, shi[i]);
        }
        unsigned* slot = w + AUG_WS_SLOTS + 32 * (blockIdx.x & 15);
        if (lo <= hi) {
            atomicMax(&slot[0], ~f2key(lo));
            atomicMax(&slot[1], f2key(hi));
        }
        __threadfence();
        last = atomicAdd(&w[2], 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (last && threadIdx.x < 32) {
        __threadfence();
        unsigned v = atomicExch(&w[AUG_WS_SLOTS + 32 * (threadIdx.x >> 1) + (threadIdx.x & 1)], 0u);      // even lanes ~key(min), odd key(max)
#pragma unroll
        for (int o = 2; o < 32; o <<= 1) v = max(v, (unsigned)__shfl_xor((int)v, o));
        // no workgroup contributed (every element NaN, or an empty patch): all slots are still 0 - leave the range {0, 0}, which the
        // consumers treat as rng = 1, instead of decoding the two ends of the key space into a range that poisons every later step
        const bool none = (__shfl((int)v, 0) | __shfl((int)v, 1)) == 0;
        if (threadIdx.x == 0) stats[0] = none ? 0.f : key2f(~v);
        if (threadIdx.x == 1) stats[1] = none ? 0.f : key2f(v);
        __threadfence();
        if (threadIdx.x == 0) atomicExch(&w[2], 0u);
    }
}

template <typename T>
__global__ void __launch_bounds__(AUG_THREADS) k_minmax_ws(const T* __restrict__ x, int64_t n, int64_t stride, int* __restrict__ ws,
                                                           float* __restrict__ stats) {
    x += blockIdx.y * stride;
    ws += blockIdx.y * AUG_WS_INTS;
    stats += blockIdx.y * 2;
    constexpr int V = 16 / sizeof(T);
    float lo = INFINITY, hi = -INFINITY;
    const int64_t nv = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) ? n / V : 0;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nv; t += (int64_t)gridDim.x * blockDim.x) {
        const uint4 q = reinterpret_cast<const uint4*>(x)[t];
        const T* e = reinterpret_cast<const T*>(&q);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const float v = to_f<T>(e[i]);
            lo = fminf(lo, v);
            hi = fmaxf(hi, v);
        }
    }
    for (int64_t t = nv * V + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const float v = to_f<T>(x[t]);
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
    emit_minmax(lo, hi, ws, stats);
}

// k_rescale that leaves the new range behind
template <typename T>
__global__ void __launch_bounds__(AUG_THREADS) k_rescale_ws(T* __restrict__ x, int64_t n, int64_t stride, float* __restrict__ stats,
                                                            int* __restrict__ ws, RescaleB P) {
    const int b = blockIdx.y;
    if (P.mode[b] == 0) return;
    x += b * stride;
    ws += b * AUG_WS_INTS;
    stats += b * 2;
    const int contrast = P.mode[b] == 2;
    const float lo = P.lo[b], hi = P.hi[b], mult = P.mult[b];
    const float omin = stats[0], omax = stats[1];
    float mn = INFINITY, mx = -INFINITY;
    constexpr int64_t CH = 4 * AUG_THREADS;                    // four loads in flight per thread (a grid-stride loop has one)
    for (int64_t c = blockIdx.x; c * CH < n; c += gridDim.x) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t t = c * CH + k * AUG_THREADS + threadIdx.x;
            v[k] = t < n ? to_f<T>(x[t]) : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t t = c * CH + k * AUG_THREADS + threadIdx.x;
            if (t >= n) continue;
            if (contrast) {
                v[k] = fminf(fmaxf(v[k], lo), hi);
                if (lo != hi) v[k] = (v[k] - lo) / (hi - lo) * (omax - omin) + omin;
                else v[k] = fminf(fmaxf(v[k], omin), omax);
            }
            const T o = from_f<T>(v[k] * mult);
            x[t] = o;
            const float w = to_f<T>(o);
            mn = fminf(mn, w);
            mx = fmaxf(mx, w);
        }
    }
    __syncthreads();
    emit_minmax(mn, mx, ws, stats);
}

// gaussian (kind 0) / speckle (kind 1) noise as k_noise, the N(0,1) draws by Box-Muller from one Philox block per four elements
template <typename T>
__global__ void __launch_bounds__(AUG_THREADS) k_noise_rng(T* __restrict__ x, int64_t n, int64_t stride, float* __restrict__ stats,
                                                           int* __restrict__ ws, int kind, float sigma, Philox rng, SeqB S) {
    const unsigned seq = S.v[blockIdx.y];
    if (seq == 0) return;
    x += blockIdx.y * stride;
    ws += blockIdx.y * AUG_WS_INTS;
    stats += blockIdx.y * 2;
    const float dmin = stats[0], rng_ = stats[1] - stats[0];
    const float scale = 1.f / (rng_ != 0.f ? rng_ : 1.f), mn = -dmin * scale;
    float lo = INFINITY, hi = -INFINITY;
    constexpr int64_t CH = 4 * AUG_THREADS;                    // a chunk = 4 x 1,024 elements: lane-contiguous accesses, 4 draws per thread
    const int64_t chunks = (n + CH - 1) / CH;
    for (int64_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        const uint64_t ctr = (uint64_t)c * AUG_THREADS + threadIdx.x;
        const uint4 r = rng((unsigned)ctr, (unsigned)(ctr >> 32), seq, 0u);
        float z[4];
        {
            const float m0 = sqrtf(-2.f * __logf(u01_open_low(r.x))), m1 = sqrtf(-2.f * __logf(u01_open_low(r.z)));
            float s0, c0, s1, c1;
            __sincosf(6.28318530717958647692f * u01(r.y), &s0, &c0);
            __sincosf(6.28318530717958647692f * u01(r.w), &s1, &c1);
            z[0] = m0 * c0; z[1] = m0 * s0; z[2] = m1 * c1; z[3] = m1 * s1;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t t = c * CH + k * AUG_THREADS + threadIdx.x;
            if (t < n) {
                float s = fminf(fmaxf(to_f<T>(x[t]) * scale + mn, 0.f), 1.f);
                const float e = z[k] * sigma;
                s = kind == 0 ? s + e : s + s * e;
                s = fminf(fmaxf(s, 0.f), 1.f);
                const T o = from_f<T>((s - mn) / scale);
                x[t] = o;
                const float v = to_f<T>(o);
                lo = fminf(lo, v);
                hi = fmaxf(hi, v);
            }
        }
    }
    __syncthreads();                                           // every thread has read stats before the last workgroup rewrites it
    emit_minmax(lo, hi, ws, stats);
}

// Poisson(lam) for 0 <= lam <= 1024.  lam < 10: Knuth's product of uniforms; else Hoermann's transformed rejection PTRS ("The transformed
// rejection method for generating Poisson random variables", 1993) - the two branches of numpy's legacy random_poisson, i.e. of the
// np.random.poisson behind skimage.util.random_noise(mode='poisson') that reference augment.py:87-94 calls.  Two uniforms per round.
struct PoissonLevel { float lam, c, b, a, invalpha, vr; };    // c = exp(-lam) for lam < 10, else log(lam)
__device__ __forceinline__ PoissonLevel poisson_level(float lam) {
    PoissonLevel L;
    L.lam = lam;
    if (lam < 10.f) {
        L.c = __expf(-lam);
        L.b = L.a = L.invalpha = L.vr = 0.f;
    } else {
        L.c = logf(lam);
        L.b = 0.931f + 2.53f * sqrtf(lam);
        L.a = -0.059f + 0.02483f * L.b;
        L.invalpha = 1.1239f + 1.1328f / (L.b - 3.4f);
        L.vr = 0.9277f - 3.6224f / (L.b - 2.f);
    }
    return L;
}
__device__ __forceinline__ float poisson_draw(const PoissonLevel L, const Philox& rng, unsigned c0, unsigned c1, unsigned seq) {
    const float lam = L.lam;
    if (!(lam > 0.f)) return 0.f;
    unsigned round = 0;
    if (lam < 10.f) {
        const float enlam = L.c;
        float prod = 1.f;
        int k = 0;
        for (;;) {
            const uint4 r = rng(c0, c1, seq, round++);
            const unsigned u[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                prod *= u01_open_low(u[i]);
                if (prod <= enlam) return (float)k;
                ++k;
            }
        }
    }
    const float loglam = L.c, b = L.b, a = L.a, invalpha = L.invalpha, vr = L.vr;
    for (;;) {
        const uint4 r = rng(c0, c1, seq, round++);
        const unsigned u[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
            const float U = u01(u[i]) - 0.5f, V = u01_open_low(u[i + 1]);
            const float us = 0.5f - fabsf(U);
            const float kf = floorf((2.f * a / us + b) * U + lam + 0.43f);
            if (us >= 0.07f && V <= vr) return kf;
            if (kf < 0.f || (us < 0.013f && V > us)) continue;
            // the squeeze failed (~15 % of the rounds): the exact test  log(V invalpha / (a / us^2 + b)) <= -lam + k log(lam) - log(k!).
            // In fp32 the right side as written is a difference of ~7,000s; with k = lam + d and Stirling's series for log(k!) it is
            // d - k log1p(d / lam) - log(2 pi k) / 2 - 1 / (12 k) + 1 / (360 k^3): every term is O(d), absolute error ~1e-5 (k < 10: the table)
            const float lhs = __logf(V * invalpha / (a / (us * us) + b));
            float rhs;
            if (kf < 10.f) {
                constexpr float LOGFACT[10] = {0.f, 0.f, 0.6931471806f, 1.7917594692f, 3.1780538303f, 4.7874917428f, 6.5792512120f, 8.5251613611f,
                                               10.6046029027f, 12.8018274801f};
                rhs = -lam + kf * loglam - LOGFACT[(int)kf];
            } else {
                const float d = kf - lam, ik = 1.f / kf;
                rhs = d - kf * log1pf(d / lam) - 0.5f * __logf(6.28318530717958647692f * kf) - ik * (1.f / 12.f) + ik * ik * ik * (1.f / 360.f);
            }
            if (lhs <= rhs) return kf;
        }
    }
}

// shot noise, launch 1 of 2: mark the occupied quantisation levels (k_shot_noise phase 0 on the workspace's table)
template <typename T>
__global__ void __launch_bounds__(256) k_shot_levels(const T* __restrict__ x, int64_t n, int64_t stride, const float* __restrict__ stats,
                                                     int* __restrict__ ws, SeqB S) {
    if (S.v[blockIdx.y] == 0) return;
    x += blockIdx.y * stride;
    ws += blockIdx.y * AUG_WS_INTS;
    stats += blockIdx.y * 2;
    const float dmin = stats[0], rng_ = stats[1] - stats[0];
    const float scale = 1.f / (rng_ != 0.f ? rng_ : 1.f), mn = -dmin * scale;
    int* present = ws + AUG_WS_PRESENT;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const float s = fminf(fmaxf(to_f<T>(x[t]) * scale + mn, 0.f), 1.f);
        present[(int)floorf(s * 1023.f)] = 1;
    }
}
// launch 2 of 2: phases 1 + 2 of k_shot_noise around an in-kernel Poisson draw; clears the level table behind the last reader
template <typename T>
__global__ void __launch_bounds__(AUG_THREADS) k_shot_draw(T* __restrict__ x, int64_t n, int64_t stride, float* __restrict__ stats,
                                                           int* __restrict__ ws, Philox rng, SeqB S) {
    const unsigned seq = S.v[blockIdx.y];
    if (seq == 0) return;
    x += blockIdx.y * stride;
    ws += blockIdx.y * AUG_WS_INTS;
    stats += blockIdx.y * 2;
    const float dmin = stats[0], rng_ = stats[1] - stats[0];
    const float scale = 1.f / (rng_ != 0.f ? rng_ : 1.f), mn = -dmin * scale;
    int* present = ws + AUG_WS_PRESENT;
    __shared__ int cnt_s;
    __shared__ int last_s;
    if (threadIdx.x == 0) cnt_s = 0;
    __syncthreads();
    {
        int c = present[threadIdx.x] != 0;                     // AUG_THREADS = the 1,024 levels
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        if ((threadIdx.x & 63) == 0) atomicAdd(&cnt_s, c);
    }
    __syncthreads();
    if (threadIdx.x == 0) {                                    // every workgroup has the count before the table may be cleared
        __threadfence();
        last_s = atomicAdd(reinterpret_cast<unsigned*>(&ws[3]), 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (last_s) {
        present[threadIdx.x] = 0;
        if (threadIdx.x == 0) ws[3] = 0;
    }
    int v2 = 1;
    while (v2 < cnt_s) v2 <<= 1;                               // skimage: vals = 2 ** ceil(log2(number of distinct values))
    const float vals = (float)v2;
    __shared__ PoissonLevel levels[AUG_THREADS];               // the sampler's constants of each of the 1,024 rates: one sqrt / log per LEVEL, not per voxel
    levels[threadIdx.x] = poisson_level((float)threadIdx.x / 1023.f * vals);
    __syncthreads();
    float lo = INFINITY, hi = -INFINITY;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        float s = fminf(fmaxf(to_f<T>(x[t]) * scale + mn, 0.f), 1.f);
        const float k = poisson_draw(levels[(int)floorf(s * 1023.f)], rng, (unsigned)t, (unsigned)((uint64_t)t >> 32), seq);
        s = fminf(fmaxf(k / vals, 0.f), 1.f);
        const T o = from_f<T>((s - mn) / scale);
        x[t] = o;
        const float v = to_f<T>(o);
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
    __syncthreads();
    emit_minmax(lo, hi, ws, stats);
}

// The elastic transform's displacement fields (ops.elastic_fields: uniform(-1, 1) noise on the padded image, separable truncated Gaussian,
// times alpha, padding cropped) in ONE launch: a workgroup makes one row segment of one field - it draws the k x (tw + k - 1) noise pixels under
// it (pixel p of the padded (2, X + 2k, Y + 2k) grid = output (p & 3) of Philox block p >> 2: any workgroup that needs the pixel draws the same
// value), blurs along the row, then down the k rows.  field 0 = d1 (imgaug's dx), field 1 = d0 (dy), as ops.elastic_fields.  k <= 31.
constexpr int EF_KMAX = 31, EF_TW = 128;
__global__ void __launch_bounds__(256) k_elastic_fields_rng(float* __restrict__ d, int X, int Y, int k, const double* __restrict__ w, AlphaB A,
                                                            Philox rng, SeqB S) {
    const int pb = blockIdx.z >> 1;                             // d: [B][2][X][Y], d[b][0] = d0, d[b][1] = d1
    float* __restrict__ d0 = d + (int64_t)pb * 2 * X * Y;
    float* __restrict__ d1 = d0 + (int64_t)X * Y;
    const unsigned seq = S.v[pb];
    const float alpha = A.v[pb];
    if (seq == 0) {                                             // no elastic transform for this patch: a zero field (the warp is the identity)
        const int jt0 = blockIdx.x * EF_TW;
        for (int j = threadIdx.x; j < min(EF_TW, Y - jt0); j += blockDim.x) ((blockIdx.z & 1) == 0 ? d1 : d0)[(int64_t)blockIdx.y * Y + jt0 + j] = 0.f;
        return;
    }
    __shared__ float nz[EF_KMAX][EF_TW + EF_KMAX - 1];
    __shared__ float rb[EF_KMAX][EF_TW];
    __shared__ float wf[EF_KMAX];
    const int jt = blockIdx.x * EF_TW, i = blockIdx.y, f = blockIdx.z & 1;
    const int hp = X + 2 * k, wp = Y + 2 * k, r = k / 2;
    const int tw = min(EF_TW, Y - jt), span = tw + k - 1;
    if (threadIdx.x < k) wf[threadIdx.x] = (float)w[threadIdx.x];
    const int row0 = i + k - r, col0 = jt + k - r;
    const int gpr = (span + 6) / 4;                            // Philox blocks that can touch one row's span (pixel p = word p & 3 of block p >> 2)
    for (int idx = threadIdx.x; idx < k * gpr; idx += blockDim.x) {
        const int a = idx / gpr, g = idx - a * gpr;
        const unsigned p0 = (unsigned)((f * hp + row0 + a) * wp + col0);
        const unsigned blk = (p0 >> 2) + g;
        const uint4 q = rng(blk, 0u, seq, 1u);
        const unsigned wv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int b = (int)(blk * 4u + e) - (int)p0;
            if (b >= 0 && b < span) nz[a][b] = u01(wv[e]) * 2.f - 1.f;
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < k * tw; idx += blockDim.x) {
        const int a = idx / tw, j = idx - a * tw;
        float acc = 0.f;
        for (int b = 0; b < k; ++b) acc = fmaf(wf[b], nz[a][j + b], acc);
        rb[a][j] = acc;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < tw; j += blockDim.x) {
        float acc = 0.f;
        for (int a = 0; a < k; ++a) acc = fmaf(wf[a], rb[a][j], acc);
        (f == 0 ? d1 : d0)[(int64_t)i * Y + jt + j] = acc * alpha;
    }
}

// coarse dropout as k_coarse_dropout, the keep grid drawn in the kernel: cell (si, sj[, c]) is dropped when its uniform draw < rate
// (imgaug CoarseDropout(p=rate): a Binomial(1 - rate) keep mask at the low resolution)
template <typename T>
__global__ void k_coarse_dropout_rng(T* __restrict__ x, int X, int Y, int C, int ld, int64_t stride, GridB G, int kc, float rate,
                                     const float* __restrict__ stats, Philox rng, SeqB S) {
    const unsigned seq = S.v[blockIdx.y];
    if (seq == 0) return;
    x += blockIdx.y * stride;
    stats += blockIdx.y * 2;
    const int hs = G.hs[blockIdx.y], wsz = G.ws[blockIdx.y];
    const double fi = (double)hs / X, fj = (double)wsz / Y;
    const float lo = stats[0];
    // cell q of the keep grid ((si * ws + sj) * kc + c) draws word q & 3 of Philox block q >> 2: a thread takes four neighbouring slices of a
    // pixel with ONE block when the slices come in fours (C % 4 == 0), else one voxel - the same cells, the same words
    const int cpt = (C & 3) == 0 ? 4 : 1, CG = C / cpt;
    const int64_t total = (int64_t)X * Y * CG;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % CG) * cpt;
        const int64_t ij = t / CG;
        const int j = (int)(ij % Y), i = (int)(ij / Y);
        const int si = min((int)floor(i * fi), hs - 1), sj = min((int)floor(j * fj), wsz - 1);
        const unsigned cell = (unsigned)((si * wsz + sj) * kc + (kc == 1 ? 0 : c));
        const uint4 q = rng(cell >> 2, 0u, seq, 0u);
        const unsigned wv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < cpt && u01(wv[kc == 1 ? (cell & 3u) : ((cell + e) & 3u)]) < rate) st_any<T>(x, ij * ld + c + e, lo);
    }
}

}  // namespace

extern "C" int fmri_shot_noise_step(void* x, int64_t n, int dtype, const float* stats, int* present, float* rates, const float* draws, int phase,
                                    fmri_stream_t stream) {
    if (n < 1 || phase < 0 || phase > 2 || !x || !stats || !present || (phase == 1 && !rates) || (phase == 2 && !draws)) return FMRI_E_SHAPE;
    hipStream_t st = as_stream(stream);
    const int grid = grid_for(n);
    if (dtype == FMRI_F32) k_shot_noise<float><<<grid, 256, 0, st>>>((float*)x, n, stats, present, rates, draws, phase);
    else if (dtype == FMRI_BF16) k_shot_noise<bf16_t><<<grid, 256, 0, st>>>((bf16_t*)x, n, stats, present, rates, draws, phase);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

static int elastic_warp_launch(const void* src, int dtype, int X, int Y, int C, int src_ld, const float* d0, const float* d1, int order, void* dst,
                               int dst_ld, int64_t src_stride, int64_t dst_stride, int64_t d_stride, int B, hipStream_t st) {
    const dim3 grid(grid_for((int64_t)X * Y * C), B);
    if (dtype == FMRI_F32)
        k_elastic_warp<float><<<grid, 256, 0, st>>>((const float*)src, X, Y, C, src_ld, d0, d1, order, (float*)dst, dst_ld, src_stride, dst_stride, d_stride);
    else if (dtype == FMRI_U8)
        k_elastic_warp<uint8_t><<<grid, 256, 0, st>>>((const uint8_t*)src, X, Y, C, src_ld, d0, d1, order, (uint8_t*)dst, dst_ld, src_stride, dst_stride,
                                                      d_stride);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
extern "C" int fmri_elastic_warp(const void* src, int dtype, int X, int Y, int C, int src_ld, const float* d0, const float* d1, int order, void* dst,
                                int dst_ld, fmri_stream_t stream) {
    if (X < 1 || Y < 1 || C < 1 || src_ld < C || dst_ld < C || (order != 0 && order != 1) || !d0 || !d1 || src == dst) return FMRI_E_SHAPE;
    return elastic_warp_launch(src, dtype, X, Y, C, src_ld, d0, d1, order, dst, dst_ld, 0, 0, 0, 1, as_stream(stream));
}
extern "C" int fmri_elastic_warp_batch(const void* src, int dtype, int X, int Y, int C, int src_ld, int64_t src_stride, const float* d, int order,
                                      void* dst, int dst_ld, int64_t dst_stride, int B, fmri_stream_t stream) {
    if (X < 1 || Y < 1 || C < 1 || src_ld < C || dst_ld < C || (order != 0 && order != 1) || !d || src == dst || B < 1 || B > 65535) return FMRI_E_SHAPE;
    return elastic_warp_launch(src, dtype, X, Y, C, src_ld, d, d + (int64_t)X * Y, order, dst, dst_ld, src_stride, dst_stride, (int64_t)2 * X * Y, B,
                               as_stream(stream));
}

extern "C" int fmri_piecewise_affine2(const void* src, int dtype, int X, int Y, int C, int src_ld, const double* tri, int order, void* dst, int dst_ld,
                                     fmri_stream_t stream) {
    if (X < 1 || Y < 1 || C < 1 || src_ld < C || dst_ld < C || (order != 0 && order != 1) || !tri || src == dst) return FMRI_E_SHAPE;
    Pw2 P;
    for (int i = 0; i < 12; ++i) P.m[i] = tri[i];
    hipStream_t st = as_stream(stream);
    const int grid = grid_for((int64_t)X * Y * C);
    if (dtype == FMRI_F32) k_piecewise_affine2<float><<<grid, 256, 0, st>>>((const float*)src, X, Y, C, src_ld, P, order, (float*)dst, dst_ld);
    else if (dtype == FMRI_U8) k_piecewise_affine2<uint8_t><<<grid, 256, 0, st>>>((const uint8_t*)src, X, Y, C, src_ld, P, order, (uint8_t*)dst, dst_ld);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

extern "C" int fmri_coarse_dropout(void* x, int dtype, int X, int Y, int C, int ld, const uint8_t* keep, int hs, int ws, int kc, const float* stats,
                                   fmri_stream_t stream) {
    if (X < 1 || Y < 1 || C < 1 || ld < C || hs < 1 || ws < 1 || (kc != 1 && kc != C) || !keep || !stats) return FMRI_E_SHAPE;
    hipStream_t st = as_stream(stream);
    const int grid = grid_for((int64_t)X * Y * C);
    if (dtype == FMRI_F32) k_coarse_dropout<float><<<grid, 256, 0, st>>>((float*)x, X, Y, C, ld, keep, hs, ws, kc, stats);
    else if (dtype == FMRI_BF16) k_coarse_dropout<bf16_t><<<grid, 256, 0, st>>>((bf16_t*)x, X, Y, C, ld, keep, hs, ws, kc, stats);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

static int affine_sample_launch(const AffineB& P, int B, int vol_dtype, int nx, int ny, int nz, int order, void* out, int out_dtype, int out_ld,
                                int64_t out_stride, hipStream_t st) {
    const dim3 grid(grid_for((int64_t)nx * ny * nz), B);
#define FMRI_AS(TI, TO) k_affine_sample<TI, TO><<<grid, 256, 0, st>>>(P, nx, ny, nz, order, (TO*)out, out_ld, out_stride)
    if (vol_dtype == FMRI_F32 && out_dtype == FMRI_F32) FMRI_AS(float, float);
    else if (vol_dtype == FMRI_F32 && out_dtype == FMRI_BF16) FMRI_AS(float, bf16_t);
    else if (vol_dtype == FMRI_U8 && out_dtype == FMRI_U8) FMRI_AS(uint8_t, uint8_t);
    else if (vol_dtype == FMRI_U8 && out_dtype == FMRI_F32) FMRI_AS(uint8_t, float);
    else if (vol_dtype == FMRI_U8 && out_dtype == FMRI_BF16) FMRI_AS(uint8_t, bf16_t);
    else return FMRI_E_DTYPE;
#undef FMRI_AS
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
extern "C" int fmri_affine_sample(const void* vol, int vol_dtype, int X, int Y, int Z, const double* affine, int x0, int y0, int z0, int nx,
                                  int ny, int nz, int order, float cval, void* out, int out_dtype, int out_ld, fmri_stream_t stream) {
    if (X < 1 || Y < 1 || Z < 1 || nx < 1 || ny < 1 || nz < 1 || out_ld < nz || (order != 0 && order != 1) || !affine) return FMRI_E_SHAPE;
    AffineB P;
    P.vol[0] = vol;
    P.X[0] = X; P.Y[0] = Y; P.Z[0] = Z; P.x0[0] = x0; P.y0[0] = y0; P.z0[0] = z0;
    P.cval[0] = cval;
    for (int i = 0; i < 12; ++i) P.A[0].a[i] = affine[i];
    return affine_sample_launch(P, 1, vol_dtype, nx, ny, nz, order, out, out_dtype, out_ld, 0, as_stream(stream));
}
extern "C" int fmri_affine_sample_batch(int B, const void* const* vols, const int* dims, const double* affines, const int* corners, const float* cvals,
                                        int vol_dtype, int nx, int ny, int nz, int order, void* out, int out_dtype, int out_ld, int64_t out_stride,
                                        fmri_stream_t stream) {
    if (B < 1 || !vols || !dims || !affines || !corners || !cvals || nx < 1 || ny < 1 || nz < 1 || out_ld < nz || (order != 0 && order != 1))
        return FMRI_E_SHAPE;
    const int esz = out_dtype == FMRI_F32 ? 4 : (out_dtype == FMRI_BF16 ? 2 : 1);
    for (int b0 = 0; b0 < B; b0 += AFF_MAXB) {
        const int nb = B - b0 < AFF_MAXB ? B - b0 : AFF_MAXB;
        AffineB P;
        for (int b = 0; b < nb; ++b) {
            const int g = b0 + b;
            if (!vols[g] || dims[3 * g] < 1 || dims[3 * g + 1] < 1 || dims[3 * g + 2] < 1) return FMRI_E_SHAPE;
            P.vol[b] = vols[g];
            P.X[b] = dims[3 * g]; P.Y[b] = dims[3 * g + 1]; P.Z[b] = dims[3 * g + 2];
            P.x0[b] = corners[3 * g]; P.y0[b] = corners[3 * g + 1]; P.z0[b] = corners[3 * g + 2];
            P.cval[b] = cvals[g];
            for (int i = 0; i < 12; ++i) P.A[b].a[i] = affines[12 * g + i];
        }
        const int rc = affine_sample_launch(P, nb, vol_dtype, nx, ny, nz, order, (char*)out + (int64_t)b0 * out_stride * esz, out_dtype, out_ld, out_stride,
                                            as_stream(stream));
        if (rc != FMRI_OK) return rc;
    }
    return FMRI_OK;
}

extern "C" int fmri_minmax(const void* x, int64_t n, int dtype, float* out2, fmri_stream_t stream) {
    if (n < 1) return FMRI_E_SHAPE;
    hipStream_t st = as_stream(stream);
    unsigned* mm = reinterpret_cast<unsigned*>(out2);
    k_minmax_init<<<1, 1, 0, st>>>(mm);
    const int grid = grid_for((n + 3) / 4, 256, 256);
    if (dtype == FMRI_F32) k_minmax<float><<<grid, 256, 0, st>>>((const float*)x, n, mm);
    else if (dtype == FMRI_BF16) k_minmax<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)x, n, mm);
    else return FMRI_E_DTYPE;
    k_minmax_decode<<<1, 1, 0, st>>>(mm);
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

extern "C" int fmri_rescale_intensity(void* x, int64_t n, int dtype, const float* stats, int contrast, float lo, float hi, float mult,
                                      fmri_stream_t stream) {
    if (n < 1) return FMRI_E_SHAPE;
    hipStream_t st = as_stream(stream);
    const int grid = grid_for(n);
    if (dtype == FMRI_F32) k_rescale<float><<<grid, 256, 0, st>>>((float*)x, n, stats, contrast, lo, hi, mult);
    else if (dtype == FMRI_BF16) k_rescale<bf16_t><<<grid, 256, 0, st>>>((bf16_t*)x, n, stats, contrast, lo, hi, mult);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

extern "C" int fmri_noise_augment(void* x, int64_t n, int dtype, const float* stats, const float* noise, int kind, float sigma,
                                  fmri_stream_t stream) {
    if (n < 1 || (kind != 0 && kind != 1)) return FMRI_E_SHAPE;
    hipStream_t st = as_stream(stream);
    const int grid = grid_for(n);
    if (dtype == FMRI_F32) k_noise<float><<<grid, 256, 0, st>>>((float*)x, n, stats, noise, kind, sigma);
    else if (dtype == FMRI_BF16) k_noise<bf16_t><<<grid, 256, 0, st>>>((bf16_t*)x, n, stats, noise, kind, sigma);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

// ---- the same intensity steps with in-kernel Philox draws and chained min / max, over the B patches of a batch (see the kernels' comment
// block above).  Per-patch arrays live on the HOST and are read at enqueue.
static inline Philox philox_of(uint64_t seed) {
    Philox p;
    p.k0 = (unsigned)seed;
    p.k1 = (unsigned)(seed >> 32);
    return p;
}
static inline SeqB seqs_of(const uint32_t* seqs, int b0, int nb) {
    SeqB S;
    for (int b = 0; b < AUG_MAXB; ++b) S.v[b] = b < nb ? seqs[b0 + b] : 0u;
    return S;
}
template <typename T> static inline T* patch_ptr(T* base, int64_t stride, int b0) { return base + (int64_t)b0 * stride; }
#define FMRI_FOR_CHUNKS(B) for (int b0 = 0, nb = 0; b0 < (B) && ((nb = (B) - b0 < AUG_MAXB ? (B) - b0 : AUG_MAXB), true); b0 += AUG_MAXB)

extern "C" int fmri_minmax_ws_batch(const void* x, int64_t n, int64_t stride, int B, int dtype, float* stats, int* ws, fmri_stream_t stream) {
    if (n < 1 || B < 1 || B > 65535 || !x || !stats || !ws) return FMRI_E_SHAPE;
    hipStream_t st = as_stream(stream);
    const dim3 grid(grid_for((n + 3) / 4, AUG_THREADS, AUG_BLOCKS), B);
    if (dtype == FMRI_F32) k_minmax_ws<float><<<grid, AUG_THREADS, 0, st>>>((const float*)x, n, stride, ws, stats);
    else if (dtype == FMRI_BF16) k_minmax_ws<bf16_t><<<grid, AUG_THREADS, 0, st>>>((const bf16_t*)x, n, stride, ws, stats);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

extern "C" int fmri_rescale_intensity_ws_batch(void* x, int64_t n, int64_t stride, int B, int dtype, float* stats, int* ws, const float* params,
                                               fmri_stream_t stream) {
    if (n < 1 || B < 1 || !x || !stats || !ws || !params) return FMRI_E_SHAPE;
    if (dtype != FMRI_F32 && dtype != FMRI_BF16) return FMRI_E_DTYPE;
    hipStream_t st = as_stream(stream);
    FMRI_FOR_CHUNKS(B) {
        RescaleB P;
        for (int b = 0; b < AUG_MAXB; ++b) {
            const float* q = params + 4 * (b0 + (b < nb ? b : 0));
            P.mode[b] = b < nb ? (int)q[0] : 0;
            P.lo[b] = q[1]; P.hi[b] = q[2]; P.mult[b] = q[3];
            if (P.mode[b] < 0 || P.mode[b] > 2) return FMRI_E_SHAPE;
        }
        const dim3 grid(grid_for(n, AUG_THREADS, AUG_BLOCKS), nb);
        if (dtype == FMRI_F32)
            k_rescale_ws<float><<<grid, AUG_THREADS, 0, st>>>(patch_ptr((float*)x, stride, b0), n, stride, stats + 2 * b0, ws + b0 * AUG_WS_INTS, P);
        else
            k_rescale_ws<bf16_t><<<grid, AUG_THREADS, 0, st>>>(patch_ptr((bf16_t*)x, stride, b0), n, stride, stats + 2 * b0, ws + b0 * AUG_WS_INTS, P);
    }
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

extern "C" int fmri_noise_rng_batch(void* x, int64_t n, int64_t stride, int B, int dtype, float* stats, int* ws, int kind, float sigma, uint64_t seed,
                                    const uint32_t* seqs, fmri_stream_t stream) {
    if (n < 1 || B < 1 || (kind != 0 && kind != 1) || !x || !stats || !ws || !seqs) return FMRI_E_SHAPE;
    if (dtype != FMRI_F32 && dtype != FMRI_BF16) return FMRI_E_DTYPE;
    hipStream_t st = as_stream(stream);
    FMRI_FOR_CHUNKS(B) {
        const dim3 grid(grid_for((n + 3) / 4, AUG_THREADS, AUG_BLOCKS), nb);
        const SeqB S = seqs_of(seqs, b0, nb);
        if (dtype == FMRI_F32)
            k_noise_rng<float><<<grid, AUG_THREADS, 0, st>>>(patch_ptr((float*)x, stride, b0), n, stride, stats + 2 * b0, ws + b0 * AUG_WS_INTS, kind, sigma,
                                                            philox_of(seed), S);
        else
            k_noise_rng<bf16_t><<<grid, AUG_THREADS, 0, st>>>(patch_ptr((bf16_t*)x, stride, b0), n, stride, stats + 2 * b0, ws + b0 * AUG_WS_INTS, kind, sigma,
                                                             philox_of(seed), S);
    }
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

extern "C" int fmri_shot_noise_rng_batch(void* x, int64_t n, int64_t stride, int B, int dtype, float* stats, int* ws, uint64_t seed, const uint32_t* seqs,
                                         fmri_stream_t stream) {
    if (n < 1 || B < 1 || !x || !stats || !ws || !seqs) return FMRI_E_SHAPE;
    if (dtype != FMRI_F32 && dtype != FMRI_BF16) return FMRI_E_DTYPE;
    hipStream_t st = as_stream(stream);
    FMRI_FOR_CHUNKS(B) {
        const dim3 g0(grid_for(n), nb), g1(grid_for(n, AUG_THREADS, AUG_BLOCKS), nb);
        const SeqB S = seqs_of(seqs, b0, nb);
        float* sp = stats + 2 * b0;
        int* wp = ws + b0 * AUG_WS_INTS;
        if (dtype == FMRI_F32) {
            k_shot_levels<float><<<g0, 256, 0, st>>>(patch_ptr((const float*)x, stride, b0), n, stride, sp, wp, S);
            k_shot_draw<float><<<g1, AUG_THREADS, 0, st>>>(patch_ptr((float*)x, stride, b0), n, stride, sp, wp, philox_of(seed), S);
        } else {
            k_shot_levels<bf16_t><<<g0, 256, 0, st>>>(patch_ptr((const bf16_t*)x, stride, b0), n, stride, sp, wp, S);
            k_shot_draw<bf16_t><<<g1, AUG_THREADS, 0, st>>>(patch_ptr((bf16_t*)x, stride, b0), n, stride, sp, wp, philox_of(seed), S);
        }
    }
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

extern "C" int fmri_elastic_fields_rng_batch(float* d, int X, int Y, int k, const double* weights, const float* alphas, uint64_t seed, const uint32_t* seqs,
                                             int B, fmri_stream_t stream) {
    if (!d || !weights || !alphas || !seqs || B < 1 || X < 1 || X > 65535 || Y < 1 || k < 1 || k > EF_KMAX || (k & 1) == 0) return FMRI_E_SHAPE;
    if ((int64_t)2 * (X + 2 * k) * (Y + 2 * k) > 0x7fffffff) return FMRI_E_SHAPE;
    FMRI_FOR_CHUNKS(B) {
        AlphaB A;
        for (int b = 0; b < AUG_MAXB; ++b) A.v[b] = b < nb ? alphas[b0 + b] : 0.f;
        const dim3 grid((Y + EF_TW - 1) / EF_TW, X, 2 * nb);
        k_elastic_fields_rng<<<grid, 256, 0, as_stream(stream)>>>(d + (int64_t)b0 * 2 * X * Y, X, Y, k, weights, A, philox_of(seed), seqs_of(seqs, b0, nb));
    }
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

extern "C" int fmri_coarse_dropout_rng_batch(void* x, int dtype, int X, int Y, int C, int ld, int64_t stride, int B, const int* grids, int kc, float rate,
                                             const float* stats, uint64_t seed, const uint32_t* seqs, fmri_stream_t stream) {
    if (X < 1 || Y < 1 || C < 1 || ld < C || B < 1 || !grids || (kc != 1 && kc != C) || !x || !stats || !seqs || !(rate >= 0.f && rate <= 1.f))
        return FMRI_E_SHAPE;
    if (dtype != FMRI_F32 && dtype != FMRI_BF16) return FMRI_E_DTYPE;
    hipStream_t st = as_stream(stream);
    FMRI_FOR_CHUNKS(B) {
        GridB G;
        for (int b = 0; b < AUG_MAXB; ++b) {
            G.hs[b] = b < nb ? grids[2 * (b0 + b)] : 1;
            G.ws[b] = b < nb ? grids[2 * (b0 + b) + 1] : 1;
            if (G.hs[b] < 1 || G.ws[b] < 1) return FMRI_E_SHAPE;
        }
        const dim3 grid(grid_for((int64_t)X * Y * C), nb);
        const SeqB S = seqs_of(seqs, b0, nb);
        if (dtype == FMRI_F32)
            k_coarse_dropout_rng<float><<<grid, 256, 0, st>>>(patch_ptr((float*)x, stride, b0), X, Y, C, ld, stride, G, kc, rate, stats + 2 * b0,
                                                             philox_of(seed), S);
        else
            k_coarse_dropout_rng<bf16_t><<<grid, 256, 0, st>>>(patch_ptr((bf16_t*)x, stride, b0), X, Y, C, ld, stride, G, kc, rate, stats + 2 * b0,
                                                              philox_of(seed), S);
    }
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
