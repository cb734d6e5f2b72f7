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
template <typename TI, typename TO>
__global__ void k_affine_sample(const TI* __restrict__ vol, int X, int Y, int Z, Affine34 A, int x0, int y0, int z0, int nx, int ny, int nz,
                                int order, float cval, TO* __restrict__ out, int ld) {
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
                               int order, T* __restrict__ dst, int dst_ld) {
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
        if (threadIdx.x == 0) stats[0] = key2f(~v);
        if (threadIdx.x == 1) stats[1] = key2f(v);
        __threadfence();
        if (threadIdx.x == 0) atomicExch(&w[2], 0u);
    }
}

template <typename T>
__global__ void __launch_bounds__(AUG_THREADS) k_minmax_ws(const T* __restrict__ x, int64_t n, int* __restrict__ ws, float* __restrict__ stats) {
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
__global__ void __launch_bounds__(AUG_THREADS) k_rescale_ws(T* __restrict__ x, int64_t n, float* __restrict__ stats, int* __restrict__ ws, int contrast,
                                                            float lo, float hi, float mult) {
    const float omin = stats[0], omax = stats[1];
    float mn = INFINITY, mx = -INFINITY;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        float v = to_f<T>(x[t]);
        if (contrast) {
            v = fminf(fmaxf(v, lo), hi);
            if (lo != hi) v = (v - lo) / (hi - lo) * (omax - omin) + omin;
            else v = fminf(fmaxf(v, omin), omax);
        }
        const T o = from_f<T>(v * mult);
        x[t] = o;
        const float w = to_f<T>(o);
        mn = fminf(mn, w);
        mx = fmaxf(mx, w);
    }
    __syncthreads();
    emit_minmax(mn, mx, ws, stats);
}

// gaussian (kind 0) / speckle (kind 1) noise as k_noise, the N(0,1) draws by Box-Muller from one Philox block per four elements
template <typename T>
__global__ void __launch_bounds__(AUG_THREADS) k_noise_rng(T* __restrict__ x, int64_t n, float* __restrict__ stats, int* __restrict__ ws, int kind, float sigma,
                                                   Philox rng, unsigned seq) {
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
__device__ __forceinline__ float poisson_draw(float lam, const Philox& rng, unsigned c0, unsigned c1, unsigned seq) {
    if (!(lam > 0.f)) return 0.f;
    unsigned round = 0;
    if (lam < 10.f) {
        const float enlam = __expf(-lam);
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
    const float slam = sqrtf(lam), loglam = logf(lam);
    const float b = 0.931f + 2.53f * slam, a = -0.059f + 0.02483f * b;
    const float invalpha = 1.1239f + 1.1328f / (b - 3.4f), vr = 0.9277f - 3.6224f / (b - 2.f);
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
__global__ void __launch_bounds__(256) k_shot_levels(const T* __restrict__ x, int64_t n, const float* __restrict__ stats, int* __restrict__ ws) {
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
__global__ void __launch_bounds__(AUG_THREADS) k_shot_draw(T* __restrict__ x, int64_t n, float* __restrict__ stats, int* __restrict__ ws, Philox rng, unsigned seq) {
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
    float lo = INFINITY, hi = -INFINITY;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        float s = fminf(fmaxf(to_f<T>(x[t]) * scale + mn, 0.f), 1.f);
        const float lam = floorf(s * 1023.f) / 1023.f * vals;
        const float k = poisson_draw(lam, rng, (unsigned)t, (unsigned)((uint64_t)t >> 32), seq);
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
__global__ void __launch_bounds__(256) k_elastic_fields_rng(float* __restrict__ d0, float* __restrict__ d1, int X, int Y, int k,
                                                            const double* __restrict__ w, float alpha, Philox rng, unsigned seq) {
    __shared__ float nz[EF_KMAX][EF_TW + EF_KMAX - 1];
    __shared__ float rb[EF_KMAX][EF_TW];
    __shared__ float wf[EF_KMAX];
    const int jt = blockIdx.x * EF_TW, i = blockIdx.y, f = blockIdx.z;
    const int hp = X + 2 * k, wp = Y + 2 * k, r = k / 2;
    const int tw = min(EF_TW, Y - jt), span = tw + k - 1;
    if (threadIdx.x < k) wf[threadIdx.x] = (float)w[threadIdx.x];
    const int row0 = i + k - r, col0 = jt + k - r;
    for (int idx = threadIdx.x; idx < k * span; idx += blockDim.x) {
        const int a = idx / span, b = idx - a * span;
        const unsigned pix = (unsigned)((f * hp + row0 + a) * wp + col0 + b);
        const uint4 q = rng(pix >> 2, 0u, seq, 1u);
        const unsigned sel = pix & 3u;
        const unsigned u = sel == 0 ? q.x : (sel == 1 ? q.y : (sel == 2 ? q.z : q.w));
        nz[a][b] = u01(u) * 2.f - 1.f;
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
__global__ void k_coarse_dropout_rng(T* __restrict__ x, int X, int Y, int C, int ld, int hs, int wsz, int kc, float rate, const float* __restrict__ stats,
                                     Philox rng, unsigned seq) {
    const int64_t total = (int64_t)X * Y * C;
    const double fi = (double)hs / X, fj = (double)wsz / Y;
    const float lo = stats[0];
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % C);
        const int64_t ij = t / C;
        const int j = (int)(ij % Y), i = (int)(ij / Y);
        const int si = min((int)floor(i * fi), hs - 1), sj = min((int)floor(j * fj), wsz - 1);
        const unsigned cell = (unsigned)((si * wsz + sj) * kc + (kc == 1 ? 0 : c));
        if (u01(rng(cell, 0u, seq, 0u).x) < rate) st_any<T>(x, ij * ld + c, lo);
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

extern "C" int fmri_elastic_warp(const void* src, int dtype, int X, int Y, int C, int src_ld, const float* d0, const float* d1, int order, void* dst,
                                int dst_ld, fmri_stream_t stream) {
    if (X < 1 || Y < 1 || C < 1 || src_ld < C || dst_ld < C || (order != 0 && order != 1) || !d0 || !d1 || src == dst) return FMRI_E_SHAPE;
    hipStream_t st = as_stream(stream);
    const int grid = grid_for((int64_t)X * Y * C);
    if (dtype == FMRI_F32) k_elastic_warp<float><<<grid, 256, 0, st>>>((const float*)src, X, Y, C, src_ld, d0, d1, order, (float*)dst, dst_ld);
    else if (dtype == FMRI_U8) k_elastic_warp<uint8_t><<<grid, 256, 0, st>>>((const uint8_t*)src, X, Y, C, src_ld, d0, d1, order, (uint8_t*)dst, dst_ld);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
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

extern "C" int fmri_affine_sample(const void* vol, int vol_dtype, int X, int Y, int Z, const double* affine, int x0, int y0, int z0, int nx,
                                  int ny, int nz, int order, float cval, void* out, int out_dtype, int out_ld, fmri_stream_t stream) {
    if (X < 1 || Y < 1 || Z < 1 || nx < 1 || ny < 1 || nz < 1 || out_ld < nz || (order != 0 && order != 1) || !affine) return FMRI_E_SHAPE;
    Affine34 A;
    for (int i = 0; i < 12; ++i) A.a[i] = affine[i];
    const int64_t total = (int64_t)nx * ny * nz;
    const int grid = grid_for(total);
    hipStream_t st = as_stream(stream);
#define FMRI_AS(TI, TO) \
    k_affine_sample<TI, TO><<<grid, 256, 0, st>>>((const TI*)vol, X, Y, Z, A, x0, y0, z0, nx, ny, nz, order, cval, (TO*)out, out_ld)
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

// ---- the same intensity steps with in-kernel Philox draws and chained min / max (see the kernels' comment block above)
static inline Philox philox_of(uint64_t seed) {
    Philox p;
    p.k0 = (unsigned)seed;
    p.k1 = (unsigned)(seed >> 32);
    return p;
}

extern "C" int fmri_minmax_ws(const void* x, int64_t n, int dtype, float* stats, int* ws, fmri_stream_t stream) {
    if (n < 1 || !x || !stats || !ws) return FMRI_E_SHAPE;
    hipStream_t st = as_stream(stream);
    const int grid = grid_for((n + 3) / 4, AUG_THREADS, AUG_BLOCKS);
    if (dtype == FMRI_F32) k_minmax_ws<float><<<grid, AUG_THREADS, 0, st>>>((const float*)x, n, ws, stats);
    else if (dtype == FMRI_BF16) k_minmax_ws<bf16_t><<<grid, AUG_THREADS, 0, st>>>((const bf16_t*)x, n, ws, stats);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

extern "C" int fmri_rescale_intensity_ws(void* x, int64_t n, int dtype, float* stats, int* ws, int contrast, float lo, float hi, float mult,
                                         fmri_stream_t stream) {
    if (n < 1 || !x || !stats || !ws) return FMRI_E_SHAPE;
    hipStream_t st = as_stream(stream);
    const int grid = grid_for(n, AUG_THREADS, AUG_BLOCKS);
    if (dtype == FMRI_F32) k_rescale_ws<float><<<grid, AUG_THREADS, 0, st>>>((float*)x, n, stats, ws, contrast, lo, hi, mult);
    else if (dtype == FMRI_BF16) k_rescale_ws<bf16_t><<<grid, AUG_THREADS, 0, st>>>((bf16_t*)x, n, stats, ws, contrast, lo, hi, mult);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

extern "C" int fmri_noise_rng(void* x, int64_t n, int dtype, float* stats, int* ws, int kind, float sigma, uint64_t seed, uint32_t seq,
                              fmri_stream_t stream) {
    if (n < 1 || (kind != 0 && kind != 1) || !x || !stats || !ws) return FMRI_E_SHAPE;
    hipStream_t st = as_stream(stream);
    const int grid = grid_for((n + 3) / 4, AUG_THREADS, AUG_BLOCKS);
    if (dtype == FMRI_F32) k_noise_rng<float><<<grid, AUG_THREADS, 0, st>>>((float*)x, n, stats, ws, kind, sigma, philox_of(seed), seq);
    else if (dtype == FMRI_BF16) k_noise_rng<bf16_t><<<grid, AUG_THREADS, 0, st>>>((bf16_t*)x, n, stats, ws, kind, sigma, philox_of(seed), seq);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

extern "C" int fmri_shot_noise_rng(void* x, int64_t n, int dtype, float* stats, int* ws, uint64_t seed, uint32_t seq, fmri_stream_t stream) {
    if (n < 1 || !x || !stats || !ws) return FMRI_E_SHAPE;
    hipStream_t st = as_stream(stream);
    const int g0 = grid_for(n), g1 = grid_for(n, AUG_THREADS, AUG_BLOCKS);
    if (dtype == FMRI_F32) {
        k_shot_levels<float><<<g0, 256, 0, st>>>((const float*)x, n, stats, ws);
        k_shot_draw<float><<<g1, AUG_THREADS, 0, st>>>((float*)x, n, stats, ws, philox_of(seed), seq);
    } else if (dtype == FMRI_BF16) {
        k_shot_levels<bf16_t><<<g0, 256, 0, st>>>((const bf16_t*)x, n, stats, ws);
        k_shot_draw<bf16_t><<<g1, AUG_THREADS, 0, st>>>((bf16_t*)x, n, stats, ws, philox_of(seed), seq);
    } else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

extern "C" int fmri_elastic_fields_rng(float* d0, float* d1, int X, int Y, int k, const double* weights, float alpha, uint64_t seed, uint32_t seq,
                                       fmri_stream_t stream) {
    if (!d0 || !d1 || !weights || X < 1 || Y < 1 || k < 1 || k > EF_KMAX || (k & 1) == 0) return FMRI_E_SHAPE;
    if ((int64_t)2 * (X + 2 * k) * (Y + 2 * k) > 0x7fffffff) return FMRI_E_SHAPE;
    const dim3 grid((Y + EF_TW - 1) / EF_TW, X, 2);
    k_elastic_fields_rng<<<grid, 256, 0, as_stream(stream)>>>(d0, d1, X, Y, k, weights, alpha, philox_of(seed), seq);
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

extern "C" int fmri_coarse_dropout_rng(void* x, int dtype, int X, int Y, int C, int ld, int hs, int ws_, int kc, float rate, const float* stats,
                                       uint64_t seed, uint32_t seq, fmri_stream_t stream) {
    if (X < 1 || Y < 1 || C < 1 || ld < C || hs < 1 || ws_ < 1 || (kc != 1 && kc != C) || !x || !stats || !(rate >= 0.f && rate <= 1.f))
        return FMRI_E_SHAPE;
    hipStream_t st = as_stream(stream);
    const int grid = grid_for((int64_t)X * Y * C);
    if (dtype == FMRI_F32) k_coarse_dropout_rng<float><<<grid, 256, 0, st>>>((float*)x, X, Y, C, ld, hs, ws_, kc, rate, stats, philox_of(seed), seq);
    else if (dtype == FMRI_BF16)
        k_coarse_dropout_rng<bf16_t><<<grid, 256, 0, st>>>((bf16_t*)x, X, Y, C, ld, hs, ws_, kc, rate, stats, philox_of(seed), seq);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
