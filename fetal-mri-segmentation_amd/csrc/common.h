// Shared device helpers for the gfx950 kernels (bf16 storage helpers, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/fmri_hip.h"

typedef unsigned short bf16_t;  // raw bf16 bits in memory

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {  // round-to-nearest-even; NaN stays NaN (v_cvt_pk_bf16_f32)
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}

template <typename T> struct Io;
template <> struct Io<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Io<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// ---- bit-reproducible gradient accumulation (round 3).  Floating-point atomics add a workgroup's partial sums in whatever order the
// workgroups finish: the default training step is not reproducible bit for bit.  Integer adds are associative, so in DETERMINISTIC mode every
// partial sum that would go to a gradient element by `atomicAdd(float)` is instead rounded to a 2^-40 fixed-point number (|v| < 2^23, which
// gradients are) and added with a 64-bit integer atomic to a SHADOW of the gradient buffer; fmri_deterministic_finish() then adds the shadow
// (converted back once) to the fp32 gradient and clears it.  The partial sums themselves come out of fixed-order MFMA / FMA chains and the
// work partition of every launch is static, so the result no longer depends on the arrival order.  fmri_set_deterministic() registers
// (gradient base, shadow base, element count); each translation unit that flushes gradients keeps its own copy of the registration (the
// library is built without relocatable device code).  Cost: 64-bit atomics, i.e. twice the flush traffic; the default mode is untouched
// apart from one scalar load and a uniform branch per flush.
struct FmriDetCfg {
    float* base;
    unsigned long long* shadow;
    long long n;
};
#define FMRI_DET_SCALE 1099511627776.f        /* 2^40 */
#define FMRI_DET_TU(tag)                                                                                         \
    static __device__ FmriDetCfg g_det_cfg;                                                                      \
    static bool h_det_on = false;                                                                                \
    int fmri_det_set_##tag(const FmriDetCfg& c) {                                                                \
        h_det_on = c.base != nullptr;                                                                            \
        return hipMemcpyToSymbol(HIP_SYMBOL(g_det_cfg), &c, sizeof(c)) == hipSuccess ? FMRI_OK : FMRI_E_LAUNCH;  \
    }
__device__ __forceinline__ void fmri_grad_add(const FmriDetCfg& c, float* p, float v) {
    const long long i = p - c.base;
    if (c.base != nullptr && i >= 0 && i < c.n)
        atomicAdd(c.shadow + i, (unsigned long long)__float2ll_rn(v * FMRI_DET_SCALE));
    else
        atomicAdd(p, v);
}

#define FMRI_LAUNCH_CHECK()                                   \
    do {                                                      \
        if (hipGetLastError() != hipSuccess) return FMRI_E_LAUNCH; \
    } while (0)

static inline hipStream_t as_stream(fmri_stream_t s) { return (hipStream_t)s; }

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- vector load/store of VEC contiguous elements (address must be VEC*sizeof(T)-aligned) ------------------------
template <typename T, int VEC> struct alignas(sizeof(T) * VEC) PackT { T e[VEC]; };

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16_t>(bf16_t v) { return bf2f(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return f2bf(v); }

template <typename T, int VEC> __device__ __forceinline__ void ldv(const T* p, float* out) {
    PackT<T, VEC> v = *reinterpret_cast<const PackT<T, VEC>*>(p);
#pragma unroll
    for (int k = 0; k < VEC; ++k) out[k] = to_f<T>(v.e[k]);
}
template <typename T, int VEC> __device__ __forceinline__ void stv(T* p, const float* in) {
    PackT<T, VEC> v;
#pragma unroll
    for (int k = 0; k < VEC; ++k) v.e[k] = from_f<T>(in[k]);
    *reinterpret_cast<PackT<T, VEC>*>(p) = v;
}

static inline int pick_vec(int C, int maxvec = 8) {
    int v = maxvec;
    while (v > 1 && (C % v)) v >>= 1;
    return v;
}
static inline int grid_for(int64_t total, int block = 256, int cap = 256 * 16) {
    int64_t g = ceil_div64(total, block);
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

#define LAUNCH_TV(KERN, T, vec, grid, block, s, ...)                                 \
    do {                                                                             \
        switch (vec) {                                                               \
            case 8: KERN<T, 8><<<grid, block, 0, s>>>(__VA_ARGS__); break;           \
            case 4: KERN<T, 4><<<grid, block, 0, s>>>(__VA_ARGS__); break;           \
            case 2: KERN<T, 2><<<grid, block, 0, s>>>(__VA_ARGS__); break;           \
            default: KERN<T, 1><<<grid, block, 0, s>>>(__VA_ARGS__); break;          \
        }                                                                            \
    } while (0)

// ---- 64 x 64 weight tile: rows = output channels, columns = input channels.  src(co, ci) -> fp32 value; the tile is written once
// row-major to f_dst[co * ld_f + ci] (coalesced along ci) and once transposed to d_dst[ci * ld_d + co] (coalesced along co) through LDS.
// Round 1 wrote the transposed image with one scattered 2-byte store per element (stride Cout): 8x write amplification in HBM
// (rocprofv3 WRITE_SIZE 240 MB for 32 MB of images) and ~1 ms per optimizer step for what is 130 MB of traffic.
// Call with 256 threads; `tile` = T[64][PACK_PITCH(T)] in LDS; ends with a barrier (the tile can be reused at once).
#define PACK_PITCH(T) (64 + (sizeof(T) == 2 ? 2 : 1))       /* odd number of 4-byte words per row: the column reads are conflict-free */
template <typename T, typename F>
__device__ __forceinline__ void pack_tile(F src, T* __restrict__ f_dst, int64_t ld_f, T* __restrict__ d_dst, int64_t ld_d, int co0, int ci0,
                                          int Cout, int Cin, T (*tile)[PACK_PITCH(T)]) {
    const int x = threadIdx.x & 63, y = threadIdx.x >> 6;
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
        const int col = k * 4 + y, co = co0 + col, ci = ci0 + x;
        if (co < Cout && ci < Cin) {
            const T v = from_f<T>(src(co, ci));
            if (f_dst) f_dst[(int64_t)co * ld_f + ci] = v;
            tile[col][x] = v;
        }
    }
    __syncthreads();
    if (d_dst) {
#pragma unroll 4
        for (int k = 0; k < 16; ++k) {
            const int row = k * 4 + y, ci = ci0 + row, co = co0 + x;
            if (co < Cout && ci < Cin) d_dst[(int64_t)ci * ld_d + co] = tile[x][row];
        }
    }
    __syncthreads();
}

// one workgroup of fmri_conv3d_pack_weights: block b = (tap, Cout tile, Cin tile) of the fp32 master [27][Cout][Cin] -> the forward image
// (same layout) and the tap-flipped transposed input-gradient image [26 - tap][Cin][Cout]
template <typename T>
__device__ __forceinline__ void pack_plain_block(int b, const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd, int Cout, int Cin,
                                                 T (*tile)[PACK_PITCH(T)]) {
    const int tci = (Cin + 63) >> 6, tco = (Cout + 63) >> 6;
    const int ci0 = (b % tci) << 6; b /= tci;
    const int co0 = (b % tco) << 6;
    const int tap = b / tco;
    const float* const wt = w + (int64_t)tap * Cout * Cin;
    pack_tile<T>([&](int co, int ci) { return wt[(int64_t)co * Cin + ci]; }, wf ? wf + (int64_t)tap * Cout * Cin : nullptr, Cin,
                 wd ? wd + (int64_t)(26 - tap) * Cin * Cout : nullptr, Cout, co0, ci0, Cout, Cin, tile);
}

