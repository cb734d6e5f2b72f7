// First-layer specialisation: single-channel input (the MRI volume), Cout a multiple of 32, bf16.
//
// With Cin = 1 the 3x3x3 convolution is a [voxels x 27] x [27 x Cout] product: the 27 taps ARE the contraction
// dimension (padded to 32 = two k-steps of v_mfma_f32_32x32x16_bf16).  The "im2col" operand is never built in memory:
// each lane gathers its 8 tap values straight from a bf16 halo tile of x in LDS.  Both kernels are HBM-bound
// (forward writes the 32-channel activation, the weight gradient reads dy once), so the point of MFMA here is only to
// get the arithmetic out of the way of the memory stream - and so is everything else in the file: a halo tile's global
// loads are all in flight before the first LDS write, the forward kernel's 16 column tiles per wave are unrolled so that
// every gather address is register + immediate, ReLU is a template parameter, the weight gradient loads its next tile
// while it multiplies the current one and gets the bias gradient from a padding column of the same product
// (round 6, 4 x 64x128x128: forward 77 -> 56 us, weight gradient 89 -> 59 us; 5 slices x 64 x 256x256: 194 -> 77, 161 -> 102).
//
// Reference ops replaced: the first Conv3D(+BiasAdd+Relu) of unet_model_3d (unet3d/unet.py:45-46,102,113) and its
// Conv3DBackpropFilterV2 / BiasAddGrad.
#include "common.h"

FMRI_DET_TU(first)

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// max as ONE v_max_f32 (fmaxf adds a canonicalising v_max x, x); a NaN operand yields the other one, like fmaxf
__device__ __forceinline__ float vmax1(float a, float b) {
    float o;
    asm("v_max_f32 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b));
    return o;
}
// two floats -> one dword of bf16, round to nearest even (one v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned pack2(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

// ---------------------------------------------------------------------------------------------------------- forward
namespace ff {
constexpr int TD = 4, TH = 16, TW = 32;                 // 2048 voxels per workgroup = 64 column tiles of 32 (one w-row each)
constexpr int HH = TH + 2, HW = TW + 2;                 // (6 x) 18 x 34 halo of bf16 scalars per input channel
constexpr int NTHREADS = 256;
}  // namespace ff
// LDS row of a halo tile (both kernels): column c (0 = w0 - 1 ... 33 = w0 + 32) sits at element (XO + c) * CIN of a PW * CIN-element row, so
// that the 32 interior columns - 64 * CIN contiguous, 16-byte aligned bytes in memory - are 16-byte aligned in LDS as well and move as 16-byte
// pieces.  (The weight gradient's B fragment reads nine different (kd, kh) rows at once; a pitch of 18 * CIN dwords, which starts them on
// nine different banks, measured the same as this one.)
constexpr int XO = 7, PW = 48;


// The contraction index k enumerates (tap, input channel): k = tap * CIN + c, tap = kd*9 + kh*3 + kw (planar: kh*3 + kw, centre kd only).
// Element offset of k inside a [hd][hh][hw][CIN] halo tile; k beyond the last real one is zero-weighted padding (any in-tile address).
template <int CIN, bool PLANAR> __device__ __forceinline__ int k_off(int k, int HHs, int HWs) {
    constexpr int K = (PLANAR ? 9 : 27) * CIN;
    k = k >= K ? K - 1 : k;
    const int tap = k / CIN, c = k % CIN;
    const int kd = PLANAR ? 0 : tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
    return ((kd * HHs + kh) * HWs + kw) * CIN + c;
}
// index of (k, co) in the packed filter [27][Cout][CIN]
template <int CIN, bool PLANAR> __device__ __forceinline__ int64_t k_widx(int k, int co, int Cout) {
    const int tap = k / CIN, c = k % CIN;
    return ((int64_t)((PLANAR ? 9 : 0) + tap) * Cout + co) * CIN + c;
}

// waves per SIMD the register allocation aims at: 2 (no instantiation of the benchmarked configurations spills there; at 3-4 the 5-slice 2-D
// kernel spilled and ran 100-128 us instead of 78), 1 for the 3- and 4-modality 3-D kernels (6-7 k-steps of gather pointers)
template <int CIN, bool PLANAR> constexpr int first_fwd_waves() { return ((PLANAR ? 9 : 27) * CIN + 15) / 16 > 4 ? 1 : 2; }

template <int CIN, bool PLANAR, bool RELU>
__global__ void __launch_bounds__(ff::NTHREADS, (first_fwd_waves<CIN, PLANAR>()))
k_conv_first_fwd(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wt /*[27][Cout][CIN]*/, const float* __restrict__ bias,
                 bf16_t* __restrict__ y, int N, int D, int H, int W, int Cout, int act, float alpha) {
    using namespace ff;
    constexpr int K = (PLANAR ? 9 : 27) * CIN, KS = (K + 15) / 16, HD = PLANAR ? TD : TD + 2, DOFF = PLANAR ? 0 : 1;
    __shared__ __attribute__((aligned(16))) bf16_t sx[HD * HH * PW * CIN];
    int tile = blockIdx.x;
    const int twn = W / TW, thn = H / TH, tdn = D / TD;
    const int w0 = (tile % twn) * TW; tile /= twn;
    const int h0 = (tile % thn) * TH; tile /= thn;
    const int d0 = (tile % tdn) * TD;
    const int n = tile / tdn;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, r = lane & 31, hk = lane >> 5;

    // halo tile: every load of a thread is in flight before the first LDS write (a loop of load - wait - write pairs cost a workgroup 15 memory
    // latencies).  Interior columns: 4 * CIN uint4 per (d, h) row; the two edge columns: CIN scalars each.
    {
        constexpr int ROWS = HD * HH, NV = ROWS * 4 * CIN, NVI = (NV + NTHREADS - 1) / NTHREADS;
        constexpr int NE = ROWS * 2 * CIN, NEI = (NE + NTHREADS - 1) / NTHREADS;
        uint4 v[NVI];
        bf16_t e[NEI];
#pragma unroll
        for (int k = 0; k < NVI; ++k) {
            const int i = t + k * NTHREADS, row = i / (4 * CIN), seg = i % (4 * CIN);
            const int gd = d0 - DOFF + row / HH, gh = h0 - 1 + row % HH;
            v[k] = make_uint4(0, 0, 0, 0);
            if (i < NV && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H)
                v[k] = *reinterpret_cast<const uint4*>(x + ((((int64_t)n * D + gd) * H + gh) * W + w0) * CIN + seg * 8);
        }
#pragma unroll
        for (int k = 0; k < NEI; ++k) {
            const int i = t + k * NTHREADS, c = i % CIN, side = (i / CIN) & 1, row = i / (2 * CIN);
            const int gd = d0 - DOFF + row / HH, gh = h0 - 1 + row % HH, gw = side ? w0 + TW : w0 - 1;
            e[k] = 0;
            if (i < NE && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W)
                e[k] = x[((((int64_t)n * D + gd) * H + gh) * W + gw) * CIN + c];
        }
#pragma unroll
        for (int k = 0; k < NVI; ++k) {
            const int i = t + k * NTHREADS, row = i / (4 * CIN), seg = i % (4 * CIN);
            if (i < NV) *reinterpret_cast<uint4*>(sx + (row * PW + XO + 1) * CIN + seg * 8) = v[k];
        }
#pragma unroll
        for (int k = 0; k < NEI; ++k) {
            const int i = t + k * NTHREADS, c = i % CIN, side = (i / CIN) & 1, row = i / (2 * CIN);
            if (i < NE) sx[(row * PW + XO + (side ? HW - 1 : 0)) * CIN + c] = e[k];
        }
    }
    // The wave's 16 column tiles are (dl, hl = wv + 4 i): lane pointers to the gathered elements of column tile (0, wv), k = 16 ks + 8 hk + j;
    // every other column tile adds a compile-time constant - the read's immediate offset - so the unrolled loop has no address arithmetic.
    const bf16_t* gp[KS][8];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) gp[ks][j] = sx + (wv * PW + XO + r) * CIN + k_off<CIN, PLANAR>(16 * ks + 8 * hk + j, HH, PW);
    const int64_t s_h = (int64_t)4 * W * Cout, s_d = (int64_t)H * W * Cout;       // output strides of i and dl, elements
    bf16_t* const y0 = y + ((((int64_t)n * D + d0) * H + h0 + wv) * W + w0 + r) * Cout + hk * 8;

    for (int cot = 0; cot < Cout / 32; ++cot) {
        // A = W^T[co = r][k], zero for k >= K
        bf16x8_t a[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            u16x8 u;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 16 * ks + 8 * hk + j;
                u[j] = k < K ? wt[k_widx<CIN, PLANAR>(k, cot * 32 + r, Cout)] : (bf16_t)0;
            }
            a[ks] = __builtin_bit_cast(bf16x8_t, u);
        }
        float bv[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) bv[k] = bias ? bias[cot * 32 + 8 * (k >> 2) + 4 * hk + (k & 3)] : 0.f;
        if (cot == 0) __syncthreads();          // the halo tile; the filter and bias loads above are in flight beside its loads

#pragma unroll
        for (int dl = 0; dl < TD; ++dl)
#pragma unroll
            for (int i = 0; i < TH / 4; ++i) {
                const int imm = ((dl * HH + 4 * i) * PW) * CIN;
                f32x16 acc;
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[k] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    u16x8 u;
#pragma unroll
                    for (int j = 0; j < 8; ++j) u[j] = gp[ks][j][imm];
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], __builtin_bit_cast(bf16x8_t, u), acc, 0, 0, 0);
                }
                // D rows = channel (reg&3) + 8*(reg>>2) + 4*hk of voxel r: a lane owns 4-channel (8-byte) pieces, and storing those directly wrote
                // a quarter of a 32-byte sector per lane (1.8 TB/s in round 1).  A half-wave exchange (v_permlane32_swap) gives every lane 8
                // consecutive channels of voxel (lane & 31): two 16-byte stores per lane, each instruction filling whole 32-byte sectors of 32
                // consecutive voxel rows.  ReLU is its own instantiation (RELU): with the activation a run-time value the compiler kept 32 scalar
                // branches per column tile.
                float o[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) o[k] = acc[k] + bv[k];
                if (RELU) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) o[k] = vmax1(o[k], 0.f);
                } else if (act == FMRI_ACT_LEAKY) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) o[k] = o[k] > 0.f ? o[k] : alpha * o[k];
                }
                unsigned pk[4][2];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    pk[g][0] = pack2(o[4 * g], o[4 * g + 1]);
                    pk[g][1] = pack2(o[4 * g + 2], o[4 * g + 3]);
                }
                bf16_t* const yp = y0 + dl * s_d + i * s_h + cot * 32;
#pragma unroll
                for (int pq = 0; pq < 2; ++pq) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        auto sw = __builtin_amdgcn_permlane32_swap(pk[2 * pq][q], pk[2 * pq + 1][q], false, false);
                        pk[2 * pq][q] = sw[0];
                        pk[2 * pq + 1][q] = sw[1];
                    }
                    // lanes 0-31 now hold channels 16pq .. 16pq+7 of voxel r, lanes 32-63 channels 16pq+8 .. 16pq+15 of the same voxel
                    *reinterpret_cast<uint4*>(yp + pq * 16) = make_uint4(pk[2 * pq][0], pk[2 * pq][1], pk[2 * pq + 1][0], pk[2 * pq + 1][1]);
                }
                // many k-steps: column tiles are not interleaved by the scheduler (it hoists the gathers of several tiles and spills)
                if (KS > 3) __builtin_amdgcn_sched_barrier(0);
            }
    }
}

// ---------------------------------------------------------------------------------------------------------- weight gradient
namespace fg {
constexpr int TD = 2, TH = 8, TW = 32;                  // 512 voxels of dy per tile = 32 k-steps of 16 voxels
constexpr int HH = TH + 2, HW = TW + 2;                 // x halo (4 x) 10 x 34 per input channel
constexpr int YVOX = TD * TH * TW;
constexpr int NTHREADS = 256;
}  // namespace fg

// dw[tap][co][c] += sum_v dy[v][co] * x[v+tap][c];  db[co] += sum_v dy[v][co].  One 32-wide co tile per blockIdx.y; the (tap, c)
// pairs are the N dimension of the product (NCT column tiles of 32).  The bias gradient is column K of the same product - the first of the
// padding columns, whose operand is a row of ones.
template <int CIN, bool PLANAR>
__global__ void __launch_bounds__(fg::NTHREADS)
k_conv_first_wgrad(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, float* __restrict__ dw, float* __restrict__ db,
                   int N, int D, int H, int W, int Cout) {
    using namespace fg;
    constexpr int K = (PLANAR ? 9 : 27) * CIN, NCT = (K + 31) / 32, HD = PLANAR ? TD : TD + 2, DOFF = PLANAR ? 0 : 1;
    static_assert(K % 32 != 0, "the bias gradient needs a padding column");
    constexpr int XE = HD * HH * PW * CIN;                              // halo elements (row layout: XO, PW)
    constexpr int ONES = XE;                                            // 8 * CIN elements of 1.0 behind the halo
    constexpr int XB = (XE + 8 * CIN) * 2;
    constexpr int LB = (YVOX * 64 + XB + 16) > (NCT * 32 * 32 * 4 + 16) ? (YVOX * 64 + XB + 16) : (NCT * 32 * 32 * 4 + 16);
    __shared__ __attribute__((aligned(16))) unsigned char lds[LB];
    unsigned char* const lds_y = lds;                                  // [voxel][32 co] 64-B rows
    bf16_t* const sx = reinterpret_cast<bf16_t*>(lds + YVOX * 64);     // halo scalars
    const int cot = blockIdx.y;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, r = lane & 31, hk = lane >> 5;
    const int twn = W / TW, thn = H / TH, tdn = D / TD;
    const int ntiles = N * tdn * thn * twn;
    // B operand: lane r = column (tap, c) of column tile ct, 8 consecutive voxels (k = 8*hk + j) along w
    int toff[NCT];
    f32x16 acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        toff[ct] = k_off<CIN, PLANAR>(ct * 32 + r, HH, PW) + (XO + 8 * hk) * CIN;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[ct][k] = 0.f;
    }
    const bool ones_lane = r == K % 32;                                 // column K of column tile K / 32: reads the ones, wherever the k-step is
    if (ones_lane) toff[K / 32] = ONES;

    // a tile's global loads, all in flight at once: dy 512 rows x 4 slots of 16 B; x halo: interior columns as uint4, edge columns as scalars
    constexpr int ROWS = HD * HH, NV = ROWS * 4 * CIN, NVI = (NV + NTHREADS - 1) / NTHREADS;
    constexpr int NE = ROWS * 2 * CIN, NEI = (NE + NTHREADS - 1) / NTHREADS;
    u32x4 vy[8], vx[NVI];            // (native vectors: arrays of HIP's uint4 struct carried around the loop stayed in scratch memory)
    bf16_t ex[NEI];
#define FIRST_LOAD_TILE(TILE_)                                                                                                            \
    {                                                                                                                                     \
        int q = (TILE_);                                                                                                                  \
        const int w0 = (q % twn) * TW; q /= twn;                                                                                          \
        const int h0 = (q % thn) * TH; q /= thn;                                                                                          \
        const int d0 = (q % tdn) * TD;                                                                                                    \
        const int n = q / tdn;                                                                                                            \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) {                                                                                   \
            const int i = t + k * NTHREADS;                                                                                               \
            const int row = i >> 2, ps = i & 3;                                                                                           \
            const int yw = row & 31, yh = (row >> 5) & 7, yd = row >> 8;                                                                  \
            vy[k] = *reinterpret_cast<const u32x4*>(dy + ((((int64_t)n * D + d0 + yd) * H + h0 + yh) * W + w0 + yw) * Cout + cot * 32 + ps * 8); \
        }                                                                                                                                 \
        _Pragma("unroll") for (int k = 0; k < NVI; ++k) {                                                                                 \
            const int i = t + k * NTHREADS, row = i / (4 * CIN), seg = i % (4 * CIN);                                                     \
            const int gd = d0 - DOFF + row / HH, gh = h0 - 1 + row % HH;                                                                  \
            vx[k] = u32x4{0, 0, 0, 0};                                                                                                      \
            if (i < NV && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H)                                                       \
                vx[k] = *reinterpret_cast<const u32x4*>(x + ((((int64_t)n * D + gd) * H + gh) * W + w0) * CIN + seg * 8);                 \
        }                                                                                                                                 \
        _Pragma("unroll") for (int k = 0; k < NEI; ++k) {                                                                                 \
            const int i = t + k * NTHREADS, c = i % CIN, side = (i / CIN) & 1, row = i / (2 * CIN);                                       \
            const int gd = d0 - DOFF + row / HH, gh = h0 - 1 + row % HH, gw = side ? w0 + TW : w0 - 1;                                    \
            ex[k] = 0;                                                                                                                    \
            if (i < NE && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W)                         \
                ex[k] = x[((((int64_t)n * D + gd) * H + gh) * W + gw) * CIN + c];                                                         \
        }                                                                                                                                 \
    }
#define FIRST_STORE_TILE()                                                                                                                \
    {                                                                                                                                     \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) *reinterpret_cast<u32x4*>(lds_y + (t + k * NTHREADS) * 16) = vy[k];                 \
        _Pragma("unroll") for (int k = 0; k < NVI; ++k) {                                                                                 \
            const int i = t + k * NTHREADS, row = i / (4 * CIN), seg = i % (4 * CIN);                                                     \
            if (i < NV) *reinterpret_cast<u32x4*>(sx + (row * PW + XO + 1) * CIN + seg * 8) = vx[k];                                      \
        }                                                                                                                                 \
        _Pragma("unroll") for (int k = 0; k < NEI; ++k) {                                                                                 \
            const int i = t + k * NTHREADS, c = i % CIN, side = (i / CIN) & 1, row = i / (2 * CIN);                                       \
            if (i < NE) sx[(row * PW + XO + (side ? HW - 1 : 0)) * CIN + c] = ex[k];                                                      \
        }                                                                                                                                 \
    }
    if (t < 8 * CIN) sx[ONES + t] = 0x3F80;
    int tile = blockIdx.x;
    if (tile < ntiles) FIRST_LOAD_TILE(tile)
    for (; tile < ntiles; tile += gridDim.x) {
        FIRST_STORE_TILE()
        __syncthreads();
        // the next tile's loads travel while this one is multiplied
        if (tile + (int)gridDim.x < ntiles) FIRST_LOAD_TILE(tile + gridDim.x)
        // 32 k-steps: (dl, hl, half of the 32-wide row); each wave takes 8
#pragma unroll 2
        for (int ks = wv; ks < 32; ks += 4) {
            const int wh = ks & 1, hl = (ks >> 1) & 7, dl = ks >> 4;
            const int row0 = (dl * TH + hl) * TW + 16 * wh;
            // A[co][k = voxel] through the transposing read (64-B rows, no swizzle needed)
            const int g = lane >> 4, qd = (lane & 15) >> 2, p = lane & 3;
            const unsigned char* pa = lds_y + (row0 + 8 * (g >> 1) + qd) * 64 + 32 * (g & 1) + 8 * p;
            s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pa);
            s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pa + 4 * 64));
            u16x8 au;
#pragma unroll
            for (int j = 0; j < 4; ++j) { au[j] = (unsigned short)v0[j]; au[4 + j] = (unsigned short)v1[j]; }
            const int xb = ((dl * HH + hl) * PW + 16 * wh) * CIN;
            const int xb1 = ones_lane ? 0 : xb;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                u16x8 bu;
#pragma unroll
                for (int j = 0; j < 8; ++j) bu[j] = sx[(ct == K / 32 ? xb1 : xb) + toff[ct] + j * CIN];
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, au), __builtin_bit_cast(bf16x8_t, bu), acc[ct], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // D rows = co, cols = (tap, c) and the bias column: reduce the four waves in LDS, then ONE global atomic per element and workgroup
    float* const red = reinterpret_cast<float*>(lds);          // [NCT*32 columns][32 co], tile buffers are dead now
    for (int i = t; i < NCT * 1024; i += NTHREADS) red[i] = 0.f;
    __syncthreads();
    // the four waves add their accumulators ONE AFTER THE OTHER (each lane owns its elements within a wave): a fixed order, so the
    // workgroup's sums are reproducible bit for bit (LDS float atomics from four waves were not)
    for (int w = 0; w < 4; ++w) {
        if (wv == w) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int col = (reg & 3) + 8 * (reg >> 2) + 4 * hk;
                    red[(ct * 32 + r) * 32 + col] += acc[ct][reg];
                }
        }
        __syncthreads();
    }
    const FmriDetCfg dc = g_det_cfg;
    for (int i = t; i < K * 32; i += NTHREADS) fmri_grad_add(dc, &dw[k_widx<CIN, PLANAR>(i >> 5, cot * 32 + (i & 31), Cout)], red[i]);
    if (db && t < 32) fmri_grad_add(dc, &db[cot * 32 + t], red[K * 32 + t]);
}

}  // namespace

// single-channel 3-D volumes, and the few-slice stacks of the 2-D models (reference config_utils.py:53-56: 5 slices by default)
bool conv3d_first_ok(int C0, int C1, int Cout, int D, int H, int W, int dtype, int up0, int planar) {
    if (dtype != FMRI_BF16 || C1 != 0 || up0 || (Cout % 32) || (D % 4) || (H % 16) || (W % 32)) return false;
    return planar ? (C0 == 1 || C0 == 3 || C0 == 5 || C0 == 7) : (C0 >= 1 && C0 <= 4);     // 3-D: up to 4 modalities (BraTS-style inputs)
}

#define FMRI_FIRST_DISPATCH(LAUNCH)                   \
    do {                                              \
        if (!planar && C0 == 1) { LAUNCH(1, false); }  \
        else if (!planar && C0 == 2) { LAUNCH(2, false); } \
        else if (!planar && C0 == 3) { LAUNCH(3, false); } \
        else if (!planar) { LAUNCH(4, false); }       \
        else if (C0 == 1) { LAUNCH(1, true); }        \
        else if (C0 == 3) { LAUNCH(3, true); }        \
        else if (C0 == 5) { LAUNCH(5, true); }        \
        else { LAUNCH(7, true); }                     \
    } while (0)

int conv3d_first_fwd(const void* x, int C0, int planar, const void* w, const float* bias, void* y, int N, int D, int H, int W, int Cout,
                     int act, float alpha, hipStream_t st) {
    if ((((uintptr_t)x) | ((uintptr_t)y)) & 15) return FMRI_E_ALIGN;          // 16-byte loads of the halo rows, 16-byte stores
    const int ntile = N * (D / ff::TD) * (H / ff::TH) * (W / ff::TW);
#define L_(CIN_, PL_)                                                                                                                      \
    do {                                                                                                                                  \
        if (act == FMRI_ACT_RELU)                                                                                                         \
            k_conv_first_fwd<CIN_, PL_, true><<<ntile, ff::NTHREADS, 0, st>>>((const bf16_t*)x, (const bf16_t*)w, bias, (bf16_t*)y, N, D, H, W, Cout, act, alpha);  \
        else                                                                                                                              \
            k_conv_first_fwd<CIN_, PL_, false><<<ntile, ff::NTHREADS, 0, st>>>((const bf16_t*)x, (const bf16_t*)w, bias, (bf16_t*)y, N, D, H, W, Cout, act, alpha); \
    } while (0)
    FMRI_FIRST_DISPATCH(L_);
#undef L_
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

int conv3d_first_wgrad(const void* x, int C0, int planar, const void* dy, float* dw, float* db, int N, int D, int H, int W, int Cout,
                       hipStream_t st) {
    if ((((uintptr_t)x) | ((uintptr_t)dy)) & 15) return FMRI_E_ALIGN;
    const int ntiles = N * (D / fg::TD) * (H / fg::TH) * (W / fg::TW);
    // workgroups in total: every one ends in K x 32 atomics on the same addresses, and needs enough tiles to keep its prefetch busy
    // (measured per launch, 4 x 64x128x128: 256 -> 69 us, 384 -> 70, 512 -> 59, 640 -> 61, 768 -> 60, 1024 -> 61, 2048 -> 75)
    int gx = 512 / (Cout / 32);
    if (gx > ntiles) gx = ntiles;
    if (gx < 1) gx = 1;
#define L_(CIN_, PL_) \
    k_conv_first_wgrad<CIN_, PL_><<<dim3(gx, Cout / 32), fg::NTHREADS, 0, st>>>((const bf16_t*)x, (const bf16_t*)dy, dw, db, N, D, H, W, Cout)
    FMRI_FIRST_DISPATCH(L_);
#undef L_
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
