// bf16 MFMA kernels for the 3x3x3 'same' Conv3D of the U-Net (gfx950 / CDNA4, wave64).
//
//   conv3d_fwd_mfma  : y[v][co] = act(bias[co] + sum_{tap,ci} x[v+tap][ci] * w[tap][co][ci])      (also used as dgrad
//                      with the tap-flipped transposed filter copy and a ReLU mask epilogue)
//   conv3d_wgrad_mfma: dw[tap][co][ci] += sum_v dy[v][co] * x[v+tap][ci]  (+ db[co] += sum_v dy[v][co])
//
// Direct (non-im2col) formulation: a workgroup stages the input HALO tile of one 32-channel chunk in LDS once and re-uses
// it for all 27 taps; the per-tap channel contraction runs on v_mfma_f32_32x32x16_bf16.  Channels-last storage makes the
// MFMA k-dimension (input channels) the contiguous one for forward/dgrad (plain ds_read_b128 fragments); for the weight
// gradient the k-dimension is the voxel index, which is strided in memory, so fragments are gathered with the hardware
// transposing read ds_read_b64_tr_b16.  The lane maps used here were checked on hardware by tools/probe/probe_mfma.hip.
//
// Reference ops replaced: Conv3D / Conv3DBackpropInputV2 / Conv3DBackpropFilterV2 / BiasAdd(Grad) / Relu(Grad) /
// UpSampling3D / ConcatV2 emitted by Keras for fetal_net/model/unet3d/unet.py:45-66,89-115,132-138.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

struct SrcB {
    const bf16_t* p0; const bf16_t* p1; int C0, C1, up0;
};

// ======================================================================================================== forward / dgrad
namespace fw {
constexpr int TD = 4, TH = 8, TW = 16;                 // 512 output voxels per workgroup = 16 MFMA column tiles of 32
constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;   // halo 6 x 10 x 18
constexpr int HVOX = HD * HH * HW;                     // 1080
constexpr int HALO_BYTES = HVOX * 64;                  // one 32-channel chunk, 64 B per voxel
constexpr int NTHREADS = 512;
}  // namespace fw

// 16-byte slot swizzle for 64-byte rows: 4 consecutive rows x 4 slots cover a 256-B bank row exactly once per slot index
__device__ __forceinline__ int swz64(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }

template <int NT>  // NT = 32-wide output-channel tiles per workgroup (BN = 32*NT)
__global__ void __launch_bounds__(fw::NTHREADS)
k_conv_fwd_mfma(SrcB s, const bf16_t* __restrict__ wt, const float* __restrict__ bias, const bf16_t* __restrict__ mask,
                bf16_t* __restrict__ y, int N, int D, int H, int W, int Cout, int act, float alpha) {
    using namespace fw;
    constexpr int BN = 32 * NT;
    constexpr int FILT_BYTES = 9 * BN * 64;
    __shared__ __attribute__((aligned(16))) unsigned char lds[HALO_BYTES + FILT_BYTES];
    unsigned char* const lds_f = lds + HALO_BYTES;

    const int Cin = s.C0 + s.C1;
    const int ncb = Cout / BN;
    const int cb = blockIdx.x % ncb;
    int tile = blockIdx.x / ncb;
    const int twn = W / TW, thn = H / TH, tdn = D / TD;
    const int w0 = (tile % twn) * TW; tile /= twn;
    const int h0 = (tile % thn) * TH; tile /= thn;
    const int d0 = (tile % tdn) * TD;
    const int n = tile / tdn;
    const int co0 = cb * BN;

    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    const int r = lane & 31, hk = lane >> 5;

    f32x16 acc[2][NT];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[j][c][k] = 0.f;

    // per-lane halo index (before adding the tap offset) of this lane's voxel in the wave's two column tiles
    int hv0[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rt = 2 * wv + j;                       // 0..15 : d = rt>>2, h-pair = rt&3
        hv0[j] = ((rt >> 2) * HH + (2 * (rt & 3) + (r >> 4))) * HW + (r & 15);
    }

    const int nch = Cin >> 5;
    for (int ch = 0; ch < nch; ++ch) {
        // ---- which source holds this 32-channel chunk
        const int cc = ch << 5;
        const bool from0 = cc < s.C0;
        const bf16_t* sp = from0 ? s.p0 : s.p1;
        const int sC = from0 ? s.C0 : s.C1;
        const int coff = from0 ? cc : cc - s.C0;
        const int sh = (from0 && s.up0) ? 1 : 0;
        const int sD = D >> sh, sH = H >> sh, sW = W >> sh;
        // ---- stage the halo tile: 1080 voxels x 4 slots of 16 B
        {
            uint4 v[9];
#pragma unroll
            for (int it = 0; it < 9; ++it) {
                const int i = t + it * NTHREADS;
                v[it] = make_uint4(0, 0, 0, 0);
                if (i < HVOX * 4) {
                    const int hv = i >> 2, ps = i & 3;
                    const int ls = ps ^ ((hv >> 2) & 3);
                    const int hw_ = hv % HW, hq = hv / HW;
                    const int hh_ = hq % HH, hd_ = hq / HH;
                    const int gd = d0 - 1 + hd_, gh = h0 - 1 + hh_, gw = w0 - 1 + hw_;
                    if ((unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W) {
                        const int64_t o = ((((int64_t)n * sD + (gd >> sh)) * sH + (gh >> sh)) * sW + (gw >> sh)) * sC + coff + ls * 8;
                        v[it] = *reinterpret_cast<const uint4*>(sp + o);
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < 9; ++it) {
                const int i = t + it * NTHREADS;
                if (i < HVOX * 4) *reinterpret_cast<uint4*>(lds + i * 16) = v[it];
            }
        }
        for (int kd = 0; kd < 3; ++kd) {
            // ---- stage the 9 filter taps of this kd plane: rows = tapl*BN + co, 64 B each
            {
                constexpr int ITEMS = 9 * BN * 4;
                constexpr int ITERS = (ITEMS + NTHREADS - 1) / NTHREADS;
                uint4 v[ITERS];
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    const int i = t + it * NTHREADS;
                    v[it] = make_uint4(0, 0, 0, 0);
                    if (i < ITEMS) {
                        const int row = i >> 2, ps = i & 3;
                        const int ls = ps ^ ((row >> 2) & 3);
                        const int tapl = row / BN, co = row % BN;
                        const int64_t o = ((int64_t)(kd * 9 + tapl) * Cout + co0 + co) * Cin + cc + ls * 8;
                        v[it] = *reinterpret_cast<const uint4*>(wt + o);
                    }
                }
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    const int i = t + it * NTHREADS;
                    if (i < ITEMS) *reinterpret_cast<uint4*>(lds_f + i * 16) = v[it];
                }
            }
            __syncthreads();
#pragma unroll
            for (int tapl = 0; tapl < 9; ++tapl) {
                const int kh = tapl / 3, kw = tapl % 3;
                const int hoff = (kd * HH + kh) * HW + kw;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8_t a[NT], b[2];
#pragma unroll
                    for (int c = 0; c < NT; ++c)
                        a[c] = *reinterpret_cast<const bf16x8_t*>(lds_f + swz64(tapl * BN + c * 32 + r, 2 * ks + hk));
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        b[j] = *reinterpret_cast<const bf16x8_t*>(lds + swz64(hv0[j] + hoff, 2 * ks + hk));
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int c = 0; c < NT; ++c)
                            acc[j][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[c], b[j], acc[j][c], 0, 0, 0);
                }
            }
            __syncthreads();
        }
    }

    // ---- epilogue: D rows = output channel (reg&3)+8*(reg>>2)+4*hk, D cols = voxel r
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rt = 2 * wv + j;
        const int d = d0 + (rt >> 2), h = h0 + 2 * (rt & 3) + (r >> 4), w = w0 + (r & 15);
        const int64_t vo = ((((int64_t)n * D + d) * H + h) * W + w) * Cout + co0;
#pragma unroll
        for (int c = 0; c < NT; ++c) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cch = c * 32 + 8 * g + 4 * hk;
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = acc[j][c][4 * g + i];
                    if (bias) v += bias[co0 + cch + i];
                    if (act == FMRI_ACT_RELU) v = fmaxf(v, 0.f);
                    else if (act == FMRI_ACT_LEAKY) v = v > 0.f ? v : alpha * v;
                    o[i] = v;
                }
                if (mask) {
                    float m[4];
                    ldv<bf16_t, 4>(mask + vo + cch, m);
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (!(m[i] > 0.f)) o[i] = 0.f;
                }
                stv<bf16_t, 4>(y + vo + cch, o);
            }
        }
    }
}

// ======================================================================================================== weight gradient
// One "unit" of work = one d-plane tile of 8 x 16 output voxels (8 k-steps of 16 voxels along w) for one kd.  A workgroup
// (4 waves, 2 workgroups per CU) owns a (kd, 64-wide Cout block, CIB-wide Cin block) slice of dw, keeps its 9 taps x
// 32x32 accumulators in registers (144 VGPRs) and walks a strided list of units.  The x plane (10 x 18 halo rows) and
// the dy plane are brought in by LDS-DMA (global_load_lds_dwordx4: no staging VGPRs, asynchronous) into a 2-deep LDS
// ring: the DMA of unit u+1 is in flight while the MFMAs of unit u run; a counted s_waitcnt + raw s_barrier hand the
// buffers over.  Out-of-volume halo rows are sourced from a zero page, the LDS swizzle is applied on the SOURCE address
// (the DMA destination is lane-linear).
namespace wg {
constexpr int TH = 8, TW = 16;
constexpr int XH = TH + 2, XW = TW + 2;                // 10 x 18 halo rows of one d-plane
constexpr int XROWS = XH * XW;                         // 180
constexpr int YROWS = TH * TW;                         // 128
constexpr int NTHREADS = 256;
}  // namespace wg

__device__ uint4 g_zero_page[8];                       // 128 B of zeros: DMA source for out-of-volume halo rows

// byte offset of 16-B slot `slot` of row `row`; 128-B rows flip their 64-B halves on bit 1 of the row so that the four
// rows touched by one transposing read land in four different 64-B bank quarters.  64-B rows need no swizzle.
template <int ROWB> __device__ __forceinline__ int wg_slot_off(int row, int slot) {
    if (ROWB == 128) return row * 128 + ((slot ^ (((row >> 1) & 1) << 2)) << 4);
    return row * 64 + (slot << 4);
}

// 32x32x16 MFMA operand M[k = 8*hk + j][r] (j = 0..7) from a row-major [k][channels] LDS image, channels tile `tile32`
// (32 channels = 64 B), rows row0 .. row0+15.  Two ds_read_b64_tr_b16, each delivering 4 k-rows x 16 channels per
// 16-lane group (lane 4q+p supplies row q, channels 4p..4p+3; lane i receives channel i).
template <int ROWB>
__device__ __forceinline__ bf16x8_t tr_frag(const unsigned char* base, int row0, int tile32, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3, hk = g >> 1;
    const int rowa = row0 + 8 * hk + q, rowb = rowa + 4;
    const int slot = tile32 * 4 + 2 * (g & 1) + (p >> 1);      // 16-B slot inside the row
    const int sub = (p & 1) * 8;
    const unsigned char* pa = base + wg_slot_off<ROWB>(rowa, slot) + sub;
    const unsigned char* pb = base + wg_slot_off<ROWB>(rowb, slot) + sub;
    s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pa);
    s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pb);
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[j] = v0[j]; f[4 + j] = v1[j]; }
    return __builtin_bit_cast(bf16x8_t, f);
}

// LDS-DMA of 16 B per lane: LDS destination = wave-uniform byte address `lds_dst` + lane*16 (M0-based), global source per
// lane.  Issued from inline asm so that hipcc does not put its own `s_waitcnt vmcnt(0)` in front of the LDS reads of the
// OTHER ring slot (it cannot prove the two slots disjoint); completion is tracked by hand with counted vmcnt waits.
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)p;
}

template <int CI_T>  // CI_T = 32-wide input-channel tiles per workgroup (1 or 2); output-channel block is always 64
__global__ void __launch_bounds__(wg::NTHREADS, 2)
k_conv_wgrad_mfma(SrcB s, const bf16_t* __restrict__ dy, float* __restrict__ dw, float* __restrict__ db, int N, int D, int H,
                  int W, int Cout, int nslab) {
    using namespace wg;
    constexpr int CIB = 32 * CI_T;
    constexpr int XROWB = CIB * 2;                       // bytes per x row
    constexpr int XS = XROWB / 16;                       // 16-B slots per x row
    constexpr int X_INSTR = (XROWS * XS + 63) / 64;      // DMA wave-instructions for the x plane (23 or 12)
    constexpr int Y_INSTR = YROWS * 8 / 64;              // 16
    constexpr int PER_WAVE = (X_INSTR + Y_INSTR + 3) / 4;  // 10 or 7 DMA instructions per wave per unit (padded with dummies)
    constexpr int X_BYTES = X_INSTR * 1024;
    constexpr int Y_BYTES = Y_INSTR * 1024;
    constexpr int DUMMY_BYTES = (PER_WAVE * 4 - X_INSTR - Y_INSTR) * 1024;
    constexpr int STAGE_BYTES = X_BYTES + Y_BYTES + DUMMY_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE_BYTES];

    const int Cin = s.C0 + s.C1;
    const int ncib = Cin / CIB, ncob = Cout / 64;
    int combo = blockIdx.x / nslab;
    const int slab = blockIdx.x % nslab;
    const int cib = combo % ncib; combo /= ncib;
    const int cob = combo % ncob;
    const int kd = combo / ncob;
    const int co0 = cob * 64, cc = cib * CIB;

    const bool from0 = cc < s.C0;
    const bf16_t* sp = from0 ? s.p0 : s.p1;
    const int sC = from0 ? s.C0 : s.C1;
    const int coff = from0 ? cc : cc - s.C0;
    const int sh = (from0 && s.up0) ? 1 : 0;
    const int sD = D >> sh, sH = H >> sh, sW = W >> sh;

    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 31, hk = lane >> 5;
    const int ct = wv & 1;
    const int it = (CI_T == 2) ? (wv >> 1) : 0;
    const int ksl = (CI_T == 2) ? 0 : (wv >> 1);
    constexpr int KSTEP = (CI_T == 2) ? 1 : 2;
    const bool do_bias = (db != nullptr) && kd == 0 && cib == 0 && it == 0;

    f32x16 acc[9];
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[a][k] = 0.f;
    float bsum = 0.f;

    const int twn = W / TW, thn = H / TH;
    const int nunits = N * D * thn * twn;

    // issue the DMA of unit `u` into ring slot `buf`: this wave's PER_WAVE instructions (instr = wv + 4*j), branch-free
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(lds));
    auto issue = [&](int u, int buf) {
        int q = u;
        const int w0 = (q % twn) * TW; q /= twn;
        const int h0 = (q % thn) * TH; q /= thn;
        const int d = q % D;
        const int n = q / D;
        const int gd = d + kd - 1;
        const bool dok = (unsigned)gd < (unsigned)D;
        const int gdc = min(max(gd, 0), D - 1) >> sh;
        const int64_t xplane = ((int64_t)n * sD + gdc) * sH;
        const int64_t yplane = (((int64_t)n * D + d) * H + h0) * W + w0;
#pragma unroll
        for (int j = 0; j < PER_WAVE; ++j) {
            const int instr = wv + 4 * j;                       // wave-uniform
            const void* src = (const void*)g_zero_page;
            if (instr < X_INSTR) {
                const int i = instr * 64 + lane;
                const int row = i / XS, ps = i % XS;
                const int ls = (XROWB == 128) ? (ps ^ (((row >> 1) & 1) << 2)) : ps;
                const int xh = row / XW, xw = row % XW;
                const int gh = h0 - 1 + xh, gw = w0 - 1 + xw;
                const bool ok = dok && row < XROWS && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
                const int ghc = min(max(gh, 0), H - 1) >> sh, gwc = min(max(gw, 0), W - 1) >> sh;
                const bf16_t* real = sp + (((xplane + ghc) * sW + gwc) * sC + coff + ls * 8);
                src = ok ? (const void*)real : (const void*)g_zero_page;
            } else if (instr < X_INSTR + Y_INSTR) {
                const int i = (instr - X_INSTR) * 64 + lane;
                const int row = i >> 3, ps = i & 7;
                const int ls = ps ^ (((row >> 1) & 1) << 2);
                const int yh = row >> 4, yw = row & 15;
                src = dy + ((yplane + (int64_t)yh * W + yw) * Cout + co0 + ls * 8);
            }
            dma16(src, __builtin_amdgcn_readfirstlane(lds0 + buf * STAGE_BYTES + instr * 1024));
        }
    };

    int u = slab;
    int buf = 0;
    if (u < nunits) issue(u, 0);
    for (; u < nunits; u += nslab, buf ^= 1) {
        const bool more = (u + nslab) < nunits;
        if (more) {
            issue(u + nslab, buf ^ 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE) : "memory");   // this unit's DMA (older) has landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        const unsigned char* const xb = lds + buf * STAGE_BYTES;
        const unsigned char* const yb = xb + X_BYTES;
#pragma unroll 1
        for (int ks = ksl; ks < 8; ks += KSTEP) {
            const bf16x8_t a = tr_frag<128>(yb, ks * TW, ct, lane);                      // A[co][k = voxel]
            if (do_bias) {
                typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
                const u16x8 au = __builtin_bit_cast(u16x8, a);
#pragma unroll
                for (int j = 0; j < 8; ++j) bsum += bf2f(au[j]);
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kh = tap / 3, kw = tap % 3;
                const bf16x8_t b = tr_frag<XROWB>(xb, (ks + kh) * XW + kw, it, lane);    // B[k = voxel][ci]
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[tap], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                                 // ring slot `buf` may be refilled
    }
    // ---- flush: D rows = co, cols = ci; one fp32 atomic per element (128-B contiguous per half-wave)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int co = co0 + ct * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hk;
            const int ci = cc + it * 32 + r;
            atomicAdd(&dw[((int64_t)(kd * 9 + tap) * Cout + co) * Cin + ci], acc[tap][reg]);
        }
    }
    if (do_bias) {
        bsum += __shfl_down(bsum, 32);
        if (hk == 0) atomicAdd(&db[co0 + ct * 32 + r], bsum);
    }
}

}  // namespace

// --------------------------------------------------------------------------------------------------------- host dispatch
bool conv3d_fwd_mfma_ok(int C0, int C1, int Cout, int D, int H, int W, int dtype) {
    if (dtype != FMRI_BF16) return false;
    if ((C0 % 32) || (C1 % 32) || C0 + C1 < 32 || (Cout % 32)) return false;
    if ((D % fw::TD) || (H % fw::TH) || (W % fw::TW)) return false;
    return true;
}
bool conv3d_wgrad_mfma_ok(int C0, int C1, int Cout, int D, int H, int W, int dtype) {
    if (dtype != FMRI_BF16) return false;
    if ((C0 % 32) || (C1 % 32) || C0 + C1 < 32 || (Cout % 64)) return false;
    if ((H % wg::TH) || (W % wg::TW)) return false;
    return true;
}

int conv3d_fwd_mfma(const void* src0, int C0, int up0, const void* src1, int C1, const void* w, const float* bias,
                    const void* mask, void* y, int N, int D, int H, int W, int Cout, int act, float alpha, hipStream_t st) {
    SrcB s{(const bf16_t*)src0, (const bf16_t*)src1, C0, C1, up0};
    const int ntile = N * (D / fw::TD) * (H / fw::TH) * (W / fw::TW);
    if (Cout % 64 == 0) {
        k_conv_fwd_mfma<2><<<ntile * (Cout / 64), fw::NTHREADS, 0, st>>>(s, (const bf16_t*)w, bias, (const bf16_t*)mask, (bf16_t*)y,
                                                                         N, D, H, W, Cout, act, alpha);
    } else {
        k_conv_fwd_mfma<1><<<ntile * (Cout / 32), fw::NTHREADS, 0, st>>>(s, (const bf16_t*)w, bias, (const bf16_t*)mask, (bf16_t*)y,
                                                                         N, D, H, W, Cout, act, alpha);
    }
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

int conv3d_wgrad_mfma(const void* src0, int C0, int up0, const void* src1, int C1, const void* dy, float* dw, float* db, int N,
                      int D, int H, int W, int Cout, hipStream_t st) {
    SrcB s{(const bf16_t*)src0, (const bf16_t*)src1, C0, C1, up0};
    const int Cin = C0 + C1;
    const bool wide = (C0 % 64 == 0) && (C1 % 64 == 0);
    const int CIB = wide ? 64 : 32;
    const int combos = 3 * (Cout / 64) * (Cin / CIB);
    const int ntiles = N * D * (H / wg::TH) * (W / wg::TW);
    int nslab = (1024 + combos - 1) / combos;        // aim for ~1024 workgroups (2 per CU, 2 rounds)
    if (nslab > ntiles) nslab = ntiles;
    if (nslab < 1) nslab = 1;
    if (wide) k_conv_wgrad_mfma<2><<<combos * nslab, wg::NTHREADS, 0, st>>>(s, (const bf16_t*)dy, dw, db, N, D, H, W, Cout, nslab);
    else k_conv_wgrad_mfma<1><<<combos * nslab, wg::NTHREADS, 0, st>>>(s, (const bf16_t*)dy, dw, db, N, D, H, W, Cout, nslab);
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
