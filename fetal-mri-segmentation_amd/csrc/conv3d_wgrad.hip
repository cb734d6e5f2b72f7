// bf16 MFMA weight-gradient kernels of the 3x3x3 'same' Conv3D (gfx950 / CDNA4, wave64); split from conv3d_mfma.hip in round 6 (the
// forward / input-gradient kernels stay there, shared device helpers in mfma_common.h).
//   dw[tap][co][ci] += sum_v dy[v][co] * x[v+tap][ci]  (+ db[co] += sum_v dy[v][co])
// Reference ops replaced: Conv3DBackpropFilterV2 / BiasAddGrad emitted by Keras for fetal_net/model/unet3d/unet.py:45-66,89-115,132-138.
#define FMRI_MFMA_TU_WGRAD
#include "mfma_common.h"

FMRI_DET_TU(mfma)

namespace {

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

// UPW ("up weight-gradient"): gradient of the pre-summed parity filters of the up-sampled channels (see k_conv_fwd_mfma MODE 1).
// dWc[p][t'] = sum over low-res voxels g of dy[2g+p] (x) x_low[g + t' + p - 1]: the kernel runs on the LOW-res grid (D,H,W = low-res
// dims), the Cout blocks enumerate (parity, 64-channel block), only kd in {pd, pd+1} gets workgroups and only the 4 (kh,kw) taps
// {ph,ph+1} x {pw,pw+1} are accumulated; the dy tile is gathered from the parity-p voxels of the [2D][2H][2W] gradient.
// Result layout [8 p][2][2][2][Cout][C0] fp32 (atomics), expanded into the 27-tap gradient by k_expand_up_wgrad.
// WS: warp-specialised variant - eight waves, ONE workgroup per CU: waves 0-3 (one per SIMD) read fragments and issue MFMAs exactly as the
// four waves of the plain kernel do, waves 4-7 issue all LDS-DMA (tools/prof_wgrad.py: a wave of the plain kernel spends 35-50 % of its time
// issuing its ten DMA instructions per unit, wherever they are placed).
template <int CI_T, bool UPW, bool WS = false>  // CI_T = 32-wide input-channel tiles per workgroup (1 or 2); output-channel block is always 64
__global__ void __launch_bounds__(WS ? 512 : wg::NTHREADS, WS ? 1 : 2)
k_conv_wgrad_mfma(SrcB s, const bf16_t* __restrict__ dy, float* __restrict__ dw, float* __restrict__ db, int N, int D, int H,
                  int W, int Cout, int nslab, float* __restrict__ slab_ws, int dw_ld) {
    using namespace wg;
    constexpr int CIB = 32 * CI_T;
    constexpr int XROWB = CIB * 2;                       // bytes per x row
    constexpr int XS = XROWB / 16;                       // 16-B slots per x row
    constexpr int X_INSTR = (XROWS * XS + 63) / 64;      // DMA wave-instructions for the x plane (23 or 12)
    constexpr int Y_INSTR = YROWS * 8 / 64;              // 16
    constexpr int XPW = (X_INSTR + 3) / 4;               // x instructions per wave (6 or 3; short waves re-issue their first)
    constexpr int YPW = Y_INSTR / 4;                     // 4
    constexpr int PER_WAVE = XPW + YPW;                  // 10 or 7 DMA instructions per wave per unit
    constexpr int X_BYTES = X_INSTR * 1024;
    constexpr int Y_BYTES = Y_INSTR * 1024;
    constexpr int STAGE_BYTES = X_BYTES + Y_BYTES;
    constexpr int NSTAGE = WS ? 3 : 2;                   // WS: the producers run TWO units ahead (one workgroup per CU leaves the LDS for it)
    __shared__ __attribute__((aligned(16))) unsigned char lds[NSTAGE * STAGE_BYTES];
    const unsigned char* const zpage = zero_page_addr();

    const int Cin = s.C0 + s.C1;
    const int ncib = Cin / CIB, ncob = (UPW ? (s.planar ? 4 : 8) : 1) * (Cout / 64);
    // slab-major block order: the (kd, Cout block, Cin block) workgroups that read the SAME planes are adjacent in launch
    // order, so they run at the same time and share those planes in L2 / Infinity Cache instead of re-reading HBM
    // (measured with rocprofv3 FETCH_SIZE: combo-major order fetched 3.6x the algorithmic bytes)
    const int ncombo = (s.planar ? 1 : (UPW ? 2 : 3)) * ncob * ncib;
    const int wg_id = xcd_logical_id(blockIdx.x, gridDim.x);      // workgroups of one slab (same planes) on one XCD / L2
    int combo = wg_id % ncombo;
    const int slab = wg_id / ncombo;
    const int cib = combo % ncib; combo /= ncib;
    const int cob = combo % ncob;
    const int par = UPW ? cob / (Cout / 64) : 0;           // output parity class (pd, ph, pw) of this workgroup; planar: (ph, pw)
    const int kdp = combo / ncob;                          // UPW: kd' in {0,1}
    const int kd = s.planar ? 1 : (UPW ? kdp + (par >> 2) : kdp);   // planar (2-D slices): only the centre kd plane exists
    const int co0 = (UPW ? cob % (Cout / 64) : cob) * 64, cc = cib * CIB;

    const bool from0 = cc < s.C0;
    const bf16_t* sp = from0 ? s.p0 : s.p1;
    const int sC = from0 ? s.C0 : s.C1;
    const int coff = from0 ? cc : cc - s.C0;
    const int sh = (from0 && s.up0) ? 1 : 0;
    const int shd = sh & s.dsh;
    const int sD = D >> shd, sH = H >> sh, sW = W >> sh;

    const int t = threadIdx.x, lane = t & 63;
    const int wv_all = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool producer = WS && wv_all >= 4;
    const int wv = wv_all & 3;                             // index among the four waves of this wave's role
    const int r = lane & 31, hk = lane >> 5;
    const int ct = wv & 1;
    const int it = (CI_T == 2) ? (wv >> 1) : 0;
    const int ksl = (CI_T == 2) ? 0 : (wv >> 1);
    // bias gradient = sum of dy over all voxels: taken from the A fragments by the workgroups that see every dy tile exactly once
    // (UPW: each parity class covers its own eighth of the voxels)
    const bool do_bias = (db != nullptr) && cib == 0 && it == 0 && (UPW ? kdp == 0 : kd == (s.planar ? 1 : 0));

    constexpr int NACC = UPW ? 4 : 9;
    f32x16 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[a][k] = 0.f;
    float bsum = 0.f;

    const int twn = W / TW, thn = H / TH;
    const int nunits = N * D * thn * twn;

    // ---- per-lane constants of the DMA address generation
    int x_pack[XPW];               // xh | xw<<4 | ls<<9 | valid<<12
    unsigned x_doff[XPW];
#pragma unroll
    for (int j = 0; j < XPW; ++j) {
        int instr = wv + 4 * j;
        if (instr >= X_INSTR) instr = wv;
        const int i = instr * 64 + lane;
        const int row = i / XS, ps = i % XS;
        const int ls = (XROWB == 128) ? (ps ^ (((row >> 1) & 1) << 2)) : ps;
        const int rowc = row < XROWS ? row : 0;
        x_pack[j] = (rowc / XW) | ((rowc % XW) << 4) | (ls << 9) | ((row < XROWS ? 1 : 0) << 12);
        x_doff[j] = instr * 1024;
    }
    int y_soff[YPW];
    unsigned y_doff[YPW];
#pragma unroll
    for (int j = 0; j < YPW; ++j) {
        const int instr = wv + 4 * j;
        const int i = instr * 64 + lane;
        const int row = i >> 3, ps = i & 7;
        const int ls = ps ^ (((row >> 1) & 1) << 2);
        y_soff[j] = UPW ? ((row >> 4) * 4 * W + (row & 15) * 2) * Cout + ls * 8 : ((row >> 4) * W + (row & 15)) * Cout + ls * 8;
        y_doff[j] = X_BYTES + instr * 1024;
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(lds));
    // ---- DMA address generation.  A workgroup issues the units of its run strictly in order, so the issue side keeps a CURSOR (n, h0, w0, d)
    // that is advanced by one unit per call - no divisions in the loop - and everything that only depends on the column (n, h0, w0): the
    // per-lane element offset of each x piece inside a plane and whether that halo row lies inside the volume, is worked out once per
    // column.  Per unit what is left is two scalar 64-bit bases and, per DMA instruction, one 64-bit add and a select.  (Round 1 redid the
    // whole decode - six divisions by run-time values, ~25 VALU instructions per piece - at every unit: tools/prof_wgrad.py showed a wave
    // spending 40-50 % of its time between the barrier and its first MFMA.)
    // unit index = column (n, h-tile, w-tile) x D + d: d runs fastest, and a workgroup walks a CONTIGUOUS run of units, i.e. up a column.
    // The three kd workgroups of a slab then read x planes d-1, d, d+1 at step d and d, d+1, d+2 at the next: two of the three planes (and
    // the dy plane all three share) were fetched a step ago by a neighbour on the same XCD, so they come out of L2 instead of HBM.
    int ic_n = 0, ic_h0 = 0, ic_w0 = 0, ic_d = 0;
    int x_off[XPW];                // element offset of this lane's 16 bytes inside an x plane of the cursor's column
    unsigned x_ok = 0;             // bit j: piece j of this lane lies inside the volume in h and w
    auto col_setup = [&]() {
        x_ok = 0;
#pragma unroll
        for (int j = 0; j < XPW; ++j) {
            const int pk = x_pack[j];
            const int gh = ic_h0 - 1 + (pk & 15), gw = ic_w0 - 1 + ((pk >> 4) & 31);
            const bool ok = (pk >> 12) && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
            const int ghc = min(max(gh, 0), H - 1) >> sh, gwc = min(max(gw, 0), W - 1) >> sh;
            x_off[j] = (ghc * sW + gwc) * sC + ((pk >> 9) & 7) * 8;
            x_ok |= (ok ? 1u : 0u) << j;
        }
    };
    auto cursor_set = [&](int u) {
        int q = u;
        ic_d = q % D; q /= D;
        ic_w0 = (q % twn) * TW; q /= twn;
        ic_h0 = (q % thn) * TH;
        ic_n = q / thn;
        col_setup();
    };
    auto issue = [&](int buf) {
        const int n = ic_n, h0 = ic_h0, w0 = ic_w0, d = ic_d;
        const int gd = d + kd - 1;
        const bool dok = (unsigned)gd < (unsigned)D;
        const int gdc = min(max(gd, 0), D - 1) >> shd;
        const bf16_t* const xbase = sp + ((int64_t)n * sD + gdc) * sH * sW * sC + coff;
        const bf16_t* const ybase =
            UPW ? (s.planar ? dy + ((((int64_t)n * D + d) * 2 * H + 2 * h0 + ((par >> 1) & 1)) * 2 * W + 2 * w0 + (par & 1)) * Cout + co0
                            : dy + ((((int64_t)n * 2 * D + 2 * d + (par >> 2)) * 2 * H + 2 * h0 + ((par >> 1) & 1)) * 2 * W + 2 * w0 + (par & 1)) * Cout + co0)
                : dy + ((((int64_t)n * D + d) * H + h0) * W + w0) * Cout + co0;
        const unsigned sbase = lds0 + buf * STAGE_BYTES;
        // (round 5: through buffer descriptors, see k_conv_wgrad_kd's issue_x - out-of-volume rows are out-of-range offsets, an
        // out-of-volume plane a descriptor of zero records)
        const i32x4 xrs = dma_rsrc(xbase, dok ? (int)DMA_OOB : 0), yrs = dma_rsrc(ybase);
#pragma unroll
        for (int j = 0; j < XPW; ++j)
            dma16_buf(xrs, ((x_ok >> j) & 1) ? (unsigned)x_off[j] * 2u : DMA_OOB, __builtin_amdgcn_readfirstlane(sbase + x_doff[j]));
#pragma unroll
        for (int j = 0; j < YPW; ++j) dma16_buf(yrs, (unsigned)y_soff[j] * 2u, __builtin_amdgcn_readfirstlane(sbase + y_doff[j]));
        // advance the cursor: up the column, then on to the next column of the run
        if (++ic_d == D) {
            ic_d = 0;
            ic_w0 += TW;
            if (ic_w0 == W) {
                ic_w0 = 0;
                ic_h0 += TH;
                if (ic_h0 == H) { ic_h0 = 0; ++ic_n; }
            }
            col_setup();
        }
    };

    // ---- per-lane constants of the transposing fragment reads.  A row index is (lane part) + (wave-uniform constant c);
    // adding a multiple of 4 rows never changes the swizzle, so addr(c) = pre[c & 3] + (c >> 2) * 4 * ROWB.
    const int gq = lane >> 4, qd = (lane & 15) >> 2, pp = lane & 3;
    const int lrow = 8 * (gq >> 1) + qd;                              // 0..11
    const int lslot_x = it * 4 + 2 * (gq & 1) + (pp >> 1), lslot_y = ct * 4 + 2 * (gq & 1) + (pp >> 1);
    int pre_x[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) pre_x[m] = wg_slot_off<XROWB>(lrow + m, lslot_x) + (pp & 1) * 8;
    const int pre_y = X_BYTES + wg_slot_off<128>(lrow, lslot_y) + (pp & 1) * 8;   // y row0 = ks*16 is a multiple of 4

    // the unit loop, instantiated per (ph, pw) in the up mode so that every fragment row constant stays compile-time
    auto run = [&](auto PHc, auto PWc) {
    constexpr int PH_ = decltype(PHc)::value, PW_ = decltype(PWc)::value;
    const int per = (nunits + nslab - 1) / nslab;          // contiguous run of units per slab
    int u = slab * per;
    const int u_end = min(nunits, u + per);
    int buf = 0;
#ifdef FMRI_PROF
    unsigned long long wprof[12] = {};
    PROF_T(wk0);
#endif
    if (u < u_end && (!WS || producer)) {
        cursor_set(u);
        issue(0);
    }
    if (WS && producer && u + 1 < u_end) issue(1);
    for (; u < u_end; ++u, buf = (buf + 1 == NSTAGE ? 0 : buf + 1)) {
        const bool more = (u + 1) < u_end;
        PROF_T(w0);
        // ONE barrier per unit: "my DMA for this unit has landed" + "everybody is done reading the ring slot of the previous unit" (a wave
        // gets here only after its MFMAs on it) - then that slot is refilled: with the unit after this one (plain kernel, 2 slots) or the
        // one after that (WS, 3 slots: the wait below leaves the youngest unit's DMA in flight)
        if (WS) {
            if (producer) {
                if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        PROF_T(w1);
        __builtin_amdgcn_s_barrier();
        PROF_T(w2);
        if (WS) {
            if (producer && u + 2 < u_end) issue(buf == 0 ? 2 : buf - 1);                 // = (buf + 2) % 3: the slot of the previous unit
        } else if (more) issue(buf ^ 1);
        PROF_T(w3);
        if (producer) continue;
        const unsigned char* const sb = lds + buf * STAGE_BYTES;
        if constexpr (WS && CI_T == 2) {
            // One consumer wave per SIMD: nobody else hides the latency of the transposing reads, so the fragments run two MFMAs ahead of
            // their use (3-deep x-fragment ring, double dy fragment), threaded between the MFMAs.
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            constexpr int NS = 8 * NACC;
            auto row_of = [&](int st) {
                const int ks8 = st / NACC, tap = st % NACC;
                return UPW ? (ks8 + (tap >> 1) + PH_) * XW + (tap & 1) + PW_ : (ks8 + tap / 3) * XW + (tap % 3);
            };
            auto load_bf = [&](int st) {
                const int c = row_of(st);
                const unsigned char* pb = sb + pre_x[c & 3] + (c >> 2) * 4 * XROWB;
                s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pb);
                s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pb + 4 * XROWB));
                return __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
            };
            auto load_af = [&](int ks8) {
                const unsigned char* pa = sb + pre_y + ks8 * TW * 128;
                s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pa);
                s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pa + 4 * 128));
                return __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
            };
            s16x8 af2[2], bf3[3];
            af2[0] = load_af(0);
            bf3[0] = load_bf(0);
            bf3[1] = load_bf(1);
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                const int ks8 = st / NACC, tap = st % NACC;
                if (st + 2 < NS) bf3[(st + 2) % 3] = load_bf(st + 2);
                if (tap == NACC - 3 && ks8 + 1 < 8) af2[(ks8 + 1) & 1] = load_af(ks8 + 1);
                if (do_bias && tap == 0) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) bsum += bf2f((unsigned short)af2[ks8 & 1][j]);
                }
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af2[ks8 & 1]), __builtin_bit_cast(bf16x8_t, bf3[st % 3]),
                                                                  acc[tap], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else
#pragma unroll
        for (int ks8 = 0; ks8 < 8; ++ks8) {
            if (CI_T == 1 && (ks8 & 1) != ksl) continue;              // Cin-block 32: the two wave pairs split the k-steps
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            s16x8 af;
            {
                const unsigned char* pa = sb + pre_y + ks8 * TW * 128;
                s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pa);
                s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pa + 4 * 128));
                af = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);      // plain concatenation: no element-wise copies
            }
            const bf16x8_t a = __builtin_bit_cast(bf16x8_t, af);                     // A[co][k = voxel]
            if (do_bias) {
#pragma unroll
                for (int j = 0; j < 8; ++j) bsum += bf2f((unsigned short)af[j]);
            }
#pragma unroll
            for (int tap = 0; tap < NACC; ++tap) {
                const int c = UPW ? (ks8 + (tap >> 1) + PH_) * XW + (tap & 1) + PW_ : (ks8 + tap / 3) * XW + (tap % 3);   // compile-time row constant
                const unsigned char* pb = sb + pre_x[c & 3] + (c >> 2) * 4 * XROWB;
                s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pb);
                s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pb + 4 * XROWB));
                const s16x8 bfv = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8_t, bfv), acc[tap], 0, 0, 0);   // B[k = voxel][ci]
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                            // fragments in registers before this wave reports "done reading"
#ifdef FMRI_PROF
        { PROF_T(w4); wprof[0] += w1 - w0; wprof[1] += w2 - w1; wprof[2] += w3 - w2; wprof[3] += w4 - w3; wprof[6] += 1; }
#endif
    }
#ifdef FMRI_PROF
    { PROF_T(wk1); wprof[5] = wk1 - wk0; }
    if (lane == 0) for (int i = 0; i < 7; ++i) atomicAdd(&g_prof[i], wprof[i]);
#endif
    };
    if constexpr (UPW) {
        switch (par & 3) {
            case 0: run(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}); break;
            case 1: run(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}); break;
            case 2: run(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}); break;
            default: run(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}); break;
        }
    } else {
        run(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    }
    if (producer) return;
    const FmriDetCfg dc = g_det_cfg;                    // deterministic mode: fixed-point shadow of the gradient buffer (common.h)
    // ---- flush: D rows = co, cols = ci (128-B contiguous per half-wave).  With a workspace: plain stores of this workgroup's partial
    // slab [9][64][CIB] (summed per element by k_wgrad_reduce: deterministic, ~5x the atomic rate); without: fp32 atomics into dw.
    if constexpr (UPW) {
        // dWc[p][kd'][kh'][kw'][Cout][C0]
#pragma unroll
        for (int tap = 0; tap < 4; ++tap) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int co = co0 + ct * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hk;
                const int ci = cc + it * 32 + r;
                atomicAdd(&dw[((int64_t)((s.planar ? par : par * 2 + kdp) * 4 + tap) * Cout + co) * Cin + ci], acc[tap][reg]);
            }
        }
    } else if (slab_ws) {
        constexpr int KSP = (CI_T == 2) ? 1 : 2;          // Cin-block 32: the two k-step halves keep separate slabs
        float* const my = slab_ws + ((int64_t)wg_id * KSP + ksl) * (9 * 64 * CIB);
#pragma unroll
        for (int tap = 0; tap < NACC; ++tap) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int col = ct * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hk;
                my[(tap * 64 + col) * CIB + it * 32 + r] = acc[tap][reg];
            }
        }
    } else {
#pragma unroll
        for (int tap = 0; tap < NACC; ++tap) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int co = co0 + ct * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hk;
                const int ci = cc + it * 32 + r;
                fmri_grad_add(dc, &dw[((int64_t)(kd * 9 + tap) * Cout + co) * dw_ld + ci], acc[tap][reg]);
            }
        }
    }
    if (do_bias) {
        bsum += __shfl_down(bsum, 32);
        if (hk == 0) fmri_grad_add(dc, &db[co0 + ct * 32 + r], bsum);
    }
}

// ============================================================================================ weight gradient, kd-sharing form
// The kernel above gives every (kd, Cout block, Cin block) its own workgroup, so each x plane and each dy plane is staged by three
// workgroups: 135 B of LDS-DMA per MFMA, and tools/prof_wgrad.py shows its waves waiting for planes they requested a whole unit earlier
// (~20 MB of DMA requests in flight chip-wide): the weight gradient is bound by its LDS-DMA stream.  Here a workgroup owns a (Cout, Cin)
// block for ALL 27 taps and walks up a column of d-plane tiles with a 4-slot ring of x planes (d-1, d, d+1 in use, d+2 in flight) and a
// 2-slot ring of dy planes: one new x plane + one dy plane per step feed all three kd.  wave = tap group of 7 | 7 | 7 | 6 taps.
//   BLK = 32: (32 Cout, 32 Cin) block, 4 waves, TWO workgroups per CU (64 KiB of LDS each) - two independent barrier domains, as in the
//             kernel above; 112 accumulator registers per wave leave room for a 4-deep fragment ring.  19.5 KB of DMA per 216 MFMAs =
//             90 B per MFMA (-33 %).
//   BLK = 64 (round 6, "one wave per SIMD"): (64 Cout, 32 Cin) block, 4 waves, ONE workgroup per CU.  A wave holds 7 taps x 2 Cout halves x
//             32 x 32 = 224 accumulator registers and has its SIMD's whole 512-entry register file: an x fragment serves two MFMAs (0.64 KB of
//             LDS reads per MFMA instead of 1.14), 27.5 KB of DMA per 432 MFMAs = 64 B per MFMA, the fragment ring runs PF reads ahead in
//             registers (nobody else hides the read latency), and the unit's DMA pieces are issued one at a time BETWEEN MFMAs (SPREAD) instead
//             of as a burst behind the barrier.  (Rounds 2-5 had a (64, 64) block on 8 waves here - two waves per SIMD at 224 accumulators
//             each, i.e. inside 256 registers: spills, one barrier domain for eight waves, 8-25 % slower.  Removed.)
// Flush: fp32 atomics (128 contiguous bytes per half-wave).  Bit-identical sums on exactly representable data (tests).
// F32 (round 6, the fp32 parity mode on this kernel's structure): fp32 planes, a (32 Cout, 32 Cin) block on v_mfma_f32_32x32x2_f32 - rows of
// 128 bytes for both operands, the plane ring / DMA pieces / cursor / flush of the bf16 kernel; a fragment is ONE voxel per half-wave
// (lane = channel), so it is a plain ds_read_b32 per operand and MFMA, no transposing read.  One workgroup per CU (124 KiB of LDS).
template <int BLK, bool F32 = false>
struct WkCfg {
    static constexpr int NW = 4;                                  // waves per workgroup = tap groups (all of them issue DMA)
    static constexpr int NH = F32 ? 1 : BLK / 32;                 // Cout halves (32 channels each) per wave
    static constexpr int XROWB = F32 ? 128 : 64, YROWB = F32 ? 128 : BLK * 2;      // bytes per x row (32 input channels) / per dy row in LDS
    static constexpr int XRS = XROWB / 16, YRS = YROWB / 16;      // 16-byte slots per row
    static constexpr int X_INSTR = (wg::XROWS * XRS + 63) / 64;   // DMA wave-instructions per x plane: 12
    static constexpr int Y_INSTR = wg::YROWS * YRS / 64;          // ... per dy plane: 8 / 16
    static constexpr int XS_BYTES = X_INSTR * 1024, YS_BYTES = Y_INSTR * 1024;
    static constexpr int NXS = 4, NYS = 2;
    static constexpr int LDS_BYTES = NXS * XS_BYTES + NYS * YS_BYTES;      // 65,536 / 81,920
    static constexpr int PF = BLK == 64 ? 4 : 3;                  // x fragments in flight ahead of their MFMAs
    static constexpr int XPW = (X_INSTR + NW - 1) / NW, YPW = (Y_INSTR + NW - 1) / NW;      // 3, 2 / 4
    // DMA pieces of the next unit issued between the MFMAs of this one (one wave per SIMD: nobody else issues under this wave's matrix work).
    // The 32-block kernel keeps its burst behind the barrier: the spread form is level there (profiles/r06_kd32_spread_ab.log)
    static constexpr bool SPREAD = BLK == 64 && !F32;
    static constexpr int SP0 = 1, SPD = 3;                        // ... piece p behind step SP0 + SPD * p (a step = NH MFMAs)
};

template <int BLK, int G, class Hook>   // tap group: taps 7G .. 7G + NTAP - 1 of the 27; hook(step) runs behind every step's MFMAs
__device__ __forceinline__ void wk_compute(const unsigned char* lds, const int (&xb)[3], int yb, const int (&pre_x)[4], int pre_y,
                                           f32x16 (&acc)[G == 3 ? 6 : 7][WkCfg<BLK>::NH], float (&bsum)[2], bool do_bias, Hook&& hook) {
    typedef WkCfg<BLK> K;
    constexpr int NTAP = G == 3 ? 6 : 7, T0 = 7 * G, NH = K::NH, XROWB = K::XROWB, YROWB = K::YROWB;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    auto load_a = [&](int ks8, s16x8 (&av)[NH]) {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            // the second Cout half sits 4 slots further: bit 6 of the in-row offset
            const unsigned char* p0 = lds + yb + (h ? (pre_y ^ 64) : pre_y) + ks8 * wg::TW * YROWB;
            s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
            s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 4 * YROWB));
            av[h] = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    };
    // x fragment of step st = ks8 * NTAP + j (tap T0 + j): base of the tap's kd slot + this lane's offset for the swizzle class of the
    // (compile-time) halo row of the fragment's first voxel
    auto load_b = [&](int st) {
        const int ks8 = st / NTAP, t = T0 + st % NTAP, kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
        const int c = (ks8 + kh) * wg::XW + kw;
        const unsigned char* pb = lds + xb[kd] + pre_x[c & 3] + (c >> 2) * 4 * XROWB;
        s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pb);
        s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pb + 4 * XROWB));
        return __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    // Fixed register budget: the dy fragments of a k-step are single-buffered, the x fragments run PF steps ahead of their MFMAs in a
    // register ring; sched_barrier keeps the compiler from hoisting more reads (left alone it spills thousands of registers at BLK = 64).
    constexpr int NS = 8 * NTAP, PF = K::PF, RING = PF + 1;
    // DBA (one wave per SIMD): the dy fragments of the next k-step are requested three steps before it starts (double-buffered) - with
    // two waves per SIMD the partner covers that latency and the registers are better spent elsewhere
    constexpr bool DBA = BLK == 64;
    s16x8 av[DBA ? 2 : 1][NH], b[RING];
#pragma unroll
    for (int q = 0; q < PF; ++q) b[q] = load_b(q);
    if constexpr (DBA) load_a(0, av[0]);
#pragma unroll
    for (int ks8 = 0; ks8 < 8; ++ks8) {
        const int ab = DBA ? (ks8 & 1) : 0;
        if constexpr (!DBA) load_a(ks8, av[0]);
        if constexpr (G == 3) {
            if (do_bias) {
                // bias gradient = sum of the dy fragments (the 6-tap group has registers to spare)
#pragma unroll
                for (int h = 0; h < NH; ++h)
#pragma unroll
                    for (int q = 0; q < 8; ++q) bsum[h] += bf2f((unsigned short)av[ab][h][q]);
            }
        }
#pragma unroll
        for (int j = 0; j < NTAP; ++j) {
            const int st = ks8 * NTAP + j;
            if (st + PF < NS) b[(st + PF) % RING] = load_b(st + PF);
            if constexpr (DBA) {
                if (j == NTAP - 3 && ks8 + 1 < 8) load_a(ks8 + 1, av[ab ^ 1]);
            }
            const bf16x8_t bb = __builtin_bit_cast(bf16x8_t, b[st % RING]);
#pragma unroll
            for (int h = 0; h < NH; ++h)
                acc[j][h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av[ab][h]), bb, acc[j][h], 0, 0, 0);
            hook(ks8 * NTAP + j);            // (the loops are fully unrolled: the step is a constant in every copy)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// S16 (round 5; BLK = 32): the same (32 Cout, 32 Cin) x 7-tap block per wave on v_mfma_f32_16x16x32_bf16 - four 16 x 16 accumulators per tap,
// k = 32 voxels = two h-rows of the 8 x 16 plane tile per step.  The order of the voxels inside the k dimension is free as long as both
// operands use the same one: k-group g (= lane >> 4, 8 voxels) of a step is h-row 2 m + (g >> 1), w = 4 (g & 1) + {0..3} and + 8, so one
// transposing read (4 voxel rows x 32 bytes per 16-lane group) touches rows R .. R + 3 in group 0 and R + 4 .. R + 7 in group 1; with the
// 32-byte halves of a 64-byte LDS row flipped on bit 2 of the row index (applied on the DMA source, WkCfg rows) the two groups of a
// half-wave fall into different bank halves - the job the Cin / Cout half-selection does in the 32x32x16 form.
template <int G>
__device__ __forceinline__ void wk_compute16(const unsigned char* lds, const int (&xb)[3], int yb, const int (&pre_x)[8], int pre_y,
                                             f32x4 (&acc)[G == 3 ? 6 : 7][2][2], float (&bsum)[2], bool do_bias) {
    constexpr int NTAP = G == 3 ? 6 : 7, T0 = 7 * G, ROWB = 64;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    // dy fragment of k-step m, Cout half h: rows m * 32 + (lane part), 16-channel half h = +32 bytes (flipped with the row's swizzle bit,
    // which the lane part already carries: m * 32 rows never change it)
    auto load_a = [&](int m, s16x8 (&av)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned char* p0 = lds + yb + (h ? (pre_y ^ 32) : pre_y) + m * 32 * ROWB;
            s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
            s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 8 * ROWB));
            av[h] = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    };
    // x fragment of (k-step m, tap T0 + j), Cin half h: halo row (2 m + kh) * XW + kw + (lane part); the lane part's swizzle bit depends on
    // the row constant mod 8 (pre_x[c & 7]), the rest of the constant is a multiple of 8 rows
    auto load_b = [&](int st, int h) {
        const int m = st / NTAP, t = T0 + st % NTAP, kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
        const int c = (2 * m + kh) * wg::XW + kw;
        const unsigned char* pb = lds + xb[kd] + (h ? (pre_x[c & 7] ^ 32) : pre_x[c & 7]) + (c >> 3) * 8 * ROWB;
        s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pb);
        // (the second read is 8 rows on: same swizzle bit)
        s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pb + 8 * ROWB));
        return __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    constexpr int NS = 4 * NTAP, PF = 2, RING = PF + 1;      // x fragment pairs in flight ahead of their MFMAs
    s16x8 av[2], b[RING][2];
#pragma unroll
    for (int q = 0; q < PF; ++q) { b[q][0] = load_b(q, 0); b[q][1] = load_b(q, 1); }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        load_a(m, av);
        if constexpr (G == 3) {
            if (do_bias) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int q = 0; q < 8; ++q) bsum[h] += bf2f((unsigned short)av[h][q]);
            }
        }
#pragma unroll
        for (int j = 0; j < NTAP; ++j) {
            const int st = m * NTAP + j;
            if (st + PF < NS) { b[(st + PF) % RING][0] = load_b(st + PF, 0); b[(st + PF) % RING][1] = load_b(st + PF, 1); }
#pragma unroll
            for (int hi = 0; hi < 2; ++hi)
#pragma unroll
                for (int ho = 0; ho < 2; ++ho)
                    acc[j][ho][hi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av[ho]), __builtin_bit_cast(bf16x8_t, b[st % RING][hi]),
                                                                            acc[j][ho][hi], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// fp32 form of wk_compute: k-step ks = voxels 2 ks, 2 ks + 1 of the 8 x 16 plane tile (h-row ks >> 3, w = 2 (ks & 7) + (lane >> 5)); lane & 31 is
// the channel of both operands.  Row index = (compile-time constant c) + (lane >> 5); 128-byte rows flip their 64-byte halves on bit 1 of the row
// (the DMA's image, wg_slot_off<128>): addr(c) = pre[c & 3] + (c >> 2) * 512, as in the bf16 form.
template <int G>
__device__ __forceinline__ void wk_compute_f32(const unsigned char* lds, const int (&xb)[3], int yb, const int (&pre)[4], f32x16 (&acc)[G == 3 ? 6 : 7][1],
                                               float (&bsum)[2], bool do_bias) {
    constexpr int NTAP = G == 3 ? 6 : 7, T0 = 7 * G;
#pragma unroll
    for (int ks = 0; ks < 64; ++ks) {
        const int ks8 = ks >> 3, w2 = 2 * (ks & 7);
        const int cy = ks8 * wg::TW + w2;
        const float av = *reinterpret_cast<const float*>(lds + yb + pre[cy & 3] + (cy >> 2) * 512);
        if constexpr (G == 3) {
            if (do_bias) bsum[0] += av;
        }
#pragma unroll
        for (int j = 0; j < NTAP; ++j) {
            const int t = T0 + j, kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
            const int cx = (ks8 + kh) * wg::XW + kw + w2;
            const float bv = *reinterpret_cast<const float*>(lds + xb[kd] + pre[cx & 3] + (cx >> 2) * 512);
            acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j][0], 0, 0, 0);
        }
    }
}

struct WkArgs {
    SrcB s;
    const bf16_t* dy;
    float* dw;
    float* db;
    int N, D, H, W, Cout, nslab, dw_ld;
};

// everything a wave does, instantiated per tap group so that the accumulators are one fixed register set for the kernel's lifetime
template <int BLK, int G, bool S16 = false, bool F32 = false>
__device__ __forceinline__ void wk_run(const WkArgs& a, unsigned char* lds, int wv, int lane) {
    static_assert(!S16 || BLK == 32, "the 16x16x32 form is built for the 4-wave kernel");
    static_assert(!F32 || (BLK == 32 && !S16), "the fp32 form: 32 x 32 blocks");
    typedef WkCfg<BLK, F32> K;
    constexpr int EU = F32 ? 2 : 1;                      // 2-byte units per element: channel counts that are memory strides come in these (s.C0, s.C1)
    constexpr int NTAP = G == 3 ? 6 : 7, NW = K::NW, NH = K::NH, XROWB = K::XROWB, YROWB = K::YROWB, XRS = K::XRS, YRS = K::YRS;
    const SrcB& s = a.s;
    const int N = a.N, D = a.D, H = a.H, W = a.W, Cout = a.Cout;
    const int Cin = s.C0 + s.C1;
    const int ncib = Cin / (32 * EU), ncob = Cout / BLK;
    const int CoutB = EU * Cout;                         // voxel stride of dy in 2-byte units
    const int ncombo = ncob * ncib;
    const int wg_id = xcd_logical_id(blockIdx.x, gridDim.x);      // the (Cout, Cin) blocks of one slab (same planes) on one XCD / L2
    const int combo = wg_id % ncombo, slab = wg_id / ncombo;
    const int cib = combo % ncib, cob = combo / ncib;
    const int co0 = cob * BLK, ccr = cib * 32, cc = ccr * EU;          // ccr: first input channel of the block; cc: the same in 2-byte units

    const bool from0 = cc < s.C0;
    const bf16_t* sp = from0 ? s.p0 : s.p1;
    const int sC = from0 ? s.C0 : s.C1;
    const int coff = from0 ? cc : cc - s.C0;
    const int sh = (from0 && s.up0) ? 1 : 0;
    const int shd = sh & s.dsh;
    const int sD = D >> shd, sH = H >> sh, sW = W >> sh;

    const int r = lane & 31, hk = lane >> 5;
    // bias gradient = sum over the dy fragments: in the 6-tap group (it has registers to spare) of Cin block 0
    const bool do_bias = G == 3 && (a.db != nullptr) && cib == 0;

    f32x16 acc[S16 ? 1 : NTAP][NH];
    f32x4 acc16[S16 ? NTAP : 1][2][2];
    if constexpr (S16) {
#pragma unroll
        for (int q = 0; q < NTAP; ++q)
#pragma unroll
            for (int h = 0; h < 4; ++h) acc16[q][h >> 1][h & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int q = 0; q < NTAP; ++q)
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[q][h][k] = 0.f;
    }
    float bsum[2] = {0.f, 0.f};

    const int twn = W / wg::TW, thn = H / wg::TH;
    const int nunits = N * D * thn * twn;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(lds));
    const unsigned char* const zpage = zero_page_addr();

    // ---- DMA.  A unit's wave-instructions (X_INSTR for an x plane, Y_INSTR for a dy plane) are dealt round-robin over the NW waves.  The
    // issue side keeps a cursor (cn, ch0, cw0) of the column being issued and, per column, the lane constants of its pieces: element offset
    // inside a plane and whether that halo row lies inside the volume (bit j of x_ok); per plane what is left is a scalar base, one 64-bit add
    // and a select per instruction.
    // (One wave per SIMD: dealing the pieces so that the 6-tap wave, which has 16 MFMAs fewer per unit, carries 13 of the 28 was measured
    // SLOWER - profiles/r06_wgrad_w1_layers.log, variant c - and is gone.)
    constexpr int XPW_ = K::XPW, YPW_ = K::YPW;
    auto xid = [&](int k) { return wv + NW * k; };
    auto yid = [&](int k) { return (NW - 1 - wv) + NW * k; };      // dealt from the other end: the waves with one x instruction fewer go first
    int cn = 0, ch0 = 0, cw0 = 0;
    int x_off[XPW_], y_off[YPW_];
    unsigned x_ok = 0;
    auto col_setup = [&](int n, int h0, int w0) {
        cn = n; ch0 = h0; cw0 = w0;
        x_ok = 0;
#pragma unroll
        for (int k = 0; k < XPW_; ++k) {
            const int id = xid(k);
            const int i = id * 64 + lane;
            const int row = i / XRS, ps = i % XRS;
            // 128-byte rows flip their 64-byte halves on row bit 1 (wg_slot_off); S16: 64-byte rows flip their 32-byte halves on row bit 2
            const int ls = XRS == 8 ? (ps ^ (((row >> 1) & 1) << 2)) : (S16 ? (ps ^ (((row >> 2) & 1) << 1)) : ps);
            const int xh = row / wg::XW, xw = row % wg::XW;
            const int gh = h0 - 1 + xh, gw = w0 - 1 + xw;
            const bool ok = id < K::X_INSTR && row < wg::XROWS && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
            const int ghc = min(max(gh, 0), H - 1) >> sh, gwc = min(max(gw, 0), W - 1) >> sh;
            x_off[k] = ((ghc * sW + gwc) * sC + ls * 8) * 2;                    // bytes
            x_ok |= (ok ? 1u : 0u) << k;
        }
#pragma unroll
        for (int k = 0; k < YPW_; ++k) {
            const int id = yid(k);
            const int i = id * 64 + lane;
            const int row = i / YRS, ps = i % YRS;
            const int ls = YRS == 8 ? (ps ^ (((row >> 1) & 1) << 2)) : (S16 ? (ps ^ (((row >> 2) & 1) << 1)) : ps);
            y_off[k] = (((row >> 4) * W + (row & 15)) * CoutB + ls * 8) * 2;     // bytes
        }
    };
    // plane `gd` (may lie outside the volume: zeros) of the cursor's column into x slot `slot`
    // (round 5: through a buffer descriptor - a halo row outside the volume in h or w gets an out-of-range offset and lands as zeros, a
    // plane outside the volume in d a descriptor of zero records: no zero page, no 64-bit pointer select per lane)
    auto x_rsrc = [&](int gd) {
        const bool dok = (unsigned)gd < (unsigned)D;
        const int gdc = min(max(gd, 0), D - 1) >> shd;
        return dma_rsrc(sp + ((int64_t)cn * sD + gdc) * sH * sW * sC + coff, dok ? (int)DMA_OOB : 0);
    };
    auto y_rsrc = [&](int d) { return dma_rsrc(a.dy + ((((int64_t)cn * D + d) * H + ch0) * W + cw0) * CoutB + EU * co0); };
    auto x_piece = [&](i32x4 rs, int slot, int k) {                 // piece k of this wave (k is a constant at every call site)
        const int id = xid(k);
        if (id < K::X_INSTR)
            dma16_buf(rs, ((x_ok >> k) & 1) ? (unsigned)x_off[k] : DMA_OOB, __builtin_amdgcn_readfirstlane(lds0 + slot * K::XS_BYTES + id * 1024));
    };
    auto y_piece = [&](i32x4 rs, int ybuf, int k) {
        const int id = yid(k);
        if (id < K::Y_INSTR) dma16_buf(rs, (unsigned)y_off[k], __builtin_amdgcn_readfirstlane(lds0 + K::NXS * K::XS_BYTES + ybuf * K::YS_BYTES + id * 1024));
    };
    auto issue_x = [&](int gd, int slot) {
        const i32x4 rs = x_rsrc(gd);
#pragma unroll
        for (int k = 0; k < XPW_; ++k) x_piece(rs, slot, k);
    };
    auto issue_y = [&](int d, int ybuf) {
        const i32x4 rs = y_rsrc(d);
#pragma unroll
        for (int k = 0; k < YPW_; ++k) y_piece(rs, ybuf, k);
    };

    // ---- per-lane constants of the transposing fragment reads (see k_conv_wgrad_mfma)
    const int gq = lane >> 4, qd = (lane & 15) >> 2, pp = lane & 3;
    const int lrow = 8 * (gq >> 1) + qd;
    const int lslot_x = 2 * (gq & 1) + (pp >> 1), lslot_y = 2 * (gq & 1) + (pp >> 1);
    int pre_x[S16 ? 8 : 4];
    int pre_y;
    if constexpr (F32) {
        // row (m + hk) of a 128-byte-row image, channel r: one dword per lane (the x and the dy image share the layout)
#pragma unroll
        for (int m = 0; m < 4; ++m) pre_x[m] = (m + hk) * 128 + ((r * 4) ^ ((((m + hk) >> 1) & 1) << 6));
        pre_y = 0;
    } else if constexpr (S16) {
        // lane (g = k-group, qd = voxel row of the read, pp = 8-byte piece of the 32-byte half): voxel (h-row g >> 1, w = 4 (g & 1) + qd) of the step
        const int lx = (gq >> 1) * wg::XW + 4 * (gq & 1) + qd, ly = (gq >> 1) * wg::TW + 4 * (gq & 1) + qd;
#pragma unroll
        for (int m = 0; m < 8; ++m) pre_x[m] = ((lx + m) * 64 + pp * 8) ^ ((((lx + m) >> 2) & 1) << 5);
        pre_y = (ly * 64 + pp * 8) ^ (((ly >> 2) & 1) << 5);
    } else {
#pragma unroll
        for (int m = 0; m < 4; ++m) pre_x[m] = wg_slot_off<XROWB>(lrow + m, lslot_x) + (pp & 1) * 8;
        pre_y = wg_slot_off<YROWB>(lrow, lslot_y) + (pp & 1) * 8;
    }

    const int per = (nunits + a.nslab - 1) / a.nslab;
    int u = slab * per;
    const int u_end = min(nunits, u + per);
    if (u < u_end) {
        int d;
        {
            int q = u;
            d = q % D; q /= D;
            const int w0 = (q % twn) * wg::TW; q /= twn;
            col_setup(q / thn, (q % thn) * wg::TH, w0);
        }
        issue_x(d - 1, 0);
        issue_x(d, 1);
        issue_x(d + 1, 2);
        issue_y(d, 0);
        int xs = 2, yb = 0;                              // ring slot of the newest x plane (kd = 2) of the current unit; dy slot
#ifdef FMRI_PROF
        unsigned long long wprof[12] = {};               // [0] DMA wait, [1] barrier, [2] DMA issue (+ column set-up), [3] fragment reads + MFMAs, [4] fresh-column tail, [6] units
        PROF_T(wk0);
#endif
        for (; u < u_end; ++u) {
            PROF_T(w0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PROF_T(w1);
            __builtin_amdgcn_s_barrier();                // this unit's planes have landed everywhere; the previous unit is fully read
            PROF_T(w2);
            const bool more = u + 1 < u_end;
            const bool fresh = more && d + 1 == D;       // the next unit starts a new column: it needs three new planes, one slot is free
            i32x4 nx_rs = {0, 0, 0, 0}, ny_rs = {0, 0, 0, 0};
            if (more) {
                if (fresh) {                             // advance the cursor to the next column of the run
                    int w0 = cw0 + wg::TW, h0 = ch0, n = cn;
                    if (w0 == W) { w0 = 0; h0 += wg::TH; if (h0 == H) { h0 = 0; ++n; } }
                    col_setup(n, h0, w0);
                    d = -1;
                }
                if constexpr (!K::SPREAD) {
                    issue_x(fresh ? -1 : d + 2, (xs + 1) & 3);
                    issue_y(d + 1, yb ^ 1);
                }
            }
            PROF_T(w3);
            int xb[3] = {((xs + 2) & 3) * K::XS_BYTES, ((xs + 3) & 3) * K::XS_BYTES, xs * K::XS_BYTES};
#pragma unroll
            for (int k = 0; k < 3; ++k) asm volatile("" : "+s"(xb[k]));       // slot bases stay scalar: base + lane offset is added per read
            const int nslot = (xs + 1) & 3, nyb = yb ^ 1;
            auto hook = [&](int st) {
                if constexpr (K::SPREAD) {
                    // behind step 0: the next unit's two descriptors - ~60 scalar instructions (64-bit multiplies) that sat between the barrier
                    // and the unit's first MFMA; piece p of the next unit behind step SP0 + SPD p, the dy pieces first
                    constexpr int SPD = K::SPD;
                    if (st == 0 && more) {
                        nx_rs = x_rsrc(fresh ? -1 : d + 2);
                        ny_rs = y_rsrc(d + 1);
                    }
                    if (st >= K::SP0 && (st - K::SP0) % SPD == 0 && (st - K::SP0) / SPD < XPW_ + YPW_ && more) {
                        const int pc = (st - K::SP0) / SPD;
                        if (pc < YPW_) y_piece(ny_rs, nyb, pc);
                        else x_piece(nx_rs, nslot, pc - YPW_);
                    }
                }
            };
            if constexpr (F32) wk_compute_f32<G>(lds, xb, K::NXS * K::XS_BYTES + yb * K::YS_BYTES, pre_x, acc, bsum, do_bias);
            else if constexpr (S16) wk_compute16<G>(lds, xb, K::NXS * K::XS_BYTES + yb * K::YS_BYTES, pre_x, pre_y, acc16, bsum, do_bias);
            else wk_compute<BLK, G>(lds, xb, K::NXS * K::XS_BYTES + yb * K::YS_BYTES, pre_x, pre_y, acc, bsum, do_bias, hook);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PROF_T(w4);
#ifdef FMRI_PROF
            wprof[0] += w1 - w0; wprof[1] += w2 - w1; wprof[2] += w3 - w2; wprof[3] += w4 - w3; wprof[6] += 1;
#endif
            if (fresh) {
                __builtin_amdgcn_s_barrier();            // everybody is done with the old column's planes: its other two slots are free
                issue_x(0, (xs + 2) & 3);
                issue_x(1, (xs + 3) & 3);
                xs = (xs + 3) & 3;
            } else {
                xs = (xs + 1) & 3;
            }
            yb ^= 1;
            ++d;
#ifdef FMRI_PROF
            { PROF_T(w5); wprof[4] += w5 - w4; }
#endif
        }
#ifdef FMRI_PROF
        { PROF_T(wk1); wprof[5] = wk1 - wk0; }
        if (lane == 0) for (int i = 0; i < 7; ++i) atomicAdd(&g_prof[i], wprof[i]);
#endif
    }
    // ---- flush: D rows = co, cols = ci; fp32 atomics, 128 contiguous bytes per half-wave (deterministic mode: fixed-point shadow)
    const FmriDetCfg dc = g_det_cfg;
    if constexpr (S16) {
        // D rows = co (4 (lane >> 4) + reg inside the 16-channel half), cols = ci (lane & 15): 64 contiguous bytes per 16-lane group
#pragma unroll
        for (int j = 0; j < NTAP; ++j) {
            const int tap = 7 * G + j;
#pragma unroll
            for (int ho = 0; ho < 2; ++ho)
#pragma unroll
                for (int hi = 0; hi < 2; ++hi)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int co = co0 + ho * 16 + 4 * (lane >> 4) + reg;
                        const int ci = ccr + hi * 16 + (lane & 15);
                        fmri_grad_add(dc, &a.dw[((int64_t)tap * Cout + co) * a.dw_ld + ci], acc16[j][ho][hi][reg]);
                    }
        }
        if (do_bias) {
            // a lane summed its 8 voxels of every step for Cout lane & 15 of half h: the four k-groups meet
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float b = bsum[h];
                b += __shfl_down(b, 32);
                b += __shfl_down(b, 16);
                if (lane < 16) fmri_grad_add(dc, &a.db[co0 + h * 16 + lane], b);
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < NTAP; ++j) {
        const int tap = 7 * G + j;
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int co = co0 + h * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hk;
                const int ci = ccr + r;
                fmri_grad_add(dc, &a.dw[((int64_t)tap * Cout + co) * a.dw_ld + ci], acc[j][h][reg]);
            }
    }
    if (do_bias) {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            float b = bsum[h];
            b += __shfl_down(b, 32);
            if (hk == 0) fmri_grad_add(dc, &a.db[co0 + h * 32 + r], b);
        }
    }
}

template <int BLK, bool S16 = false, bool F32 = false>
__global__ void __launch_bounds__(256, (BLK == 64 || F32) ? 1 : 2) k_conv_wgrad_kd(WkArgs a) {     // BLK = 64: no second workgroup -> 512 registers per wave
    __shared__ __attribute__((aligned(16))) unsigned char lds[WkCfg<BLK, F32>::LDS_BYTES];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    switch (wv & 3) {                                     // tap group
        case 0: wk_run<BLK, 0, S16, F32>(a, lds, wv, lane); break;
        case 1: wk_run<BLK, 1, S16, F32>(a, lds, wv, lane); break;
        case 2: wk_run<BLK, 2, S16, F32>(a, lds, wv, lane); break;
        default: wk_run<BLK, 3, S16, F32>(a, lds, wv, lane); break;
    }
}

// ============================================================================ parity-form weight gradient, kd'-sharing (round 6)
// k_conv_wgrad_mfma<.., UPW> gives every (kd', parity, Cout block, Cin block) its own workgroup with 4 accumulators per wave: a 23 KiB x plane and
// a 16 KiB gathered dy plane are staged for 128 MFMAs - 305 B of LDS-DMA and 1.25 KB of LDS reads per MFMA, the matrix pipe 0.39-0.47 busy
// (profiles/r06_pmc_mfma.json).  Here a workgroup owns (parity, 64 Cout, 64 Cin) for BOTH kd' planes and all four (kh', kw'): 8 taps.  Walking up
// a column of low-res d-plane tiles, unit g needs the x planes g + pd - 1 (kd' = 0) and g + pd (kd' = 1): the second is the next unit's first, so
// ONE new x plane and one gathered dy plane per unit feed 256 MFMAs - 152 B of LDS-DMA per MFMA.  Wave = (kd', kh') tap pair x both kw' x the
// whole 64 x 64 block: 2 x 2 x 2 accumulators (128 registers); a dy fragment serves 4 MFMAs, an x fragment 2: 0.75 KB of LDS reads per MFMA.
// One workgroup per CU (3 x 23 + 2 x 16 KiB of LDS), one wave per SIMD; the DMA pieces and descriptors of the next unit are issued between the
// MFMAs as in k_conv_wgrad_kd<64>.  Same result layout as the UPW kernel: dWc[p][kd'][kh'][kw'][Cout][C0] fp32 (atomics).
struct WuCfg {
    static constexpr int NW = 4;
    static constexpr int ROWB = 128, RS = 8;                               // 64 channels x 2 B per x row and per dy row
    static constexpr int X_INSTR = (wg::XROWS * RS + 63) / 64;             // 23
    static constexpr int Y_INSTR = wg::YROWS * RS / 64;                    // 16
    static constexpr int XS_BYTES = X_INSTR * 1024, YS_BYTES = Y_INSTR * 1024;
    static constexpr int NXS = 3, NYS = 2;
    static constexpr int LDS_BYTES = NXS * XS_BYTES + NYS * YS_BYTES;      // 103,424
    static constexpr int XPW = (X_INSTR + NW - 1) / NW, YPW = Y_INSTR / NW;        // 6, 4
    static constexpr int PF = 3;                                           // x fragments in flight ahead of their MFMAs (5: level)
    static constexpr int SP0 = 1, SPD = 2;                                 // DMA piece p of the next unit behind step SP0 + SPD p (32 steps per unit; SPD 1 level, 3 slower: profiles/r06_upw_kd_tune.log)
};

// W8: eight waves - waves 0-3 own Cin half 0 of the block, waves 4-7 Cin half 1 (64 accumulator registers each): two waves per SIMD, so that one
// wave's DMA issue (10 pieces of ~100 cycles per 64-MFMA unit in the 4-wave form: 66 cycles per MFMA, profiles/r06_upw_kd_prof.log) runs under
// its partner's MFMAs; a dy fragment then serves 2 MFMAs instead of 4 (1 KB of LDS reads per MFMA instead of 0.75).  Measured level with the
// 4-wave form (62-66 cycles per MFMA slot, a 15 % barrier share: profiles/r06_upw_kd8_prof.log): kept as FMRI_UPW_KD=2, not the default.
template <int G, bool W8>          // wave group G: taps kd' = G >> 1, kh' = G & 1, kw' = 0 | 1
__device__ __forceinline__ void wu_run(const WkArgs& a, unsigned char* lds, int wv, int lane) {
    typedef WuCfg K;
    constexpr int KDP = G >> 1, KHP = G & 1, NW = W8 ? 8 : K::NW, ROWB = K::ROWB, RS = K::RS;
    constexpr int NI = W8 ? 1 : 2;                                         // Cin halves per wave
    constexpr int XPW = (K::X_INSTR + NW - 1) / NW, YPW = K::Y_INSTR / NW; // DMA pieces per wave and unit: 6 + 4 | 3 + 2
    const int ih0 = W8 ? (wv >> 2) : 0;                                    // this wave's (first) Cin half
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const SrcB& s = a.s;
    const int N = a.N, D = a.D, H = a.H, W = a.W, Cout = a.Cout;          // D, H, W: the LOW-res grid (dy is [N][2D][2H][2W][Cout])
    const int Cin = s.C0;
    const int ncib = Cin / 64, ncob = Cout / 64;
    const int ncombo = 8 * ncob * ncib;
    const int wg_id = xcd_logical_id(blockIdx.x, gridDim.x);
    int combo = wg_id % ncombo;
    const int slab = wg_id / ncombo;
    const int cib = combo % ncib; combo /= ncib;
    const int cob = combo % ncob;
    const int par = combo / ncob;                                          // output parity class (pd, ph, pw)
    const int pd = par >> 2, ph = (par >> 1) & 1, pw = par & 1;
    const int co0 = cob * 64, cc = cib * 64;
    const int r = lane & 31, hk = lane >> 5;
    const bool do_bias = G == 0 && a.db != nullptr && cib == 0 && ih0 == 0;            // each parity class covers its own eighth of the voxels

    f32x16 acc[2][2][NI];                                                  // [kw'][Cout half][Cin half]
#pragma unroll
    for (int q = 0; q < 4 * NI; ++q)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[q / (2 * NI)][(q / NI) & 1][q % NI][k] = 0.f;
    float bsum[2] = {0.f, 0.f};

    const int twn = W / wg::TW, thn = H / wg::TH;
    const int nunits = N * D * thn * twn;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(lds));

    // ---- DMA: per-lane constants of the column (cn, ch0, cw0); x rows outside the volume get an out-of-range offset (zeros)
    int cn = 0, ch0 = 0, cw0 = 0;
    int x_off[XPW], y_off[YPW];
    unsigned x_ok = 0;
    auto col_setup = [&](int n, int h0, int w0) {
        cn = n; ch0 = h0; cw0 = w0;
        x_ok = 0;
#pragma unroll
        for (int k = 0; k < XPW; ++k) {
            const int id = wv + NW * k;
            const int i = id * 64 + lane;
            const int row = i / RS, ps = i % RS;
            const int ls = ps ^ (((row >> 1) & 1) << 2);
            const int xh = row / wg::XW, xw = row % wg::XW;
            const int gh = h0 - 1 + xh, gw = w0 - 1 + xw;
            const bool ok = id < K::X_INSTR && row < wg::XROWS && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
            const int ghc = min(max(gh, 0), H - 1), gwc = min(max(gw, 0), W - 1);
            x_off[k] = ((ghc * W + gwc) * Cin + ls * 8) * 2;                // bytes
            x_ok |= (ok ? 1u : 0u) << k;
        }
#pragma unroll
        for (int k = 0; k < YPW; ++k) {
            const int id = (NW - 1 - wv) + NW * k;
            const int i = id * 64 + lane;
            const int row = i / RS, ps = i % RS;
            const int ls = ps ^ (((row >> 1) & 1) << 2);
            y_off[k] = (((row >> 4) * 4 * W + (row & 15) * 2) * Cout + ls * 8) * 2;        // voxel (2 h + ph, 2 w + pw) of the full-res plane
        }
    };
    auto x_rsrc = [&](int gd) {                           // low-res plane gd of sample cn (outside the volume: zero records)
        const bool dok = (unsigned)gd < (unsigned)D;
        const int gdc = min(max(gd, 0), D - 1);
        return dma_rsrc(s.p0 + ((int64_t)cn * D + gdc) * H * W * Cin + cc, dok ? (int)DMA_OOB : 0);
    };
    auto y_rsrc = [&](int d) {
        return dma_rsrc(a.dy + ((((int64_t)cn * 2 * D + 2 * d + pd) * 2 * H + 2 * ch0 + ph) * 2 * W + 2 * cw0 + pw) * Cout + co0);
    };
    auto x_piece = [&](i32x4 rs, int slot, int k) {
        const int id = wv + NW * k;
        if (id < K::X_INSTR)
            dma16_buf(rs, ((x_ok >> k) & 1) ? (unsigned)x_off[k] : DMA_OOB, __builtin_amdgcn_readfirstlane(lds0 + slot * K::XS_BYTES + id * 1024));
    };
    auto y_piece = [&](i32x4 rs, int ybuf, int k) {
        const int id = (NW - 1 - wv) + NW * k;
        dma16_buf(rs, (unsigned)y_off[k], __builtin_amdgcn_readfirstlane(lds0 + K::NXS * K::XS_BYTES + ybuf * K::YS_BYTES + id * 1024));
    };
    auto issue_x = [&](int gd, int slot) {
        const i32x4 rs = x_rsrc(gd);
#pragma unroll
        for (int k = 0; k < XPW; ++k) x_piece(rs, slot, k);
    };
    auto issue_y = [&](int d, int ybuf) {
        const i32x4 rs = y_rsrc(d);
#pragma unroll
        for (int k = 0; k < YPW; ++k) y_piece(rs, ybuf, k);
    };

    // ---- per-lane constants of the transposing fragment reads.  x: halo row of output voxel (h, w) under tap (kh', kw') = (h + kh' + ph) * XW + w + kw' + pw:
    // the parity's shift ph * XW + pw goes into the lane part, the rest is a compile-time constant c: addr(c) = pre_x[c & 3] + (c >> 2) * 4 * ROWB
    const int gq = lane >> 4, qd = (lane & 15) >> 2, pp = lane & 3;
    const int lrow = 8 * (gq >> 1) + qd;
    const int lslot = 2 * (gq & 1) + (pp >> 1);
    const int shift = ph * wg::XW + pw;
    int pre_x[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) pre_x[m] = (wg_slot_off<ROWB>(lrow + shift + m, lslot) + (pp & 1) * 8) ^ (ih0 << 6);       // Cin half 1: ^ 64
    const int pre_y = wg_slot_off<ROWB>(lrow, lslot) + (pp & 1) * 8;                                         // Cout half 1: ^ 64

    const int per = (nunits + a.nslab - 1) / a.nslab;
    int u = slab * per;
    const int u_end = min(nunits, u + per);
    if (u < u_end) {
        int d;
        {
            int q = u;
            d = q % D; q /= D;
            const int w0 = (q % twn) * wg::TW; q /= twn;
            col_setup(q / thn, (q % thn) * wg::TH, w0);
        }
        issue_x(d + pd - 1, 0);
        issue_x(d + pd, 1);
        issue_y(d, 0);
        int lo = 0, yb = 0;                               // ring slot of the kd' = 0 plane (kd' = 1: lo + 1, free: lo + 2, mod 3); dy slot
#ifdef FMRI_PROF
        unsigned long long wprof[12] = {};               // [0] DMA wait, [1] barrier, [2] column set-up, [3] fragment reads + MFMAs + DMA issue, [4] fresh-column tail, [6] units
        PROF_T(wk0);
#endif
        for (; u < u_end; ++u) {
            PROF_T(w0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PROF_T(w1);
            __builtin_amdgcn_s_barrier();                 // this unit's planes have landed everywhere; the previous unit is fully read
            PROF_T(w2);
            const bool more = u + 1 < u_end;
            const bool fresh = more && d + 1 == D;        // the next unit starts a new column: two new planes, one slot is free
            if (more && fresh) {
                int w0 = cw0 + wg::TW, h0 = ch0, n = cn;
                if (w0 == W) { w0 = 0; h0 += wg::TH; if (h0 == H) { h0 = 0; ++n; } }
                col_setup(n, h0, w0);
                d = -1;
            }
            PROF_T(w3);
            const int hi = lo == 2 ? 0 : lo + 1, fr = hi == 2 ? 0 : hi + 1;
            int xbase = (KDP ? hi : lo) * K::XS_BYTES;
            asm volatile("" : "+s"(xbase));
            const int ybase = K::NXS * K::XS_BYTES + yb * K::YS_BYTES;
            const int nyb = yb ^ 1;
            i32x4 nx_rs = {0, 0, 0, 0}, ny_rs = {0, 0, 0, 0};
            // fragments: dy (A) per k-step and Cout half, x (B) per (k-step, kw', Cin half) = step st, PF steps ahead in a register ring
            auto load_a = [&](int ks8, s16x8 (&av)[2]) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const unsigned char* p0 = lds + ybase + (h ? (pre_y ^ 64) : pre_y) + ks8 * wg::TW * ROWB;
                    s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
                    s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 4 * ROWB));
                    av[h] = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            };
            auto load_b = [&](int st) {
                const int ks8 = st / (2 * NI), j = (st / NI) & 1, i = st % NI;
                const int c = (ks8 + KHP) * wg::XW + j;
                const unsigned char* pb = lds + xbase + (i ? (pre_x[c & 3] ^ 64) : pre_x[c & 3]) + (c >> 2) * 4 * ROWB;
                s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)pb);
                s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pb + 4 * ROWB));
                return __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
            };
            constexpr int NS = 16 * NI, PF = K::PF, RING = PF + 1;
            s16x8 av[2][2], b[RING];
#pragma unroll
            for (int q = 0; q < PF; ++q) b[q] = load_b(q);
            load_a(0, av[0]);
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                const int ks8 = st / (2 * NI), j = (st / NI) & 1, i = st % NI, ab = ks8 & 1;
                if (st + PF < NS) b[(st + PF) % RING] = load_b(st + PF);
                if (st % (2 * NI) == (NI == 2 ? 1 : 0) && ks8 + 1 < 8) load_a(ks8 + 1, av[ab ^ 1]);
                if (st % (2 * NI) == 0 && do_bias) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int q = 0; q < 8; ++q) bsum[h] += bf2f((unsigned short)av[ab][h][q]);
                }
                const bf16x8_t bb = __builtin_bit_cast(bf16x8_t, b[st % RING]);
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    acc[j][h][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av[ab][h]), bb, acc[j][h][i], 0, 0, 0);
                // the next unit: its two descriptors behind step 0, its DMA pieces one at a time behind steps SP0 + SPD p (dy first)
                if (more) {
                    if (st == 0) {
                        nx_rs = x_rsrc(fresh ? pd - 1 : d + 1 + pd);
                        ny_rs = y_rsrc(d + 1);
                    }
                    if (st >= K::SP0 && (st - K::SP0) % K::SPD == 0 && (st - K::SP0) / K::SPD < XPW + YPW) {
                        const int pc = (st - K::SP0) / K::SPD;
                        if (pc < YPW) y_piece(ny_rs, nyb, pc);
                        else x_piece(nx_rs, fr, pc - YPW);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PROF_T(w4);
#ifdef FMRI_PROF
            wprof[0] += w1 - w0; wprof[1] += w2 - w1; wprof[2] += w3 - w2; wprof[3] += w4 - w3; wprof[6] += 1;
#endif
            if (fresh) {
                __builtin_amdgcn_s_barrier();             // everybody is done with the old column's planes: its kd' = 0 slot takes the new column's second plane
                issue_x(pd, lo);
                lo = fr;                                  // new column: kd' = 0 plane in the slot that was free, kd' = 1 plane in the old `lo`: the ring runs lo, lo + 1
                // (slots: new lo = fr, new hi must be fr + 1 mod 3 = old lo: holds)
            } else {
                lo = hi;
            }
            yb ^= 1;
            ++d;
#ifdef FMRI_PROF
            { PROF_T(w5); wprof[4] += w5 - w4; }
#endif
        }
#ifdef FMRI_PROF
        { PROF_T(wk1); wprof[5] = wk1 - wk0; }
        if (lane == 0) for (int i = 0; i < 7; ++i) atomicAdd(&g_prof[i], wprof[i]);
#endif
    }
    // ---- flush: dWc[p][kd'][kh'][kw'][Cout][C0], fp32 atomics (128 contiguous bytes per half-wave)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int co = co0 + h * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hk;
                    const int ci = cc + (ih0 + i) * 32 + r;
                    atomicAdd(&a.dw[((int64_t)((par * 2 + KDP) * 4 + KHP * 2 + j) * Cout + co) * Cin + ci], acc[j][h][i][reg]);
                }
    if (do_bias) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float bq = bsum[h];
            bq += __shfl_down(bq, 32);
            if (hk == 0) atomicAdd(&a.db[co0 + h * 32 + r], bq);
        }
    }
}

template <bool W8>
__global__ void __launch_bounds__(W8 ? 512 : 256, 1) k_conv_wgrad_up_kd(WkArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[WuCfg::LDS_BYTES];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    switch (wv & 3) {
        case 0: wu_run<0, W8>(a, lds, wv, lane); break;
        case 1: wu_run<1, W8>(a, lds, wv, lane); break;
        case 2: wu_run<2, W8>(a, lds, wv, lane); break;
        default: wu_run<3, W8>(a, lds, wv, lane); break;
    }
}

// dw[(kd*9+tap)][co0+co][cc+ci] += sum over the slabs of combo (kd, cob, cib).  Block = 32 float4 columns x 8 slab groups: every
// thread sums its share of the slabs for 4 consecutive elements (4 loads in flight), the 8 partial sums meet in LDS in a fixed order
// (bit-reproducible), one thread per column adds the total to dw.  Workgroup index of the wgrad launch = slab * ncombo + combo.
template <int CIB>
__global__ void __launch_bounds__(256)
k_wgrad_reduce(const float* __restrict__ ws, float* __restrict__ dw, int Cout, int Cin, int nslab, int ncombo, int ncob, int ncib,
               int planar, int ksp) {
    constexpr int PER4 = 9 * 64 * CIB / 4;                     // float4 columns per combo
    __shared__ float4 red[8][32];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t col = (int64_t)blockIdx.x * 32 + tx;        // global float4 column = combo * PER4 + e4
    const bool ok = col < (int64_t)ncombo * PER4;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    int combo = 0, e4 = 0;
    if (ok) {
        combo = (int)(col / PER4);
        e4 = (int)(col % PER4);
        const int nsl = nslab * ksp;                           // slabs of this combo: index (sl * ncombo + combo) * ksp + k
        const float4* base = reinterpret_cast<const float4*>(ws) + e4;
        int j = ty;
        for (; j + 24 < nsl; j += 32) {
            const int64_t i0 = ((int64_t)(j / ksp) * ncombo + combo) * ksp + (j % ksp);
            const int64_t i1 = ((int64_t)((j + 8) / ksp) * ncombo + combo) * ksp + ((j + 8) % ksp);
            const int64_t i2 = ((int64_t)((j + 16) / ksp) * ncombo + combo) * ksp + ((j + 16) % ksp);
            const int64_t i3 = ((int64_t)((j + 24) / ksp) * ncombo + combo) * ksp + ((j + 24) % ksp);
            const float4 v0 = base[i0 * PER4], v1 = base[i1 * PER4], v2 = base[i2 * PER4], v3 = base[i3 * PER4];
            a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
            a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
            a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
            a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
        }
        for (; j < nsl; j += 8) {
            const int64_t i0 = ((int64_t)(j / ksp) * ncombo + combo) * ksp + (j % ksp);
            const float4 v0 = base[i0 * PER4];
            a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        }
    }
    red[ty][tx] = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                              (a0.w + a1.w) + (a2.w + a3.w));
    __syncthreads();
    if (ty == 0 && ok) {
        float4 t = red[0][tx];
#pragma unroll
        for (int k = 1; k < 8; ++k) { t.x += red[k][tx].x; t.y += red[k][tx].y; t.z += red[k][tx].z; t.w += red[k][tx].w; }
        const int cib = combo % ncib;
        int q = combo / ncib;
        const int cob = q % ncob;
        const int kd = planar ? 1 : q / ncob;
        const int e = e4 * 4;
        const int ci = e % CIB, colo = (e / CIB) % 64, tap = e / (CIB * 64);
        float4* dst = reinterpret_cast<float4*>(dw + ((int64_t)(kd * 9 + tap) * Cout + cob * 64 + colo) * Cin + cib * CIB + ci);
        float4 d = *dst;
        d.x += t.x; d.y += t.y; d.z += t.z; d.w += t.w;
        *dst = d;
    }
}

}  // namespace

static int wgrad_f32_mfma() {           // FMRI_F32_MFMA=0: fp32 tensors stay on the VALU kernels of conv3d_generic.hip (rounds 1-5)
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("FMRI_F32_MFMA");
        v = e ? atoi(e) : 1;
    }
    return v;
}
bool conv3d_wgrad_mfma_ok(int C0, int C1, int Cout, int D, int H, int W, int dtype) {
    if (dtype == FMRI_F32)                // k_conv_wgrad_kd<32, false, F32>: (32 Cout, 32 Cin) blocks of fp32 planes
        return wgrad_f32_mfma() && !(C0 % 32) && !(C1 % 32) && C0 + C1 >= 32 && !(Cout % 32) && !(H % wg::TH) && !(W % wg::TW) &&
               (long long)H * W * (C0 > C1 ? C0 : C1) * 4 < (1ll << 31) && (long long)H * W * Cout * 4 < (1ll << 31);
    if (dtype != FMRI_BF16) return false;
    if ((C0 % 32) || (C1 % 32) || C0 + C1 < 32 || (Cout % 64)) return false;
    if ((H % wg::TH) || (W % wg::TW)) return false;
    // buffer-descriptor LDS-DMA: per-plane byte offsets are 32-bit and must stay below DMA_OOB = 2^31 (see conv3d_fwd_mfma_ok)
    if ((long long)H * W * (C0 > C1 ? C0 : C1) * 2 >= (1ll << 31) || (long long)H * W * Cout * 2 >= (1ll << 31)) return false;
    return true;
}

static void wgrad_plan(int C0, int C1, int Cout, int N, int D, int H, int W, int planar, bool with_ws, int& CIB, int& combos, int& nslab) {
    const int Cin = C0 + C1;
    const bool wide = (C0 % 64 == 0) && (C1 % 64 == 0);
    CIB = wide ? 64 : 32;
    combos = (planar ? 1 : 3) * (Cout / 64) * (Cin / CIB);
    const int ntiles = N * D * (H / wg::TH) * (W / wg::TW);
    // Workgroups per launch.  Atomic flush: every workgroup adds 9 x 64 x 64 fp32 accumulators (147 KB) at ~1.3 TB/s, so small layers
    // want few workgroups and large ones many (load balance / tail): measured per layer at BASELINE config 2 (tools/bench_conv.py,
    // FMRI_WGRAD_WGS sweep) >= 0.9 TFLOP layers are fastest at ~2048, the rest at ~768, the smallest at ~512.  With a slab workspace the
    // flush is a plain store + one reduction pass, and the sweep is repeated in FMRI_WGRAD_WGS_SLAB.
    static int forced_wgs = -1, forced_slab = -1;
    if (forced_wgs < 0) {
        const char* e = getenv("FMRI_WGRAD_WGS");
        forced_wgs = e ? atoi(e) : 0;
        const char* f = getenv("FMRI_WGRAD_WGS_SLAB");
        forced_slab = f ? atoi(f) : 0;
    }
    const double flops = 2.0 * (planar ? 9 : 27) * (double)Cin * Cout * (double)N * D * H * W;
    int target_wgs = flops >= 0.9e12 ? 2048 : (flops >= 0.06e12 ? 768 : 512);
    if (planar) target_wgs = 512;       // 2-D slices (one kd plane per combo): two workgroups per CU throughout (configs[3] step: 13.7 -> 13.5 ms)
    if (with_ws && forced_slab >= 64) target_wgs = forced_slab;
    if (!with_ws && forced_wgs >= 64) target_wgs = forced_wgs;
    nslab = (target_wgs + combos - 1) / combos;
    if (nslab > ntiles) nslab = ntiles;
    if (nslab < 1) nslab = 1;
}

int64_t conv3d_wgrad_mfma_ws_bytes(int C0, int C1, int Cout, int N, int D, int H, int W, int planar) {
    int CIB, combos, nslab;
    wgrad_plan(C0, C1, Cout, N, D, H, W, planar, true, CIB, combos, nslab);
    return (int64_t)combos * nslab * (CIB == 32 ? 2 : 1) * 9 * 64 * CIB * (int64_t)sizeof(float);
}

// launches of at least this many FLOPs take the kd-sharing kernel with its atomic flush, smaller ones the per-kd kernel with the slab flush
// (FMRI_WGRAD_KD_MIN_GFLOP overrides: A/B)
static double wgrad_kd_min_flops() {
    static double v = -1.0;
    if (v < 0) {
        const char* e = getenv("FMRI_WGRAD_KD_MIN_GFLOP");
        v = e ? atof(e) * 1e9 : 0.3e12;
    }
    return v;
}
// does this launch take the kd-sharing kernel (k_conv_wgrad_kd)?  use_ws: the slab flush was chosen for it
static bool wgrad_takes_kd(int C0, int C1, int Cout, int N, int D, int H, int W, int planar, int up0, bool use_ws, int* blk_out) {
    static int kd_mode = -1, kd_blk = 32;
    if (kd_mode < 0) {
        const char* e = getenv("FMRI_WGRAD_KD");
        kd_mode = e ? atoi(e) : 2;
        const char* f = getenv("FMRI_WGRAD_KD_BLK");
        kd_blk = (f && atoi(f) == 64) ? 64 : 32;
    }
    const double flops_ = 2.0 * (planar ? 9 : 27) * (double)(C0 + C1) * Cout * (double)N * D * H * W;
    const bool narrow = (C0 % 64) || (C1 % 64);
    const bool base = kd_mode && !planar && !use_ws && (!up0 || kd_mode == 3) && (C0 % 32 == 0) && (C1 % 32 == 0) && (Cout % 32 == 0);
    // FMRI_WGRAD_KD_BLK=64: the one-wave-per-SIMD (64 Cout, 32 Cin) form on every launch of >= 0.3 TFLOP whose Cout it tiles; 32 elsewhere
    const bool w1 = base && kd_blk == 64 && (Cout % 64 == 0) && (kd_mode == 3 || (kd_mode >= 2 && flops_ >= wgrad_kd_min_flops()));
    if (blk_out) *blk_out = w1 ? 64 : 32;
    return w1 || (base && ((kd_mode >= 2 && flops_ >= wgrad_kd_min_flops()) || (narrow && flops_ >= 0.1e12)));
}
// A 32-wide Cout (not a multiple of the per-kd kernel's 64-wide block) is fine where the kd-sharing kernel with its 32 x 32 blocks takes the
// launch: no workspace (slab flush), see wgrad_takes_kd.  Lets the channel-padded layer-graph engine pass a 32-channel dy as it is (it
// made a zero-extended 64-channel copy of it per layer and step: 1.4 ms of copies per Isensee step).
bool conv3d_wgrad_cout32_ok(int C0, int C1, int Cout, int N, int D, int H, int W, int dtype, int planar, int up0) {
    if (dtype != FMRI_BF16 || (C0 % 32) || (C1 % 32) || C0 + C1 < 32 || (Cout % 32) || (H % wg::TH) || (W % wg::TW)) return false;
    int blk = 32;
    return wgrad_takes_kd(C0, C1, Cout, N, D, H, W, planar, up0, false, &blk) && blk == 32;
}
// dw_ld: row length of the dw image the gradient is added into (>= C0 + C1; lets a launch over a subset of the input channels write
// its columns of the full [27][Cout][Cin] gradient: pass dw already offset to the first column)
int conv3d_wgrad_mfma_ld(const void* src0, int C0, int up0, int planar, const void* src1, int C1, const void* dy, float* dw, int dw_ld,
                         float* db, int N, int D, int H, int W, int Cout, void* workspace, int64_t workspace_bytes, hipStream_t st) {
    SrcB s{(const bf16_t*)src0, (const bf16_t*)src1, C0, C1, up0, planar ? 0 : 1, planar};
    const int Cin = C0 + C1;
    // The slab flush wins where the flush dominates (small layers: -25..35 %) and loses ~4 % on the >= 0.3 TFLOP layers, whose atomic
    // flush overlaps other workgroups' MFMA work while the reduction pass is a serial tail (tools/bench_conv.py, BENCH_WGRAD_WS=0|1).
    static int force_slab = -1;
    if (force_slab < 0) {
        const char* e = getenv("FMRI_WGRAD_SLAB");      // 1: always use the workspace when given (bit-reproducible), 0: heuristic
        force_slab = e ? atoi(e) : 0;
    }
    const double flops_ = 2.0 * (planar ? 9 : 27) * (double)(C0 + C1) * Cout * (double)N * D * H * W;
    const bool use_ws = workspace != nullptr && workspace_bytes >= conv3d_wgrad_mfma_ws_bytes(C0, C1, Cout, N, D, H, W, planar) &&
                        (force_slab == 1 || flops_ < wgrad_kd_min_flops());
    // kd-sharing kernels (a workgroup owns a (Cout, Cin) block for all 27 taps and walks columns; see k_conv_wgrad_kd).  Measured per layer
    // (profiles/r02_wgrad_kd_sharing_ab.log): the 4-wave form wins where the per-kd kernel has to fall back to 32-wide Cin blocks (enc0b
    // 32 -> 64 at full resolution: -17 %), is level with it on the 64- and 128-channel layers (0 ... -5 %) and loses on the fused-upsample
    // launches.  Inside the training step (after the issue-cursor rewrite of both kernels; same-box A/B of whole steps and of the per-layer
    // exclusive times, late round 2) the 4-wave form is level or ahead on EVERY launch that does not take the slab flush: dec0b 0.735 ->
    // 0.667 ms, the skip halves of dec0a / dec1a 1.33 -> 1.19 / 0.695 -> 0.657, dec1b 0.39 -> 0.367, nothing slower; weight-gradient
    // family 4.64 -> 4.34 ms per step, step 13.96 -> 13.73 ms.  FMRI_WGRAD_KD: 0 = off, 1 = only the layers with a 32-wide Cin block and
    // >= 0.1 TFLOP (the default until then), 2 (default) = also every launch of >= 0.3 TFLOP without fused up-sampling (below that the
    // per-kd kernel is ahead, with the slab flush or - callers without a workspace, e.g. the layer-graph engine - with the atomic one:
    // Isensee defaults 11.3 vs 11.65 ms per step), 3 = the fused-upsample launches too;
    // FMRI_WGRAD_KD_BLK = 32 (two workgroups per CU, 32 x 32 blocks) | 64 (one workgroup per CU, one wave per SIMD, 64 x 32 blocks; round 6).
    int kd_blk = 32;
    if (wgrad_takes_kd(C0, C1, Cout, N, D, H, W, planar, up0, use_ws, &kd_blk)) {
        const int combos_kd = (Cout / kd_blk) * (Cin / 32);
        const int nunits = N * D * (H / wg::TH) * (W / wg::TW);
        static int kd_wgs = -1;
        if (kd_wgs < 0) {
            const char* e = getenv("FMRI_WGRAD_KD_WGS");
            kd_wgs = e ? atoi(e) : 0;
        }
        // workgroups per launch: two per CU, i.e. ONE resident round (round 3: four per CU until then - every workgroup ends in a flush of
        // 27 x 32 x 32 fp32 atomics onto the same filter block as the others of its column, so half the workgroups are half the flush
        // traffic: weight-gradient family 4.64-4.73 -> 4.44-4.57 ms per step, step +0.2 ... 2.1 % in four interleaved same-box pairs;
        // one per CU leaves the second slot of the CUs empty: 5.1-5.2 ms).  FMRI_WGRAD_KD_WGS overrides.
        const int target = kd_wgs > 0 ? kd_wgs : (kd_blk == 64 ? 1 : 2) * fwd_cu_count();      // BLK = 64: one (512-register) workgroup per CU
        int nsl = (target + combos_kd - 1) / combos_kd;
        if (nsl > nunits) nsl = nunits;
        if (nsl < 1) nsl = 1;
        const WkArgs wa{s, (const bf16_t*)dy, dw, db, N, D, H, W, Cout, nsl, dw_ld};
        static int wg16 = -1;                // FMRI_WGRAD_MFMA16=1: v_mfma_f32_16x16x32_bf16 in the kd-sharing kernel (wk_compute16)
        if (wg16 < 0) {
            const char* e = getenv("FMRI_WGRAD_MFMA16");
            wg16 = e ? atoi(e) : 0;
        }
        if (kd_blk == 64) k_conv_wgrad_kd<64><<<combos_kd * nsl, 256, 0, st>>>(wa);
        else if (wg16) k_conv_wgrad_kd<32, true><<<combos_kd * nsl, 256, 0, st>>>(wa);
        else k_conv_wgrad_kd<32><<<combos_kd * nsl, 256, 0, st>>>(wa);
        FMRI_LAUNCH_CHECK();
        return FMRI_OK;
    }
    if (Cout % 64) return FMRI_E_SHAPE;                      // the per-kd kernel tiles Cout by 64
    int CIB, combos, nslab;
    wgrad_plan(C0, C1, Cout, N, D, H, W, planar, use_ws, CIB, combos, nslab);
    float* ws = use_ws ? (float*)workspace : nullptr;
    // Warp-specialised variant (8 waves, one workgroup per CU, 3-slot ring with the producers two units ahead): alone it is 2-5 % faster on
    // the layers with >= 0.3 TFLOP and a few % slower on the small ones, but inside the two-stream training step the whole-CU workgroups
    // interleave worse with the forward-type kernels of the other stream (-0.7 % per step) - kept as an option, off by default.
    // FMRI_WGRAD_WS = 1 always / 2 by size.
    static int wg_ws = -2;
    if (wg_ws == -2) {
        const char* e = getenv("FMRI_WGRAD_WS");
        wg_ws = e ? atoi(e) : 0;
    }
    if (wg_ws == 1 || (wg_ws == 2 && flops_ >= 0.3e12)) {
        if (CIB == 64) k_conv_wgrad_mfma<2, false, true><<<combos * nslab, 512, 0, st>>>(s, (const bf16_t*)dy, dw, db, N, D, H, W, Cout, nslab, ws, dw_ld);
        else k_conv_wgrad_mfma<1, false, true><<<combos * nslab, 512, 0, st>>>(s, (const bf16_t*)dy, dw, db, N, D, H, W, Cout, nslab, ws, dw_ld);
    } else if (CIB == 64) k_conv_wgrad_mfma<2, false><<<combos * nslab, wg::NTHREADS, 0, st>>>(s, (const bf16_t*)dy, dw, db, N, D, H, W, Cout, nslab, ws, dw_ld);
    else k_conv_wgrad_mfma<1, false><<<combos * nslab, wg::NTHREADS, 0, st>>>(s, (const bf16_t*)dy, dw, db, N, D, H, W, Cout, nslab, ws, dw_ld);
    if (use_ws) {
        const int ncob = Cout / 64, ncib = Cin / CIB;
        const int64_t cols = (int64_t)combos * 9 * 64 * CIB / 4;
        const int grid = (int)((cols + 31) / 32);
        if (CIB == 64) k_wgrad_reduce<64><<<grid, 256, 0, st>>>(ws, dw, Cout, dw_ld, nslab, combos, ncob, ncib, planar, 1);
        else k_wgrad_reduce<32><<<grid, 256, 0, st>>>(ws, dw, Cout, dw_ld, nslab, combos, ncob, ncib, planar, 2);
    }
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
// fp32 planes (3-D; a fused x2 up-sampled first source is fine): one kernel form for every size, atomic flush, one workgroup per CU
int conv3d_wgrad_mfma_f32(const void* src0, int C0, int up0, const void* src1, int C1, const void* dy, float* dw, float* db, int N, int D, int H, int W,
                          int Cout, hipStream_t st) {
    if (!conv3d_wgrad_mfma_ok(C0, C1, Cout, D, H, W, FMRI_F32)) return FMRI_E_SHAPE;
    SrcB s{(const bf16_t*)src0, (const bf16_t*)src1, 2 * C0, 2 * C1, up0, 1, 0};      // memory strides in 2-byte units (see wk_run, EU)
    const int combos = (Cout / 32) * ((C0 + C1) / 32);
    const int nunits = N * D * (H / wg::TH) * (W / wg::TW);
    int nsl = (fwd_cu_count() + combos - 1) / combos;
    if (nsl > nunits) nsl = nunits;
    if (nsl < 1) nsl = 1;
    const WkArgs wa{s, (const bf16_t*)dy, dw, db, N, D, H, W, Cout, nsl, C0 + C1};
    k_conv_wgrad_kd<32, false, true><<<combos * nsl, 256, 0, st>>>(wa);
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
int conv3d_wgrad_mfma(const void* src0, int C0, int up0, int planar, const void* src1, int C1, const void* dy, float* dw, float* db, int N,
                      int D, int H, int W, int Cout, void* workspace, int64_t workspace_bytes, hipStream_t st) {
    return conv3d_wgrad_mfma_ld(src0, C0, up0, planar, src1, C1, dy, dw, C0 + C1, db, N, D, H, W, Cout, workspace, workspace_bytes, st);
}

// ---- weight gradient of up-sample + concat + conv in parity form.  dwc: fp32 scratch [8][8][Cout][C0] (zeroed here).
namespace {
__device__ __forceinline__ int tap_class_w(int p, int k) { return p == 0 ? (k >= 1) : (k >= 2); }
// dw[kd,kh,kw][co][c0] += sum over the 8 parity classes of dWc[p][class of the tap under p]
__global__ void k_expand_up_wgrad(const float* __restrict__ dwc, float* __restrict__ dw, int Cout, int C0, int dw_ld, int planar) {
    const int64_t total = (int64_t)(planar ? 9 : 27) * Cout * C0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % C0);
        const int64_t q = i / C0;
        const int co = (int)(q % Cout);
        const int t = (int)(q / Cout) + (planar ? 9 : 0);          // planar: the centre kd plane of the 27-tap image
        const int kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
        float acc = 0.f;
        if (planar) {                                              // dWc [4 (ph,pw)][2][2][Cout][C0]
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int cls = tap_class_w(p >> 1, kh) * 2 + tap_class_w(p & 1, kw);
                acc += dwc[((int64_t)(p * 4 + cls) * Cout + co) * C0 + c0];
            }
        } else {
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int cls = (tap_class_w(p >> 2, kd) * 2 + tap_class_w((p >> 1) & 1, kh)) * 2 + tap_class_w(p & 1, kw);
                acc += dwc[((int64_t)(p * 8 + cls) * Cout + co) * C0 + c0];
            }
        }
        dw[((int64_t)t * Cout + co) * dw_ld + c0] += acc;
    }
}
}  // namespace

int conv3d_upcat_wgrad_mfma_ex(const void* src0_low, int C0, const void* src1, int C1, const void* dy, float* dw, float* db, float* dwc, int N,
                               int D, int H, int W, int Cout, int planar, int expand, void* workspace, int64_t workspace_bytes, hipStream_t st);
int conv3d_upcat_wgrad_mfma(const void* src0_low, int C0, const void* src1, int C1, const void* dy, float* dw, float* db, float* dwc, int N,
                            int D, int H, int W, int Cout, int planar, void* workspace, int64_t workspace_bytes, hipStream_t st) {
    return conv3d_upcat_wgrad_mfma_ex(src0_low, C0, src1, C1, dy, dw, db, dwc, N, D, H, W, Cout, planar, 1, workspace, workspace_bytes, st);
}
// expand = 0: the 64 parity-filter gradients stay in dwc [8][8][Cout][C0] for the caller (the folded transposed conv chains them through
// its own weights); columns [0, C0) of dw are not touched
int conv3d_upcat_wgrad_mfma_ex(const void* src0_low, int C0, const void* src1, int C1, const void* dy, float* dw, float* db, float* dwc, int N,
                               int D, int H, int W, int Cout, int planar, int expand, void* workspace, int64_t workspace_bytes, hipStream_t st) {
    // D,H,W = output dims (planar: D = slices, not doubled).  1. parity-filter gradients over the low-res grid
    if (hipMemsetAsync(dwc, 0, (size_t)(planar ? 16 : 64) * Cout * C0 * sizeof(float), st) != hipSuccess) return FMRI_E_LAUNCH;
    {
        SrcB s{(const bf16_t*)src0_low, nullptr, C0, 0, 0, planar ? 0 : 1, planar};
        const int Dl = planar ? D : D / 2, Hl = H / 2, Wl = W / 2;
        const bool wide = C0 % 64 == 0;
        const int CIB = wide ? 64 : 32;
        const int combos = (planar ? 4 : 16) * (Cout / 64) * (C0 / CIB);
        const int nunits = N * Dl * (Hl / wg::TH) * (Wl / wg::TW);
        int nslab = (1536 + combos - 1) / combos;
        if (nslab > nunits) nslab = nunits;
        if (nslab < 1) nslab = 1;
        float* const dbu = C1 == 0 ? db : nullptr;         // with skip channels the plain launch below produces the bias gradient
        const char* e = getenv("FMRI_WGRAD_WS");
        const int ws_mode = e ? atoi(e) : 0;
        // kd'-sharing form (k_conv_wgrad_up_kd, round 6): 3-D, 64-wide blocks of both channel counts.  FMRI_UPW_KD=0: the per-kd' kernel (rounds 2-5),
        // 1 (default): four waves, 2: eight waves.  Same-box A/B (profiles/r06_upw_kd8_ab.log): step 299.0 -> 300.8 / 301.0 patches/s, the three
        // launches' weight gradients 2.30 -> 2.21 ms; both forms run at 62-66 cycles per MFMA slot (profiles/r06_upw_kd_prof.log, r06_upw_kd8_prof.log):
        // at 152 B of LDS-DMA per MFMA the kernel asks the L2 for ~5.6 TB/s (2.5 GB per dec0a launch)
        static int upw_kd = -1;
        if (upw_kd < 0) {
            const char* f = getenv("FMRI_UPW_KD");
            upw_kd = f ? atoi(f) : 1;
        }
        if (upw_kd && !planar && wide && Cout % 64 == 0) {
            const int combos_kd = 8 * (Cout / 64) * (C0 / 64);
            int nsl = (fwd_cu_count() + combos_kd - 1) / combos_kd;
            if (nsl > nunits) nsl = nunits;
            if (nsl < 1) nsl = 1;
            const WkArgs wa{s, (const bf16_t*)dy, dwc, dbu, N, Dl, Hl, Wl, Cout, nsl, C0};
            if (upw_kd == 2) k_conv_wgrad_up_kd<true><<<combos_kd * nsl, 512, 0, st>>>(wa);        // FMRI_UPW_KD=2: the 8-wave form (two waves per SIMD): level
            else k_conv_wgrad_up_kd<false><<<combos_kd * nsl, 256, 0, st>>>(wa);
        } else if (wide && (ws_mode == 1 || (ws_mode == 2 && 2.0 * 8 * 8 * (double)C0 * Cout * N * Dl * Hl * Wl >= 0.3e12))) k_conv_wgrad_mfma<2, true, true><<<combos * nslab, 512, 0, st>>>(s, (const bf16_t*)dy, dwc, dbu, N, Dl, Hl, Wl, Cout, nslab, nullptr, C0);
        else if (wide) k_conv_wgrad_mfma<2, true><<<combos * nslab, wg::NTHREADS, 0, st>>>(s, (const bf16_t*)dy, dwc, dbu, N, Dl, Hl, Wl, Cout, nslab, nullptr, C0);
        else k_conv_wgrad_mfma<1, true><<<combos * nslab, wg::NTHREADS, 0, st>>>(s, (const bf16_t*)dy, dwc, dbu, N, Dl, Hl, Wl, Cout, nslab, nullptr, C0);
    }
    // 2. fold them into the 27-tap gradient of the up-sampled input channels (columns [0, C0) of dw)
    if (expand) k_expand_up_wgrad<<<grid_for((int64_t)(planar ? 9 : 27) * Cout * C0, 256, 1024), 256, 0, st>>>(dwc, dw, Cout, C0, C0 + C1, planar);
    FMRI_LAUNCH_CHECK();
    if (C1 == 0) return FMRI_OK;
    // 3. the skip channels: plain weight gradient into columns [C0, C0+C1), bias gradient included
    return conv3d_wgrad_mfma_ld(src1, C1, 0, planar, nullptr, 0, dy, dw + C0, C0 + C1, db, N, D, H, W, Cout, workspace, workspace_bytes, st);
}
