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
#include "mfma_common.h"

namespace {


struct FwdItem {          // one (tile, Cout block, 32-channel chunk) unit of the persistent stream
    int n, d0, h0, w0, co0, ch, par;
};

// a lane's value of `v` moved by a DPP control (quad_perm 0x00-0xFF, row_half_mirror 0x141, ...): one VALU instruction, no LDS
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// Optional extra outputs of a plain (MODE 0, no residual) warp-specialised launch, produced from the tile while it sits in LDS as bf16
// - i.e. from exactly the values that are stored to y - so that the HBM-bound consumers of y need no pass of their own:
//   pool    [N][D/2][H/2][W/2][Cout]   2x2x2 max of y            (MaxPooling3D behind an encoder block, reference unet.py:51)
//   logits  [N*D*H*W] fp32             sum_c w1[c] * y[v][c] + b1  (final Conv3D(1, (1,1,1)), reference unet.py:68); needs Cout == BN
struct FwdTail {
    bf16_t* pool;
    const float* w1;
    const float* b1;
    float* logits;
    // RES launches only: bias per BORDER CLASS of the output voxel, [27 = (cd*3 + ch)*3 + cw][Cout] fp32, c = 0 first voxel of the axis /
    // 1 interior / 2 last voxel (round 3: Deconvolution3D folded into the following conv - the transposed conv's bias reaches an output
    // voxel through the in-volume taps only, so the effective bias differs on the volume's faces, edges and corners); nullptr: `bias`
    const float* bias27;
    // normalisation tails of the asynchronous epilogue (round 3, EPI 4 / 5 / 6): per-(group, channel) sums added to nws[g][Cout][2] (double,
    // zeroed by the caller), g = sample (n_per = 1, instance norm) or 0 (batch norm).  EPI 4 / 6: {sum y, sum y^2} of the stored bf16 values.
    // EPI 5 (input-gradient launch in front of a normalised block): {sum dz, sum dz * x}, dz = dy * act'(z), z = fma(x, sc, sh) with the
    // block's scale / shift nss[g][Cout][2] = {sc, sh} (k_norm_scale_shift: the operations of the forward's apply pass, the same z bit for
    // bit); x (the block's conv output) comes in `mask`
    // The sums go to the workgroup's OWN slot, nws[1 + blockIdx.x][G][Cout][2] (slot 0 is the total the caller folds them into): every
    // workgroup of a launch flushes at the same moments (its last tile; the tile where the sample changes), and device-scope fp64 atomics of
    // 1,024 waves on the same 128 addresses cost 30-45 us per flush (measured: instance norm, four flushes per workgroup, +50-80 % on the
    // launch).  Only the four producer waves of a workgroup meet in a slot.
    double* nws;
    int n_per;
    int n_grp;              // G
    const float* nss;
    int prio;               // != 0: the MFMA waves raise their issue priority (s_setprio 3) over the producer wave on their SIMD (FMRI_FWD_PRIO, A/B)
};

// MODE selects what the 3x3x3 machinery computes:
//   0  the plain convolution (forward, and dgrad with tap-flipped transposed weights);
//   1  "up-forward": the conv of a nearest x2 up-sampled source WITHOUT the redundant taps.  Output voxel 2g+p (p = parity in
//      {0,1}^3) only sees the low-res voxels g+e, e in {-1,0} (p = 0) or {0,+1} (p = 1) per axis, with pre-summed weights, so each
//      parity class is a 2x2x2-tap conv of the LOW-res tensor: the kernel runs on the low-res grid (D,H,W = low-res dims), the
//      Cout blocks enumerate (parity, channel block), 4 (kd',kh') phases x 2 kw' taps per chunk instead of 9 x 3, and the
//      epilogue scatters to voxel 2g+p of the [2D][2H][2W] output (no bias / activation: the result is a partial sum that the
//      MODE 0 + RES launch over the skip channels finishes);
//   2  "up-backward": gradient of that conv w.r.t. the low-res tensor = the same 2x2x2-tap structure over the space-to-depth view
//      of dy: chunk -> (parity, 32-channel slice), halo rows are gathered from voxel 2g+p of dy, taps mirrored.
// RES (MODE 0): the epilogue adds `residual` (same layout as y, may alias it) before bias + activation, in fp32.
// CUBE: the workgroup tile is 8 x 8 x 8 instead of 4 x 8 x 16 voxels (halo 10^3 = 1000 rows): the shape of the deepest levels of
// deep models (e.g. 8^3 at level 4 of a 128^3 Isensee net), whose W is not a multiple of 16.  A 32-voxel column tile is then 4 h-rows
// of 8 voxels; everything else is the same machinery.
// TAIL (round 6; planar plain launches): the epilogue also writes the 2x2 max-pooled slices (tail.pool, [N][D][H/2][W/2][Cout]) and / or the
// logits of a final 1x1 conv to one label (tail.logits; needs Cout == BN) from the staged tile - what EPI 0 / 2 of k_conv_fwd_ws do for the
// 3-D launches (MaxPooling2D behind an encoder block and the final Conv2D of unet_model_2d: reference unet/unet.py:67,82).
template <int NT, bool PL, int MODE, bool RES, bool CUBE = false, bool FH = false, bool TAIL = false>  // NT = 32-wide Cout tiles per workgroup (BN = 32*NT); PL = planar; FH: see k_conv_fwd_ws
__global__ void __launch_bounds__(fw::NTHREADS)
k_conv_fwd_mfma(SrcB s, const bf16_t* __restrict__ wt, const float* __restrict__ bias, const bf16_t* __restrict__ mask,
                const bf16_t* residual, bf16_t* y, int N, int D, int H, int W, int Cout, int act, float alpha, FwdTail tail) {
    static_assert(!TAIL || (PL && MODE == 0 && !RES && !CUBE), "the tails of this kernel: planar plain launches");
    constexpr int NTHREADS = fw::NTHREADS;
    constexpr int TD = CUBE ? 8 : fw::TD, TH = CUBE ? 8 : fw::TH, TW = CUBE ? 8 : fw::TW;
    constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2, HVOX = HD * HH * HW;
    constexpr int H_INSTR = (HVOX * 4 + 63) / 64, HALO_BYTES = H_INSTR * 1024;
    static_assert(!(RES && MODE != 0) && !(CUBE && (PL || MODE != 0 || RES)), "unsupported combination");
    (void)NTHREADS;
    constexpr bool PAR = MODE != 0;
    constexpr int BN = 32 * NT;
    constexpr int NKW = PAR ? 2 : 3;                     // kw taps per phase
    constexpr int FILT_BYTES = 3 * BN * 64;              // ring slot: one (kd,kh) slab of up to 3 kw taps x BN rows x 64 B
    constexpr int F_INSTR = NKW * BN * 64 / 1024;        // DMA wave-instructions per slab: 12 / 6 (8 / 4 in the up modes)
    constexpr int F_PER_WAVE = (F_INSTR + 7) / 8;        // 2 or 1 (short waves re-issue their first instruction)
    constexpr int NPAR = PL ? 4 : 8;                     // parity classes of the up modes (planar: (ph,pw) only)
    constexpr int NPH = PAR ? (PL ? 2 : 4) : (PL ? 3 : 9);   // phases per chunk = (kd,kh) rows that exist
    constexpr int PH0 = PL ? 3 : 0;                      // first (kd,kh) row (planar: kd = 1)
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * HALO_BYTES + 2 * FILT_BYTES];
    const unsigned char* const zpage = zero_page_addr();

    const int Cin = s.C0 + s.C1;
    const int kpc = MODE == 2 ? (s.C0 >> 5) : 1;         // up-backward: chunks per parity class (s.C0 = channels of dy)
    const int nch = MODE == 2 ? NPAR * kpc : (Cin >> 5);
    const int Krow = MODE == 2 ? s.C0 : Cin;             // k-extent of one filter row in global memory
    const int cbn = Cout / BN;
    const int ncb = MODE == 1 ? NPAR * cbn : cbn;
    const int twn = W / TW, thn = H / TH, tdn = D / TD;
    const int npairs = N * tdn * thn * twn * ncb;

    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 31, hk = lane >> 5;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(lds));
    const unsigned ldsf0 = lds0 + 2 * HALO_BYTES;

    auto decode = [&](int pair, int ch) {
        FwdItem it;
        it.ch = ch;
        const int cb = pair % ncb;
        it.par = MODE == 1 ? cb / cbn : 0;
        it.co0 = (MODE == 1 ? cb % cbn : cb) * BN;
        int q = pair / ncb;
        // tiles are numbered in compact 2 (d) x 4 (h) x 4 (w) blocks when the grid allows it: the ~32 consecutive pairs an XCD works on at any
        // time then share their halo planes / rows / columns in its L2 instead of being a one-tile-thick row of the volume
        if (((twn | thn) & 3) == 0 && (tdn & 1) == 0) {
            const int in = q & 31;
            int blk = q >> 5;
            const int nbw = twn >> 2, nbh = thn >> 2, nbd = tdn >> 1;
            const int bw_ = blk % nbw; blk /= nbw;
            const int bh_ = blk % nbh; blk /= nbh;
            it.w0 = (bw_ * 4 + (in & 3)) * TW;
            it.h0 = (bh_ * 4 + ((in >> 2) & 3)) * TH;
            it.d0 = ((blk % nbd) * 2 + (in >> 4)) * TD;
            it.n = blk / nbd;
            return it;
        }
        it.w0 = (q % twn) * TW; q /= twn;
        it.h0 = (q % thn) * TH; q /= thn;
        it.d0 = (q % tdn) * TD;
        it.n = q / tdn;
        return it;
    };

    // ---- per-lane constants of the DMA address generation (hoisted: the phase loop only adds wave-uniform bases)
    unsigned f_voff[F_PER_WAVE];   // byte offset of this lane's 16 B inside a filter slab's global image
#pragma unroll
    for (int k = 0; k < F_PER_WAVE; ++k) {
        const int instr = (wv + 8 * k) % F_INSTR;                  // waves past the slab's end repeat an instruction (k = 0 only)
        const int i = instr * 64 + lane;
        const int row = i >> 2, ps = i & 3;
        const int ls = ps ^ ((row >> 2) & 3);
        f_voff[k] = (unsigned)(((row / BN) * Cout + (row % BN)) * Krow + ls * 8) * 2u;
    }
    int h_pack[9];                 // halo piece ph: hd | hh<<4 | hw<<8 | ls<<13 | valid<<15 | edge<<16 (faces of the halo box the row lies on, see k_conv_fwd_ws)
#pragma unroll
    for (int ph = 0; ph < 9; ++ph) {
        const int i = (ph * 8 + wv) * 64 + lane;
        const int hv = i >> 2, ps = i & 3;
        const int ls = ps ^ ((hv >> 2) & 3);
        const int hvc = hv < HVOX ? hv : 0;
        const int hw_ = hvc % HW, hq = hvc / HW;
        const int hd_ = hq / HH, hh_ = hq % HH;
        const int edge = (hd_ == 0 ? 1 : 0) | (hd_ == HD - 1 ? 2 : 0) | (hh_ == 0 ? 4 : 0) | (hh_ == HH - 1 ? 8 : 0) | (hw_ == 0 ? 16 : 0) | (hw_ == HW - 1 ? 32 : 0);
        h_pack[ph] = hd_ | (hh_ << 4) | (hw_ << 8) | (ls << 13) | ((hv < HVOX ? 1 : 0) << 15) | (edge << 16);
    }
    // filter slab of phase `pl` (= (kd,kh) row PH0 + pl; up modes: (parity, kd', kh')) for item `it` into filter ring slot `fb`:
    // scalar base + constant lane offset
    auto issue_filter = [&](const FwdItem& it, int pl, int fb) {
        int64_t slab;
        int koff;
        if constexpr (MODE == 0) { slab = PH0 + pl; koff = it.ch << 5; }
        else if constexpr (MODE == 1) { slab = it.par * NPH + pl; koff = it.ch << 5; }
        else { slab = (it.ch / kpc) * NPH + pl; koff = (it.ch % kpc) << 5; }
        const bf16_t* const base = wt + ((slab * NKW * Cout + it.co0) * Krow + koff);
#pragma unroll
        for (int k = 0; k < F_PER_WAVE; ++k)
            if (k == 0 || wv + 8 * k < F_INSTR)
                dma16_s(base, f_voff[k], __builtin_amdgcn_readfirstlane(ldsf0 + fb * FILT_BYTES + ((wv + 8 * k) % F_INSTR) * 1024));
    };
    // Halo source pointers, one per piece (1/9 of the tile) and lane.  They are worked out in full only when the stream moves to
    // a new tile or crosses from source 0 to source 1 of a concatenation; otherwise the next chunk is 32 channels further on.
    const bf16_t* hp[9];
    auto halo_src = [&](const FwdItem& it, int pk) -> const bf16_t* {
        if constexpr (MODE == 2) {
            // space-to-depth view of dy [N][2D][2H][2W][C0]: chunk -> (parity, 32-channel slice); low-res halo voxel g reads 2g+p
            const int p = it.ch / kpc, coff = (it.ch % kpc) << 5;
            const int gd = it.d0 - 1 + (pk & 15), gh = it.h0 - 1 + ((pk >> 4) & 15), gw = it.w0 - 1 + ((pk >> 8) & 31);
            const int ls = (pk >> 13) & 3;
            const bool ok = ((pk >> 15) & 1) && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
            // planar: p = (ph, pw), the slice axis is not doubled
            const int sd = PL ? min(max(gd, 0), D - 1) : 2 * min(max(gd, 0), D - 1) + (p >> 2);
            const int sh = 2 * min(max(gh, 0), H - 1) + ((p >> 1) & 1), sw = 2 * min(max(gw, 0), W - 1) + (p & 1);
            const int64_t off = ((((int64_t)it.n * (PL ? D : 2 * D) + sd) * 2 * H + sh) * 2 * W + sw) * s.C0 + coff + ls * 8;
            return ok ? s.p0 + off : (const bf16_t*)zpage;
        }
        const int cc = it.ch << 5;
        const bool from0 = cc < s.C0;
        const bf16_t* sp = from0 ? s.p0 : s.p1;
        const int sC = from0 ? s.C0 : s.C1;
        const int coff = from0 ? cc : cc - s.C0;
        const int sh = (from0 && s.up0) ? 1 : 0;
        const int shd = sh & s.dsh;
        const int sD = D >> shd, sH = H >> sh, sW = W >> sh;
        const int gd = it.d0 - 1 + (pk & 15), gh = it.h0 - 1 + ((pk >> 4) & 15), gw = it.w0 - 1 + ((pk >> 8) & 31);
        const int ls = (pk >> 13) & 3;
        const bool ok = ((pk >> 15) & 1) && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
        const int gdc = min(max(gd, 0), D - 1) >> shd, ghc = min(max(gh, 0), H - 1) >> sh, gwc = min(max(gw, 0), W - 1) >> sh;
        const int off = ((gdc * sH + ghc) * sW + gwc) * sC + coff + ls * 8;      // inside one sample: < 2^31 elements
        const bf16_t* real = sp + (int64_t)it.n * sD * sH * sW * sC + off;
        return ok ? real : (const bf16_t*)zpage;
    };
    auto issue_halo = [&](int ph, int hb) {            // ph is a compile-time constant at every call site
#ifdef FMRI_CHECK
        {
            const char* q = (const char*)hp[ph];
            const char* z = (const char*)zpage;
            const char* a0 = (const char*)s.p0; const char* a1 = (const char*)s.p1;
            const long long e0 = (long long)N * (D >> (s.up0 & s.dsh)) * (H >> s.up0) * (W >> s.up0) * s.C0 * 2;
            const long long e1 = (long long)N * D * H * W * s.C1 * 2;
            const bool okz = q >= z && q + 16 <= z + sizeof(g_zero_page);
            const bool ok0 = q >= a0 && q + 16 <= a0 + e0;
            const bool ok1 = s.C1 > 0 && q >= a1 && q + 16 <= a1 + e1;
            if (!(okz || ok0 || ok1)) {
                if (atomicAdd((unsigned long long*)&g_chk[0], 1ull) == 0) {
                    g_chk[1] = ph; g_chk[2] = wv; g_chk[3] = lane; g_chk[4] = (long long)q; g_chk[5] = (long long)z; g_chk[6] = (long long)a0;
                    g_chk[7] = e0; g_chk[8] = blockIdx.x; g_chk[9] = hb;
                }
                hp[ph] = (const bf16_t*)zpage;
            }
        }
#endif
        // planar (2-D slices): only the centre kd taps exist, so halo planes 0 and 5 (rows 0-179 and 900-1079) are never read: the
        // DMA instructions that lie entirely inside them (16 rows each) are not issued
        if (PL && (ph * 8 + wv < 11 || ph * 8 + wv >= 57)) return;
        if (ph * 8 + wv < H_INSTR) dma16(hp[ph], __builtin_amdgcn_readfirstlane(lds0 + hb * HALO_BYTES + (ph * 8 + wv) * 1024));
    };
    // FH (round 4): fast halo addressing as in k_conv_fwd_ws - here EVERY wave issues halo pieces, so the ~50 vector instructions of a freshly
    // worked-out address sat in front of the MFMAs of all eight waves (the 2-D layers of configs[3] are one- and two-chunk layers at 256 x 256)
    constexpr bool FASTH_CT = FH && MODE == 0 && !CUBE;
    unsigned hoff[FASTH_CT ? 9 : 1];
    auto fast_base = [&](const FwdItem& it, unsigned& tmask) -> const char* {
        tmask = __builtin_amdgcn_readfirstlane((it.d0 == 0 ? 1u : 0u) | (it.d0 + TD == D ? 2u : 0u) | (it.h0 == 0 ? 4u : 0u) | (it.h0 + TH == H ? 8u : 0u) |
                                               (it.w0 == 0 ? 16u : 0u) | (it.w0 + TW == W ? 32u : 0u));
        const unsigned long long a = reinterpret_cast<unsigned long long>(s.p0 + ((((int64_t)it.n * D + it.d0 - 1) * H + it.h0 - 1) * W + it.w0 - 1) * s.C0 + (it.ch << 5));
        const unsigned alo = __builtin_amdgcn_readfirstlane((unsigned)a), ahi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        return reinterpret_cast<const char*>(((unsigned long long)ahi << 32) | alo);
    };
    auto issue_halo_fast = [&](int ph, int hb, const char* sb, unsigned tmask) {
        const int instr = ph * 8 + wv;
        if (PL && (instr < 11 || instr >= 57)) return;
        if (instr >= H_INSTR) return;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + hb * HALO_BYTES + instr * 1024);
        // (round 5: through a buffer descriptor based at the box corner - a row outside the volume gets an out-of-range offset and lands as
        // zeros; the round-4 form selected the zero page with a 64-bit pointer per lane)
        unsigned off = hoff[ph];
        if (tmask != 0) {
            const int pk = h_pack[ph];
            const bool bad = !((pk >> 15) & 1) || (((unsigned)pk >> 16) & tmask) != 0;
            off = bad ? DMA_OOB : off;
        }
        dma16_buf(dma_rsrc(sb), off, dst);
    };

    // TAIL: this lane's 8 weights of the final 1x1 conv (the store loop gives a lane piece q = lane % (BN / 8) of a voxel in every iteration)
    float4 w1a = make_float4(0.f, 0.f, 0.f, 0.f), w1b = w1a;
    float b1v = 0.f;
    if constexpr (TAIL) {
        if (tail.logits) {
            const int q = lane % (32 * NT / 8);
            w1a = *reinterpret_cast<const float4*>(tail.w1 + q * 8);
            w1b = *reinterpret_cast<const float4*>(tail.w1 + q * 8 + 4);
            b1v = tail.b1[0];
        }
    }
    f32x16 acc[2][NT];
    // The accumulators of a tile START from its bias (MFMA D rows = output channel (reg&3) + 8*(reg>>2) + 4*hk), so the
    // epilogue has no bias add and needs no bias registers during the phase loop; the next tile's values are fetched inside the
    // epilogue, under the LDS transposition.  The activation is branch-free: act(v) = max(v, fma(v, act_s, +0)) with act_s = 0 (ReLU),
    // alpha (LeakyReLU) or 1 (none) - per-element `if (act == ...)` compiled to ~500 scalar compare/branch pairs per tile.
    const float act_s = act == FMRI_ACT_RELU ? 0.f : (act == FMRI_ACT_LEAKY ? alpha : 1.f);
    auto load_bias = [&](int co0, float4 (&bv)[NT][4]) {
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                bv[c][gq] = bias ? *reinterpret_cast<const float4*>(bias + co0 + c * 32 + 8 * gq + 4 * hk) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto init_acc = [&](const float4 (&bv)[NT][4]) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int c = 0; c < NT; ++c)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    acc[j][c][4 * gq] = bv[c][gq].x;
                    acc[j][c][4 * gq + 1] = bv[c][gq].y;
                    acc[j][c][4 * gq + 2] = bv[c][gq].z;
                    acc[j][c][4 * gq + 3] = bv[c][gq].w;
                }
    };

    // per-lane halo index (before adding the tap offset) of this lane's voxel in the wave's two column tiles
    // Lane r of a 32-voxel column tile covers h-row (r>>4) and w = wl.  The second h-row is rotated by 2 (= HW mod 16) so that
    // the halo rows read by one ds_read_b128 lane group are distinct mod 16: with the plain map rows r and r+18 collided on
    // 2 of 16 lanes per group (rocprofv3: SQ_LDS_BANK_CONFLICT = 33 % of SQ_LDS_IDX_ACTIVE); the rotation makes it conflict-free.
    // (CUBE: lane r covers h-row r>>3 of its tile's four and w = r&7; the rarely used variant lives with the bank conflicts)
    const int wl = CUBE ? (r & 7) : ((r >> 4) ? (((r & 15) + 16 - (HW & 15)) & 15) : (r & 15));
    // column tile rt (0..15) -> (d, first h-row) inside the workgroup tile, lane -> h-row inside the column tile
    auto tile_d = [](int rt) { return CUBE ? rt >> 1 : rt >> 2; };
    auto tile_h = [](int rt, int rr) { return CUBE ? 4 * (rt & 1) + (rr >> 3) : 2 * (rt & 3) + (rr >> 4); };
    auto lane_w = [](int rr) { return CUBE ? (rr & 7) : ((rr >> 4) ? (((rr & 15) + 16 - (HW & 15)) & 15) : (rr & 15)); };
    int hv0[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rt = 2 * wv + j;
        hv0[j] = (tile_d(rt) * HH + tile_h(rt, r)) * HW + wl;
    }
    // filter fragment read offset inside a slab for k-step 0 (k-step 1 = same address with bit 5 flipped): rows kw*BN + c*32 + r
    // keep the swizzle term of row r because kw*BN + c*32 is a multiple of 16
    const int fa[2] = {swz64(r, hk), swz64(r, hk) ^ 32};

    int pair = xcd_logical_id(blockIdx.x, gridDim.x);            // neighbouring (tile, Cout block) pairs on one XCD / L2
    if (pair >= npairs) return;
    FwdItem cur = decode(pair, 0);
    {
        float4 bv0[NT][4];
        load_bias(cur.co0, bv0);
        init_acc(bv0);
    }
    // prologue: the whole halo of the first item and its first filter slab
    if constexpr (FASTH_CT) {
        unsigned tm0;
        const char* const sb0 = fast_base(cur, tm0);
#pragma unroll
        for (int ph = 0; ph < 9; ++ph) {
            const int pk = h_pack[ph];
            hoff[ph] = (unsigned)((((pk & 15) * H + ((pk >> 4) & 15)) * W + ((pk >> 8) & 31)) * s.C0 + ((pk >> 13) & 3) * 8) * 2u;
            issue_halo_fast(ph, 0, sb0, tm0);
        }
    } else {
#pragma unroll
        for (int ph = 0; ph < 9; ++ph) {
            hp[ph] = halo_src(cur, h_pack[ph]);
            issue_halo(ph, 0);
        }
    }
    issue_filter(cur, 0, 0);
    int g = 0, hb = 0;
#ifdef FMRI_PROF
    unsigned long long prof[12] = {};
    PROF_T(tk0);
#endif
    while (true) {
        // the item after `cur` in this workgroup's stream
        bool has_next = true;
        FwdItem nxt = cur;
        int npair = pair;
        if (cur.ch + 1 < nch) nxt.ch = cur.ch + 1;
        else {
            npair = pair + gridDim.x;
            has_next = npair < npairs;
            if (has_next) nxt = decode(npair, 0);
        }
        // new tile, first chunk of the second source, or (up-backward) first chunk of the next parity class
        const bool fresh = MODE == 2 ? (nxt.ch % kpc == 0) : (nxt.ch == 0 || (nxt.ch << 5) == s.C0);
        unsigned ntm = 0;
        const char* nsb = nullptr;
        if constexpr (FASTH_CT) {
            if (has_next) nsb = fast_base(nxt, ntm);
        }
        // CHEAP_FRESH (the parity modes, i.e. the decoder 'a' layers of the 2-D path, which run on this kernel): pointers kept and advanced as
        // before, a FRESH pointer formed as base of the box corner + lane constant, zero page where an edge row leaves the volume (see
        // k_conv_fwd_ws) - every wave of this kernel issues halo pieces, so halo_src's ~50 instructions per piece sat in front of all MFMAs
        constexpr bool CHEAP_FRESH = PAR && !CUBE;
        if constexpr (CHEAP_FRESH) {
            if (has_next && fresh) {
                const FwdItem& it = nxt;
                ntm = __builtin_amdgcn_readfirstlane((it.d0 == 0 ? 1u : 0u) | (it.d0 + TD == D ? 2u : 0u) | (it.h0 == 0 ? 4u : 0u) | (it.h0 + TH == H ? 8u : 0u) |
                                                     (it.w0 == 0 ? 16u : 0u) | (it.w0 + TW == W ? 32u : 0u));
                int64_t eoff;
                if constexpr (MODE == 2) {
                    const int p = it.ch / kpc, coff = (it.ch % kpc) << 5;
                    const int64_t dd = PL ? ((int64_t)it.n * D + it.d0 - 1) : ((int64_t)it.n * 2 * D + 2 * (it.d0 - 1) + (p >> 2));
                    eoff = ((dd * 2 * H + 2 * (it.h0 - 1) + ((p >> 1) & 1)) * 2 * W + 2 * (it.w0 - 1) + (p & 1)) * s.C0 + coff;
                } else {
                    eoff = ((((int64_t)it.n * D + it.d0 - 1) * H + it.h0 - 1) * W + it.w0 - 1) * s.C0 + (it.ch << 5);
                }
                const unsigned long long a = reinterpret_cast<unsigned long long>(s.p0 + eoff);
                const unsigned alo = __builtin_amdgcn_readfirstlane((unsigned)a), ahi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
                nsb = reinterpret_cast<const char*>(((unsigned long long)ahi << 32) | alo);
            }
        }
        auto cheap_off = [&](int pk) -> unsigned {
            const int hd_ = pk & 15, hh_ = (pk >> 4) & 15, hw_ = (pk >> 8) & 31, ls = (pk >> 13) & 3;
            if constexpr (MODE == 2) return (unsigned)((((PL ? hd_ : 2 * hd_) * (2 * H) + 2 * hh_) * (2 * W) + 2 * hw_) * s.C0 + ls * 8) * 2u;
            else return (unsigned)(((hd_ * H + hh_) * W + hw_) * s.C0 + ls * 8) * 2u;
        };
        const unsigned char* const lh = lds + hb * HALO_BYTES;
        // keep the 54 per-(phase,tap) fragment addresses out of long-lived registers: recomputing them costs a few VALU
        // instructions per MFMA, which issue in the MFMA's shadow, whereas hoisting them spills
        asm volatile("" : "+v"(hv0[0]), "+v"(hv0[1]));
#pragma unroll
        for (int pl = 0; pl < NPH; ++pl, ++g) {
            PROF_T(t0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's DMA of the previous phase has landed
            PROF_T(t1);
            __builtin_amdgcn_s_barrier();                          // ... and everybody else's; previous phase fully read
            PROF_T(t2);
            // DMA of the next phase (an LDS-DMA instruction costs its wave 100-180 cycles of issue here wherever it is placed:
            // staggering the two waves of a SIMD, or issuing mid-phase, measured the same or slower - tools/prof_phases.py)
            // the chunk's 9 halo pieces spread over its phases: 1 each (9 phases), 3 each (3), 3+2+2+2 (4), 5+4 (2)
            const int HP0 = NPH == 4 ? (pl == 0 ? 0 : 1 + 2 * pl) : (NPH == 2 ? 5 * pl : pl * (9 / NPH));
            const int HPN = NPH == 4 ? (pl == 0 ? 3 : 2) : (NPH == 2 ? 5 - pl : 9 / NPH);
            auto issue_dma = [&]() {
                if (pl < NPH - 1) issue_filter(cur, pl + 1, (g + 1) & 1);
                else if (has_next) issue_filter(nxt, 0, (g + 1) & 1);
                if (has_next) {
#pragma unroll
                    for (int q = 0; q < 5; ++q) {
                        if (q < HPN) {
                            if constexpr (FASTH_CT) issue_halo_fast(HP0 + q, hb ^ 1, nsb, ntm);
                            else {
                                if (fresh) {
                                    if constexpr (CHEAP_FRESH) {
                                        const int pk = h_pack[HP0 + q];
                                        const bool bad = !((pk >> 15) & 1) || (((unsigned)pk >> 16) & ntm) != 0;
                                        hp[HP0 + q] = bad ? (const bf16_t*)zpage : reinterpret_cast<const bf16_t*>(nsb + cheap_off(pk));
                                    } else hp[HP0 + q] = halo_src(nxt, h_pack[HP0 + q]);
                                } else hp[HP0 + q] += 32;
                                issue_halo(HP0 + q, hb ^ 1);
                            }
                        }
                    }
                }
            };
            issue_dma();
            PROF_T(t3);
            const unsigned char* const lf = lds + 2 * HALO_BYTES + (g & 1) * FILT_BYTES;
            // halo row offset of this phase's (kd,kh) and of its kw taps.  Up modes: phase = (kd',kh') in {0,1}^2 shifted by the
            // parity of the output class (up-forward: offsets {-1,0} / {0,+1} for p = 0 / 1) or of the chunk (up-backward: mirrored)
            int hoff, kw0 = 0;
            if constexpr (!PAR) hoff = (((PH0 + pl) / 3) * HH + ((PH0 + pl) % 3)) * HW;
            else {
                const int p = MODE == 1 ? cur.par : (NPAR - 1) - cur.ch / kpc;
                if constexpr (PL) hoff = (HH + pl + ((p >> 1) & 1)) * HW;           // centre plane; phase = kh'
                else hoff = (((pl >> 1) + (p >> 2)) * HH + ((pl & 1) + ((p >> 1) & 1))) * HW;
                kw0 = p & 1;
            }
#pragma unroll
            for (int kw = 0; kw < NKW; ++kw) {
                int hb0[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) hb0[j] = swz64(hv0[j] + hoff + kw + kw0, hk);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8_t a[NT], b[2];
#pragma unroll
                    for (int c = 0; c < NT; ++c)
                        a[c] = *reinterpret_cast<const bf16x8_t*>(lf + fa[ks] + (kw * BN + c * 32) * 64);
#pragma unroll
                    for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(lh + (hb0[j] ^ (ks << 5)));
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int c = 0; c < NT; ++c)
                            acc[j][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[c], b[j], acc[j][c], 0, 0, 0);
                }
            }
            PROF_T(t4);
            PROF_ADD(0, t0, t1); PROF_ADD(1, t1, t2); PROF_ADD(2, t2, t3); PROF_ADD(3, t3, t4);
#ifdef FMRI_PROF
            prof[6] += 1;
#endif
        }
        if (RES && cur.ch == nch - 1) {
            // ---- epilogue with a residual: y = act(acc + residual + bias).  The sum has to be formed in fp32 before the activation,
            // so the raw accumulators are transposed through LDS as fp32, one 32-voxel column tile at a time (wave-private 8 KiB of
            // the consumed halo slot), and finished line-major: 8 (4) lanes per voxel, 8 channels = one 16-byte bf16 piece each.
            constexpr int PPV = BN / 4;                  // 16-byte fp32 pieces per voxel
            constexpr int LPV = BN / 8;                  // lanes per voxel
            constexpr int VPI = 64 / LPV;                // voxels per instruction
            float4 bvn[NT][4];
            load_bias(has_next ? nxt.co0 : cur.co0, bvn);
            const int64_t org = ((((int64_t)cur.n * D + cur.d0) * H + cur.h0) * W + cur.w0) * Cout + cur.co0;
            // the residual's lines of this wave's two column tiles, requested in front of the barrier and the fp32 transposition (round 6: they
            // were loaded inside the store loop, one exposed memory latency per column tile - see the ReLU-mask lines of the plain epilogue)
            uint4 rq[2][32 / VPI];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int kk = 0; kk < 32 / VPI; ++kk) {
                    const int rt = 2 * wv + j, rr = kk * VPI + lane / LPV, q8 = lane % LPV;
                    rq[j][kk] = *reinterpret_cast<const uint4*>(residual + org + ((tile_d(rt) * H + tile_h(rt, rr)) * W + lane_w(rr)) * Cout + q8 * 8);
                }
            __builtin_amdgcn_s_barrier();
            unsigned char* const stage = lds + hb * HALO_BYTES + wv * (32 * BN * 4);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int c = 0; c < NT; ++c)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int q = c * 8 + 2 * gq + hk;
                        *reinterpret_cast<float4*>(stage + r * (BN * 4) + (((q ^ r) & (PPV - 1)) << 4)) =
                            make_float4(acc[j][c][4 * gq], acc[j][c][4 * gq + 1], acc[j][c][4 * gq + 2], acc[j][c][4 * gq + 3]);
                    }
                const int rt = 2 * wv + j;
#pragma unroll
                for (int kk = 0; kk < 32 / VPI; ++kk) {
                    const int rr = kk * VPI + lane / LPV, q8 = lane % LPV;
                    const float4 a0 = *reinterpret_cast<const float4*>(stage + rr * (BN * 4) + ((((2 * q8) ^ rr) & (PPV - 1)) << 4));
                    const float4 a1 = *reinterpret_cast<const float4*>(stage + rr * (BN * 4) + ((((2 * q8 + 1) ^ rr) & (PPV - 1)) << 4));
                    const int64_t ao = org + ((tile_d(rt) * H + tile_h(rt, rr)) * W + lane_w(rr)) * Cout + q8 * 8;
                    const uint4 r4 = rq[j][kk];
                    // (the partial sum is rounded to bf16 before the residual is added - what the asynchronous form, EPI 3 of k_conv_fwd_ws,
                    // stages: every form of this launch gives the same bits, whichever one the grid size selects)
                    const unsigned p0 = pack2bf(a0.x, a0.y), p1 = pack2bf(a0.z, a0.w), p2 = pack2bf(a1.x, a1.y), p3 = pack2bf(a1.z, a1.w);
                    float o[8] = {__uint_as_float(p0 << 16) + __uint_as_float(r4.x << 16), __uint_as_float(p0 & 0xffff0000u) + __uint_as_float(r4.x & 0xffff0000u),
                                  __uint_as_float(p1 << 16) + __uint_as_float(r4.y << 16), __uint_as_float(p1 & 0xffff0000u) + __uint_as_float(r4.y & 0xffff0000u),
                                  __uint_as_float(p2 << 16) + __uint_as_float(r4.z << 16), __uint_as_float(p2 & 0xffff0000u) + __uint_as_float(r4.z & 0xffff0000u),
                                  __uint_as_float(p3 << 16) + __uint_as_float(r4.w << 16), __uint_as_float(p3 & 0xffff0000u) + __uint_as_float(r4.w & 0xffff0000u)};
#pragma unroll
                    for (int i = 0; i < 8; ++i) o[i] = vmax(o[i], __builtin_fmaf(o[i], act_s, 0.f));
                    *reinterpret_cast<uint4*>(y + ao) = make_uint4(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7]));
                }
            }
            init_acc(bvn);
        }
        if (!RES && cur.ch == nch - 1) {
            // ---- epilogue.  D rows = output channel (reg&3)+8*(reg>>2)+4*hk, D cols = voxel r: a lane owns 4-channel pieces of ONE
            // voxel, and storing those directly touches 64 cache lines per instruction (measured: 35 % of the kernel at Cin = 32,
            // store-issue bound).  Instead the wave's 64 voxels x BN channels go through a wave-private 8 KiB of the halo slot that
            // was just consumed (free until the next phase's DMA; the barrier below retires its readers) and come back
            // line-major: 8 (4) lanes write one voxel's 128 (64) contiguous bytes.
            PROF_T(te0);
            constexpr int CPV = BN / 8;                  // 16-byte pieces per voxel
            constexpr int VPI = 64 / CPV;                // voxels per store instruction
            constexpr int SWM = NT == 2 ? 7 : 3;         // piece swizzle: row v keeps piece q at slot q ^ sw(v)
            float4 bvn[NT][4];                           // bias of the next tile (lands while this one is packed)
            load_bias(has_next ? nxt.co0 : cur.co0, bvn);
            // address in y (and in the mask) of the 16-byte piece this lane stores in iteration kk: wave-uniform 64-bit tile origin + 32-bit
            // offset inside the tile's bounding box
            auto out_off = [&](int kk) -> int64_t {
                const int v = kk * VPI + lane / CPV, q = lane % CPV;
                const int rt = 2 * wv + (v >> 5), rr = v & 31;
                if constexpr (MODE == 1 && PL) {   // parity class (ph, pw) of the [D][2H][2W] output
                    const int64_t org = ((((int64_t)cur.n * D + cur.d0) * 2 * H + 2 * cur.h0 + ((cur.par >> 1) & 1)) * 2 * W + 2 * cur.w0 + (cur.par & 1)) * Cout +
                                        cur.co0;
                    return org + ((tile_d(rt) * 2 * H + 2 * tile_h(rt, rr)) * 2 * W + 2 * lane_w(rr)) * Cout + q * 8;
                } else if constexpr (MODE == 1) {   // parity class p of the [2D][2H][2W] output
                    const int64_t org = ((((int64_t)cur.n * 2 * D + 2 * cur.d0 + (cur.par >> 2)) * 2 * H + 2 * cur.h0 + ((cur.par >> 1) & 1)) * 2 * W +
                                         2 * cur.w0 + (cur.par & 1)) * Cout + cur.co0;
                    return org + ((2 * tile_d(rt) * 2 * H + 2 * tile_h(rt, rr)) * 2 * W + 2 * lane_w(rr)) * Cout + q * 8;
                } else {
                    const int64_t org = ((((int64_t)cur.n * D + cur.d0) * H + cur.h0) * W + cur.w0) * Cout + cur.co0;
                    return org + ((tile_d(rt) * H + tile_h(rt, rr)) * W + lane_w(rr)) * Cout + q * 8;
                }
            };
            // the ReLU mask lines of this wave's stores are requested HERE, in front of the barrier and the packing of the tile (round 6: they
            // were loaded inside the store loop, their latency exposed once per tile - the input-gradient launches of the 2-D path, which stay on
            // this kernel, ran 25-35 % behind the forward launches of the same shape: profiles/r06_cfg3_per_layer.json)
            uint4 mk4[CPV];
            if (mask) {
#pragma unroll
                for (int kk = 0; kk < CPV; ++kk) mk4[kk] = *reinterpret_cast<const uint4*>(mask + out_off(kk));
            }
            __builtin_amdgcn_s_barrier();
            PROF_T(te2);
            unsigned char* const stage = lds + hb * HALO_BYTES + wv * (64 * BN * 2);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int v = j * 32 + r;
                const int vs = NT == 2 ? (v & 7) : ((v >> 2) & 3);
#pragma unroll
                for (int c = 0; c < NT; ++c) {
#pragma unroll
                    for (int pq = 0; pq < 2; ++pq) {
                        unsigned pk[2][2];                          // [group 2pq, 2pq+1][dword]: this lane's 4 channels, bf16
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int gq = 2 * pq + u;
                            float o[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float vv = acc[j][c][4 * gq + i];
                                o[i] = vmax(vv, __builtin_fmaf(vv, act_s, 0.f));
                            }
                            pk[u][0] = pack2bf(o[0], o[1]);
                            pk[u][1] = pack2bf(o[2], o[3]);
                        }
                        // half-wave exchange: lanes 0-31 end up with channels 16pq..16pq+7 of their voxel, lanes 32-63 with +8..+15
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            auto sw = __builtin_amdgcn_permlane32_swap(pk[0][q], pk[1][q], false, false);
                            pk[0][q] = sw[0];
                            pk[1][q] = sw[1];
                        }
                        const int q = c * 4 + pq * 2 + hk;
                        *reinterpret_cast<uint4*>(stage + v * (BN * 2) + (((q ^ vs) & SWM) << 4)) =
                            make_uint4(pk[0][0], pk[0][1], pk[1][0], pk[1][1]);
                    }
                }
            }
            init_acc(bvn);
            PROF_T(te3);
            PROF_ADD(7, te0, te2); PROF_ADD(8, te2, te3);
#pragma unroll
            for (int kk = 0; kk < CPV; ++kk) {
                const int v = kk * VPI + lane / CPV, q = lane % CPV;
                const int vs = NT == 2 ? (v & 7) : ((v >> 2) & 3);
                uint4 o4 = *reinterpret_cast<const uint4*>(stage + v * (BN * 2) + (((q ^ vs) & SWM) << 4));
                const int64_t ao = out_off(kk);
                if (mask) {
                    const unsigned mm[4] = {mk4[kk].x, mk4[kk].y, mk4[kk].z, mk4[kk].w};
                    unsigned* const oo = reinterpret_cast<unsigned*>(&o4);
                    unsigned z2 = 0u, o2 = 0x00010001u;
                    asm volatile("" : "+v"(z2), "+v"(o2));
#pragma unroll
                    for (int i = 0; i < 4; ++i) oo[i] &= pos_mask2(mm[i], z2, o2);        // three packed-integer instructions per dword (was eight: see k_conv_fwd_ws)
                }
                *reinterpret_cast<uint4*>(y + ao) = o4;
                if constexpr (TAIL) {
                    if (tail.logits) {
                        // final 1x1 conv to one label: the CPV lanes of a voxel hold its BN (= Cout) channels, 8 each (the form of k_conv_fwd_ws)
                        float part = __uint_as_float(o4.x << 16) * w1a.x;
                        part = __builtin_fmaf(__uint_as_float(o4.x & 0xffff0000u), w1a.y, part);
                        part = __builtin_fmaf(__uint_as_float(o4.y << 16), w1a.z, part);
                        part = __builtin_fmaf(__uint_as_float(o4.y & 0xffff0000u), w1a.w, part);
                        part = __builtin_fmaf(__uint_as_float(o4.z << 16), w1b.x, part);
                        part = __builtin_fmaf(__uint_as_float(o4.z & 0xffff0000u), w1b.y, part);
                        part = __builtin_fmaf(__uint_as_float(o4.w << 16), w1b.z, part);
                        part = __builtin_fmaf(__uint_as_float(o4.w & 0xffff0000u), w1b.w, part);
                        // sum over the voxel's CPV adjacent lanes with DPP moves (quad swaps, then the other quad of the half row) - the
                        // same pairs in the same order as __shfl_xor 1, 2, 4, whose three dependent ds_bpermute round trips per store
                        // iteration cost this launch 47 us on the MFMA waves (isolated, dec0b of configs[3]); measured after: see DESIGN 6.4
                        part += dpp_mov<0xB1>(part);               // quad_perm [1, 0, 3, 2]
                        part += dpp_mov<0x4E>(part);               // quad_perm [2, 3, 0, 1]
                        if constexpr (CPV == 8) part += dpp_mov<0x141>(part);      // row_half_mirror: lane i <- lane 7 - i of its 8
                        const int rt = 2 * wv + (v >> 5), rr = v & 31;
                        if (q == 0)
                            tail.logits[(((int64_t)cur.n * D + cur.d0 + tile_d(rt)) * H + cur.h0 + tile_h(rt, rr)) * W + cur.w0 + lane_w(rr)] = part + b1v;
                    }
                }
            }
            if constexpr (TAIL) {
                if (tail.pool) {
                    // 2x2 max pooling of the wave's own 64 staged voxels: a column tile is two h-rows x 16 w of one slice = 8 windows, so the wave
                    // owns 16 pooled voxels x CPV 16-byte pieces.  After ReLU every value is >= +0 and the bf16 bit patterns order like
                    // unsigned integers: four packed-integer maxima per source piece; any other activation goes through fp32.
                    constexpr int NPP = (16 * CPV + 63) / 64;
#pragma unroll
                    for (int pi = 0; pi < NPP; ++pi) {
                        const int idx = pi * 64 + lane;
                        const bool live = idx < 16 * CPV;
                        const int pv = (live ? idx : 0) / CPV, q = idx % CPV, j = pv >> 3, pw = pv & 7;
                        uint4 src[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const int w_ = 2 * pw + (c & 1);
                            const int v = j * 32 + ((c >> 1) ? 16 + ((w_ + (HW & 15)) & 15) : w_);       // inverse of lane_w: see the row rotation above
                            const int vs = NT == 2 ? (v & 7) : ((v >> 2) & 3);
                            src[c] = *reinterpret_cast<const uint4*>(stage + v * (BN * 2) + (((q ^ vs) & SWM) << 4));
                        }
                        uint4 o;
                        if (act_s == 0.f) {
                            unsigned* const oo = reinterpret_cast<unsigned*>(&o);
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                unsigned a = reinterpret_cast<const unsigned*>(&src[0])[i], b = reinterpret_cast<const unsigned*>(&src[1])[i];
                                unsigned c2 = reinterpret_cast<const unsigned*>(&src[2])[i], d2 = reinterpret_cast<const unsigned*>(&src[3])[i];
                                asm("v_pk_max_u16 %0, %0, %1" : "+v"(a) : "v"(b));
                                asm("v_pk_max_u16 %0, %0, %1" : "+v"(c2) : "v"(d2));
                                asm("v_pk_max_u16 %0, %0, %1" : "+v"(a) : "v"(c2));
                                oo[i] = a;
                            }
                        } else {
                            float mx[8];
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const unsigned pp[4] = {src[c].x, src[c].y, src[c].z, src[c].w};
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const float lo = __uint_as_float(pp[i] << 16), hi = __uint_as_float(pp[i] & 0xffff0000u);
                                    mx[2 * i] = c == 0 ? lo : vmax(mx[2 * i], lo);
                                    mx[2 * i + 1] = c == 0 ? hi : vmax(mx[2 * i + 1], hi);
                                }
                            }
                            o.x = (__float_as_uint(mx[0]) >> 16) | (__float_as_uint(mx[1]) & 0xffff0000u);
                            o.y = (__float_as_uint(mx[2]) >> 16) | (__float_as_uint(mx[3]) & 0xffff0000u);
                            o.z = (__float_as_uint(mx[4]) >> 16) | (__float_as_uint(mx[5]) & 0xffff0000u);
                            o.w = (__float_as_uint(mx[6]) >> 16) | (__float_as_uint(mx[7]) & 0xffff0000u);
                        }
                        const int rt = 2 * wv + j;
                        const int64_t po = ((((int64_t)cur.n * D + cur.d0 + tile_d(rt)) * (H >> 1) + (cur.h0 >> 1) + (rt & 3)) * (W >> 1) + (cur.w0 >> 1) + pw) * Cout +
                                           cur.co0 + q * 8;
                        if (live) *reinterpret_cast<uint4*>(tail.pool + po) = o;
                    }
                }
            }
            PROF_T(te1);
            PROF_ADD(4, te0, te1);
        }
        if (!has_next) break;
        cur = nxt;
        pair = npair;
        hb ^= 1;
    }
#ifdef FMRI_PROF
    PROF_T(tk1);
    prof[5] = tk1 - tk0;
    if (lane == 0)
        for (int i = 0; i < 12; ++i) atomicAdd(&g_prof[i], prof[i]);
#endif
}

// ============================================================================================ forward / dgrad, warp-specialised
// Same tiling, LDS rings and phase protocol as k_conv_fwd_mfma, with the eight waves split into two roles:
//   waves 0-3 (consumers, one per SIMD): nothing but fragment reads + MFMAs + the epilogue.  Each owns FOUR 32-voxel column tiles (one
//              d-plane of the 4 x 8 x 16 tile) x NT Cout tiles, so a (kw, k-step) needs NT + 4 ds_read_b128 for 4*NT MFMAs: 0.75 KB of LDS
//              reads per MFMA instead of 1 KB.  At 1 KB per MFMA the four SIMDs ask for exactly the LDS peak (4 MFMAs x 1 KB per 32 cycles =
//              128 B/clk), i.e. the 2 x 2 kernel is LDS-bound and MFMA-bound at once and every LDS-DMA write or bank conflict costs time;
//   waves 4-7 (producers, one per SIMD): all LDS-DMA (filter slabs, halo pieces), its address arithmetic and the counted waits.  Their issue
//              stalls (100-180 cycles per instruction) no longer sit in front of the MFMAs of the same wave.
// One s_barrier per phase hands both rings over exactly as before (a producer waits for its own DMA before it arrives).
//
// ASYNC (round 3; 3-D launches without a residual): the tile's stores leave the MFMA waves' critical path.  The consumers only stage the
// finished tile as bf16 in the consumed halo slot and go straight on to the next tile; the PRODUCERS read it back and store it (ReLU mask,
// pooled copy and final 1x1x1 logits included) during phase 0 of the next tile, under the consumers' MFMAs.  The staged slot is the one the
// next item's halo pieces go to, so those start one phase later (phases 1 .. NPH-2 instead of 0 .. NPH-2): every producer's reads of the
// stage are in registers before it arrives at the phase-1 barrier, behind which the first piece is issued.  The mask lines of a tile are
// requested by the producers during the tile's last phase (that phase carries no halo pieces) and have landed by the full wait of the next
// item's phase 0.  The "staged" barrier IS the next tile's phase-0 barrier: one barrier less per tile.  On the level-0 layers of
// BASELINE configs[1] (1-2 chunks per tile, 64 KB of stores + 64 KB of mask per 18-36 us tile) the epilogue was 13-25 % of a consumer's time.
// EPI (ASYNC instantiations): what the epilogue does besides storing the tile - 0 nothing or the pooled copy, 1 the ReLU mask of an input
// gradient, 2 the logits of the final 1x1x1 conv; -1 = decided at run time (all of it compiled in).  One instantiation per case keeps the
// producers' registers apart: the 16 prefetched mask lines (64 registers) and the 9 logit weights are only allocated where they are used -
// with everything in one kernel the producers' path spilled 18 registers and the pooled-copy launch (enc0b forward, bound by its producers)
// ran 5 % slower for registers only the other cases need.
// S16 (round 5): the MFMA waves issue v_mfma_f32_16x16x32_bf16 instead of 32x32x16 - the SAME output tile per wave (one d-plane of the tile:
// 128 voxels x BN channels, 128 accumulator registers) cut into 8 h-rows of 16 voxels x BN / 16 channel fragments, the whole 32-channel chunk
// as ONE k-step.  LDS bytes per MAC are the same ((M + N) * K); what changes is the energy per MAC: on live data the chip is power-limited
//  and holds a higher clock on this shape (MI355X guide, 'DVFS give-back' item 7).  A 16-lane group of a fragment read is one
// h-row of the halo at one 16-byte slot - the access pattern the column-keyed swizzle was derived for - and no row rotation is needed.
// F32 (round 6, the parity mode on the benchmarked kernel structure): fp32 tensors and filters on v_mfma_f32_32x32x2_f32 (exact fp32: the
// guide's chip table - an fmaf chain per output).  A 64-byte halo / filter row is then 16 fp32 channels instead of 32 bf16 ones and everything
// that moves or addresses BYTES is unchanged: the launcher passes every channel count that is a memory stride in units of 2 bytes (s.C0, s.C1 =
// twice the real counts; `Cout` stays the real count of filter rows, CoutB = 2 Cout is the voxel stride of y / mask / residual / pool), so halo
// box, LDS-DMA pieces, swizzle, rings, phases, barriers, the asynchronous drain and its tails are the instructions of the bf16 launch.  What
// differs: a 16-byte fragment is 4 k-values per lane instead of 8 - four MFMAs of k = 2 in place of one of k = 16 - the staged tile holds fp32
// (NT = 1: 32 channels x 4 B = the 128-byte voxel line of the 64-wide bf16 block), and the element-wise tails (ReLU mask, residual, pool)
// work on floats.
template <int NT, bool PL, int MODE, bool RES, bool ASYNC = false, int EPI = -1, bool FH = false, bool S16 = false, bool F32 = false>   // FH: fast halo addressing (producers, below)
__global__ void __launch_bounds__(fw::NTHREADS)
k_conv_fwd_ws(SrcB s, const bf16_t* __restrict__ wt, const float* __restrict__ bias, const bf16_t* __restrict__ mask,
              const bf16_t* residual, bf16_t* y, int N, int D, int H, int W, int Cout, int act, float alpha, FwdTail tail) {
    constexpr int TD = fw::TD, TH = fw::TH, TW = fw::TW;
    static_assert(!(RES && MODE != 0), "unsupported combination");
    static_assert(!F32 || (NT == 1 && !PL && !RES && !S16 && EPI < 4 && EPI != 2), "fp32 form: 32-wide blocks, 3-D, asynchronous residual, no logits / normalisation tails");
    constexpr bool PAR = MODE != 0;
    // TIGHT (round 3; the 3-D parity modes): a parity class reads, per axis, only the low-res voxels {g-1, g} or {g, g+1} - a
    // (TD+1) x (TH+1) x (TW+1) box whose origin depends on the parity, not the 6 x 10 x 18 box of the 3-tap conv: 765 instead of 1080 rows,
    // 48 instead of 68 LDS-DMA instructions per chunk.  These launches are bound by their producers' DMA issue (7.7 instructions per wave
    // and 32-MFMA phase): the matrix pipe was 0.38-0.48 busy on them (profiles/r03_pmc_mfma.json).
    constexpr bool TIGHT = PAR && !PL;
    constexpr int HD = TIGHT ? TD + 1 : TD + 2, HH = TIGHT ? TH + 1 : TH + 2, HW = TIGHT ? TW + 1 : TW + 2, HVOX = HD * HH * HW;
    constexpr int H_INSTR = (HVOX * 4 + 63) / 64, HALO_BYTES = H_INSTR * 1024;
    // the two halo slots: back to back, or - TIGHT, where a slot (48 KiB) is smaller than the staged tile (64 KiB) - with a 16 KiB gap between
    // them that the stage of either slot extends into: stage(slot) = [slot * HALO_BYTES, + 64 KiB) covers slot 0 + gap, resp. gap + slot 1
    constexpr int STAGE_BYTES = 4 * 32 * 4 * 64 * 2 > 4 * 32 * 64 * 4 ? 4 * 32 * 4 * 64 * 2 : 4 * 32 * 64 * 4;
    constexpr int HALO_STRIDE = TIGHT ? STAGE_BYTES : HALO_BYTES;
    constexpr int HALO_SPAN = HALO_STRIDE + HALO_BYTES;          // bytes of the halo / stage area
    constexpr int BN = 32 * NT;
    constexpr int BNB = F32 ? 2 * BN : BN;           // the block's width in 2-byte units: bytes of a staged / stored voxel line = 2 * BNB
    constexpr int NKW = PAR ? 2 : 3;
    // TIGHT also has LDS to spare (2 x 48 + 16 KiB of halo / stage): a phase covers RPP = 2 (kd', kh') filter rows - 64 instead of 32 MFMAs
    // per wave and barrier, 2 phases per chunk - with 16 KiB filter slots (BN = 64)
    constexpr int RPP = TIGHT ? 2 : 1;                   // (kd, kh) filter rows per phase
    constexpr int FILT_BYTES = (RPP * NKW > 3 ? RPP * NKW : 3) * BN * 64;
    constexpr int F_INSTR = RPP * NKW * BN * 64 / 1024;  // 12 / 6 (8 / 4 in the planar up modes, 16 / 8 in the 3-D ones)
    constexpr int DW = 4;                                // DMA (producer) waves = MFMA (consumer) waves
    constexpr int JT = 4;                                // column tiles per consumer wave
    constexpr int F_PER_WAVE = (F_INSTR + DW - 1) / DW;  // 3 / 2 (2 / 1)
    // halo DMA instructions per producer wave and chunk.  Planar (2-D slices): halo planes 0 and 5 are never read, only the instructions
    // 11..56 that touch planes 1-4 are issued (46 -> 12 per wave, the last two are zero copies so that every wave issues the same number)
    constexpr int H_I0 = PL ? 11 : 0, H_I1 = PL ? 57 : H_INSTR;
    constexpr int NPIECE = (H_I1 - H_I0 + DW - 1) / DW;  // 17 (12)
    constexpr int NPAR = PL ? 4 : 8;
    constexpr int NROW = PAR ? (PL ? 2 : 4) : (PL ? 3 : 9);       // (kd, kh) filter rows per chunk
    constexpr int NPH = NROW / RPP;                                // phases per chunk
    static_assert(NPH * RPP == NROW, "rows per phase");
    constexpr int PH0 = PL ? 3 : 0;
    constexpr bool ASY = ASYNC && !RES && !PL && MODE == 0;
    constexpr bool HAS_MASK = EPI < 0 || EPI == 1, HAS_LOGITS = !F32 && (EPI < 0 || EPI == 2), HAS_POOL = EPI <= 0;
    // EPI 3 (ASYNC only): y = act(staged + residual) - the skip launch of the parity form.  The MFMA waves stage bf16(acc + bias) WITHOUT the
    // activation; the producers add the residual lines (prefetched like the mask lines) in fp32, activate, round and store.  One bf16 rounding
    // more than the RES epilogue (which adds the residual to the fp32 accumulators): both partial sums of the parity form - the up-sampled
    // channels' (stored as bf16 by the MODE 1 launch) and now the skip channels' - are rounded before they meet.
    constexpr bool HAS_RESID = EPI == 3 || EPI == 6;
    // EPI 4 / 6 (ASYNC only): the producers also sum the values they store and their squares per channel - the statistics of the
    // normalisation layer behind this conv (reference create_convolution_block, unet.py:103-115), which then needs no pass of its own over
    // the tensor.  EPI 5: the launch is the input gradient in front of a normalised block: the producers turn dy into dz = dy * act'(z)
    // (z recomputed from the block's conv output x, whose lines they prefetch like mask lines) and sum dz and dz * x per channel - the
    // reductions of the normalisation's backward pass.  Per lane 8 + 8 fp32 partial sums (a lane keeps its 8 channels for the whole
    // kernel), reduced across the lanes of a wave and added to the fp64 accumulators when the (group, channel block) changes.
    constexpr bool HAS_STATS = EPI == 4 || EPI == 6;
    constexpr bool HAS_NBWD = EPI == 5;
    constexpr bool HAS_SUMS = HAS_STATS || HAS_NBWD;
    constexpr bool HAS_LINES = HAS_MASK || HAS_RESID || HAS_NBWD;      // per-store 16-byte lines the producers prefetch into mk[]
    // EPI 5: the lines are prefetched one PART of the drain ahead (two buffers of a part's lines) instead of the whole tile's at once - the
    // partial sums and the scale / shift need the registers (64 -> 32 for the lines at BN = 64)
    constexpr bool LINES_PIPE = HAS_NBWD;
    // the producers drain a staged tile in DP parts, one per phase, in phases 0 .. DP-1 of the next tile's first item (one part fits a
    // phase beside the filter slab's DMA; the whole drain in phase 0 made the producers late for the phase-1 barrier: measured 8 % SLOWER
    // than the synchronous epilogue); the halo pieces of that item's successor follow in phases DP .. NPH-2
    constexpr int DP = ASY ? 4 : 0;
    static_assert(!ASY || NPH >= DP + 3, "asynchronous epilogue: phases DP .. NPH-2 carry the halo pieces");
    static_assert(4 * 32 * JT * BNB * 2 <= (TIGHT ? STAGE_BYTES : HALO_BYTES) && 4 * 32 * BN * 4 <= (TIGHT ? STAGE_BYTES : HALO_BYTES),
                  "epilogue staging must fit the consumed halo slot (+ gap)");
    static_assert(!TIGHT || HALO_BYTES + STAGE_BYTES - HALO_BYTES <= HALO_STRIDE, "stage of slot 0 must end where slot 1 begins");
    __shared__ __attribute__((aligned(16))) unsigned char lds[HALO_SPAN + 2 * FILT_BYTES];
    const unsigned char* const zpage = zero_page_addr();

    const int Cin = s.C0 + s.C1;
    const int kpc = MODE == 2 ? (s.C0 >> 5) : 1;
    const int nch = MODE == 2 ? NPAR * kpc : (Cin >> 5);
    const int Krow = MODE == 2 ? s.C0 : Cin;
    const int cbn = Cout / BN;
    const int ncb = MODE == 1 ? NPAR * cbn : cbn;
    const int twn = W / TW, thn = H / TH, tdn = D / TD;
    const int npairs = N * tdn * thn * twn * ncb;
    const int CoutB = F32 ? 2 * Cout : Cout;         // voxel stride of y / mask / residual / pool in 2-byte units

    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 31, hk = lane >> 5;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(lds));
    const unsigned ldsf0 = lds0 + HALO_SPAN;

    auto decode = [&](int pair, int ch) {
        FwdItem it;
        it.ch = ch;
        const int cb = pair % ncb;
        it.par = MODE == 1 ? cb / cbn : 0;
        it.co0 = (MODE == 1 ? cb % cbn : cb) * BN;
        int q = pair / ncb;
        // tiles are numbered in compact 2 (d) x 4 (h) x 4 (w) blocks when the grid allows it: the ~32 consecutive pairs an XCD works on at any
        // time then share their halo planes / rows / columns in its L2 instead of being a one-tile-thick row of the volume
        if (((twn | thn) & 3) == 0 && (tdn & 1) == 0) {
            const int in = q & 31;
            int blk = q >> 5;
            const int nbw = twn >> 2, nbh = thn >> 2, nbd = tdn >> 1;
            const int bw_ = blk % nbw; blk /= nbw;
            const int bh_ = blk % nbh; blk /= nbh;
            it.w0 = (bw_ * 4 + (in & 3)) * TW;
            it.h0 = (bh_ * 4 + ((in >> 2) & 3)) * TH;
            it.d0 = ((blk % nbd) * 2 + (in >> 4)) * TD;
            it.n = blk / nbd;
            return it;
        }
        it.w0 = (q % twn) * TW; q /= twn;
        it.h0 = (q % thn) * TH; q /= thn;
        it.d0 = (q % tdn) * TD;
        it.n = q / tdn;
        return it;
    };
    int pair = xcd_logical_id(blockIdx.x, gridDim.x);            // neighbouring (tile, Cout block) pairs on one XCD / L2
    if (pair >= npairs) return;
    FwdItem cur = decode(pair, 0);
    int g = 0, hb = 0;

    // Plain epilogue, second half: the tile's 512 voxels x BN channels sit in the consumed halo slot as bf16, voxel-major (written by the
    // consumers); NW waves (this one is number w of them) read them back line-major and store 512 / NW voxels each (8 or 4 lanes per voxel
    // write its 128 or 64 contiguous bytes), with the optional ReLU mask of the producer of the tensor.  NW = 8: all waves, right behind the
    // "staged" barrier (the producers have nothing else to do at that point); NW = 4: the producers alone, under the next tile's MFMAs (ASY).
    auto tile_d = [](int rt) { return rt >> 2; };
    auto tile_h = [](int rt, int rr) { return 2 * (rt & 3) + (rr >> 4); };
    auto lane_w = [](int rr) { return (rr >> 4) ? (((rr & 15) + 16 - (HW & 15)) & 15) : (rr & 15); };
    // Slot swizzle of the halo rows (64 B = four 16-byte slots per voxel): slot ^= key(COLUMN of the row in the halo box).  Rounds 1-3 keyed on
    // the row number ((row >> 2) & 3), which every tap moves: an MFMA wave rebuilt each fragment address with 6 vector instructions (32 per
    // 8-MFMA step, on the SIMD it shares with a producer wave).  Keyed on the column, a tap's (kd, kh) and the column tile only add a constant
    // to the address - the instruction's immediate offset - and the three kw x two k-steps are six registers per lane, set up once: no
    // address arithmetic in the matrix loop at all.  The key tables are 4-colourings of the "same bank group" graph of the four 16-lane
    // groups a ds_read_b128 is served in, for every kw (tests/test_host_lds_swizzle.py re-derives them): conflict-free like the row key.
    constexpr unsigned long long HKEY = HW == 18 ? 0xfa50fa50ull /* (hw >> 1) & 3 */ : 0x267fe640ull /* HW = 17 (TIGHT): 0 0 0 1 2 1 2 3 3 3 3 1 2 1 2 0 0 */;
    static_assert(HW == 18 || HW == 17, "halo width");
    auto halo_key = [](int hw) { return (int)((HKEY >> (2 * hw)) & 3); };
    constexpr int CPV_ = BNB / 8;                    // 16-byte pieces per voxel
    constexpr int VPI_ = 64 / CPV_;                  // voxels per store instruction
    // address in y (and in the mask / residual) of piece q of tile voxel v (column tile v >> 5, lane v & 31):
    // wave-uniform tile origin (64-bit, scalar registers) + 32-bit lane offset - as one 64-bit sum per lane hipcc hoisted a sign-extended
    // offset per store instruction out of the tile loop (32 registers at BN = 64, the first to be spilled)
    auto piece_ptr = [&](const bf16_t* base, const FwdItem& it, int v, int q) -> const char* {
        const int rt = v >> 5, rr = v & 31;
        int64_t org;
        unsigned off;
        if constexpr (MODE == 1 && PL) {
            org = ((((int64_t)it.n * D + it.d0) * 2 * H + 2 * it.h0 + ((it.par >> 1) & 1)) * 2 * W + 2 * it.w0 + (it.par & 1)) * CoutB + (F32 ? 2 : 1) * it.co0;
            off = ((tile_d(rt) * 2 * H + 2 * tile_h(rt, rr)) * 2 * W + 2 * lane_w(rr)) * CoutB + q * 8;
        } else if constexpr (MODE == 1) {
            org = ((((int64_t)it.n * 2 * D + 2 * it.d0 + (it.par >> 2)) * 2 * H + 2 * it.h0 + ((it.par >> 1) & 1)) * 2 * W + 2 * it.w0 + (it.par & 1)) * CoutB + (F32 ? 2 : 1) * it.co0;
            off = ((2 * tile_d(rt) * 2 * H + 2 * tile_h(rt, rr)) * 2 * W + 2 * lane_w(rr)) * CoutB + q * 8;
        } else {
            org = ((((int64_t)it.n * D + it.d0) * H + it.h0) * W + it.w0) * CoutB + (F32 ? 2 : 1) * it.co0;
            off = ((tile_d(rt) * H + tile_h(rt, rr)) * W + lane_w(rr)) * CoutB + q * 8;
        }
        return reinterpret_cast<const char*>(base + org) + (off * 2u);
    };
    // EPI >= 4 (the producers' drain, 4 waves x 4 parts): store instruction kk of producer wave w covers one half (BN = 64) or one row
    // (BN = 32) of column tile 4 w + kk / LPQ, so everything but the lane's place in it is wave-uniform: address = scalar base (tile origin +
    // the instruction's rows) + one of LPQ 32-bit lane offsets that never change - no per-store 64-bit offset to keep in registers (the
    // partial sums need them) and none to rebuild with quarter-rate integer multiplies either
    constexpr int LPQ = 32 / VPI_;                   // store instructions per column tile (4 | 2) = per part of the drain
    auto line_ptr = [&](const bf16_t* base, const FwdItem& it, int w, int kk) -> const char* {
        const int rt = 4 * w + kk / LPQ, j = kk % LPQ;
        const int hrow = 2 * (rt & 3) + ((j * VPI_) >> 4);
        const int64_t org = ((((int64_t)it.n * D + it.d0 + (rt >> 2)) * H + it.h0 + hrow) * W + it.w0) * Cout + it.co0;
        const int rr = (j * VPI_ + lane / CPV_) & 31;
        const unsigned off = (unsigned)(lane_w(rr) * Cout + (lane % CPV_) * 8) * 2u;
        return reinterpret_cast<const char*>(base + org) + off;
    };
    constexpr bool FASTD = ASY && EPI >= 4;         // (round 4: tried for EPI 0-3 as well - nothing gained, enc0b's input gradient 3 % slower)
    // vector-memory instructions one wave issues in store_share<NW> (the producers' counted waits step over exactly these): one store per
    // iteration, one more per iteration for the logits, the pooled pieces
    // (part `part` of `nparts`: the store instructions [part, part + 1) * NIT / nparts; the pooled pieces go with the last part)
    auto store_share_vmops = [&](int nw, int part, int nparts) {
        const int nit = 512 / (nw * VPI_) / nparts;
        int n = nit;
        if (MODE == 0 && !RES) {
            if (HAS_LOGITS && tail.logits) n += nit;
            if (HAS_POOL && tail.pool && part == nparts - 1) n += (64 * CPV_ + nw * 64 - 1) / (nw * 64);
        }
        return n;
    };
    // weights of the final 1x1x1 conv for this lane's 16-byte piece of a voxel (q = lane % CPV), fetched once per kernel: as loads inside
    // store_share they sat in every part of the asynchronous drain, in front of its stores
    float4 w1a = make_float4(0.f, 0.f, 0.f, 0.f), w1b = w1a;
    float b1v = 0.f;
    const float act_s = act == FMRI_ACT_RELU ? 0.f : (act == FMRI_ACT_LEAKY ? alpha : 1.f);
    if constexpr (MODE == 0 && !RES && HAS_LOGITS) {
        if (tail.logits) {
            w1a = *reinterpret_cast<const float4*>(tail.w1 + (lane % CPV_) * 8);
            w1b = *reinterpret_cast<const float4*>(tail.w1 + (lane % CPV_) * 8 + 4);
            b1v = tail.b1[0];
        }
    }
    float ns_[8] = {}, nq_[8] = {}, nsc[8] = {}, nsh[8] = {};
    int ng_n = -1, ng_co0 = -1;                      // (group, channel block) the partial sums belong to
    auto nsum_flush = [&]() {
        if constexpr (HAS_SUMS) {
            if (ng_co0 < 0) return;
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int m = CPV_; m < 64; m <<= 1) {
                    ns_[k] += __shfl_xor(ns_[k], m);
                    nq_[k] += __shfl_xor(nq_[k], m);
                    __builtin_amdgcn_sched_barrier(0);
                }
            if (lane < CPV_) {
                double* const p = tail.nws + ((((int64_t)blockIdx.x + 1) * tail.n_grp + ng_n) * Cout + ng_co0 + lane * 8) * 2;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    unsafeAtomicAdd(p + 2 * k, (double)ns_[k]);
                    unsafeAtomicAdd(p + 2 * k + 1, (double)nq_[k]);
                    __builtin_amdgcn_sched_barrier(0);          // one pair at a time: hoisted, the 16 conversions and addresses take 64 registers
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) ns_[k] = nq_[k] = 0.f;
        }
    };
    // in front of a tile's first store: the sums move on to the tile's (group, channel block) - and with them, EPI 5, the lane's scale / shift
    auto nsum_group = [&](const FwdItem& it) {
        if constexpr (HAS_SUMS) {
            const int gn = tail.n_per ? it.n : 0;
            if (gn == ng_n && it.co0 == ng_co0) return;
            nsum_flush();
            ng_n = gn;
            ng_co0 = it.co0;
            if constexpr (HAS_NBWD) {
                const float4* const t4 = reinterpret_cast<const float4*>(tail.nss + ((int64_t)gn * Cout + it.co0 + (lane % CPV_) * 8) * 2);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 t = t4[j];
                    nsc[2 * j] = t.x; nsh[2 * j] = t.y; nsc[2 * j + 1] = t.z; nsh[2 * j + 1] = t.w;
                }
            }
        }
    };
    auto store_share = [&](const FwdItem& it, int slot, int w, auto nw_tag, auto pre_tag, const uint4* mk, auto part_tag, auto nparts_tag) {
        constexpr int NW = decltype(nw_tag)::value;
        constexpr bool PRE = decltype(pre_tag)::value;       // the mask lines were requested earlier (mk[kk], kk = store instruction)
        constexpr int PART = decltype(part_tag)::value, NPARTS = decltype(nparts_tag)::value;
        constexpr int CPV = CPV_, VPI = VPI_;
        constexpr int NIT = 512 / (NW * VPI);        // store instructions per wave
        static_assert(NIT % NPARTS == 0, "parts");
        constexpr int SWM = BNB == 64 ? 7 : 3;
        const unsigned char* const stage = lds + slot * HALO_BYTES;
        if constexpr (HAS_SUMS && PART == 0) nsum_group(it);
        constexpr bool FD = FASTD && NW == 4;
        const int ln = lane;
        // (FD: the swizzle term of the staged voxel depends on the lane only - v & 7 = (lane / 8) & 7, (v >> 2) & 3 = (lane / 16) & 3 - so a
        // lane reads all its lines from one LDS address + immediate offsets)
        const int vs_l = BNB == 64 ? ((lane / CPV) & 7) : (((lane / CPV) >> 2) & 3);
        const unsigned char* const stage_l = stage + (lane / CPV) * (BNB * 2) + ((((lane % CPV) ^ vs_l) & SWM) << 4);
        auto gp = [&](const bf16_t* base, int kk, int v, int q) -> const char* {
            if constexpr (FD) return line_ptr(base, it, w, kk);
            else return piece_ptr(base, it, v, q);
        };
#pragma unroll
        for (int kk = PART * (NIT / NPARTS); kk < (PART + 1) * (NIT / NPARTS); ++kk) {
            const int v = w * (512 / NW) + kk * VPI + ln / CPV, q = ln % CPV;       // tile-wide voxel index: column tile v >> 5, lane v & 31
            const int vs = BNB == 64 ? (v & 7) : ((v >> 2) & 3);
            uint4 o4;
            if constexpr (FD) o4 = *reinterpret_cast<const uint4*>(stage_l + (w * (512 / NW) + kk * VPI) * (BNB * 2));
            else o4 = *reinterpret_cast<const uint4*>(stage + v * (BNB * 2) + (((q ^ vs) & SWM) << 4));
            const int rt = v >> 5, rr = v & 31;
            if (HAS_MASK && mask) {
                uint4 m4;
                if constexpr (PRE) m4 = mk[kk];
                else m4 = *reinterpret_cast<const uint4*>(gp(mask, kk, v, q));
                const unsigned mm[4] = {m4.x, m4.y, m4.z, m4.w};
                unsigned* const oo = reinterpret_cast<unsigned*>(&o4);
                if constexpr (F32) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) oo[i] = __uint_as_float(mm[i]) > 0.f ? oo[i] : 0u;       // four floats: dx = y > 0 ? dx : 0
                } else {
                unsigned z2 = 0u, o2 = 0x00010001u;
                asm volatile("" : "+v"(z2), "+v"(o2));          // two registers for the whole drain, not an immediate per instruction
#pragma unroll
                for (int i = 0; i < 4; ++i) oo[i] &= pos_mask2(mm[i], z2, o2);
                }
            }
            if constexpr (HAS_RESID) {
                uint4 r4;
                if constexpr (PRE) r4 = mk[kk];
                else r4 = *reinterpret_cast<const uint4*>(gp(residual, kk, v, q));
                const unsigned rr4[4] = {r4.x, r4.y, r4.z, r4.w};
                unsigned* const oo = reinterpret_cast<unsigned*>(&o4);
                if constexpr (F32) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float vv = __uint_as_float(oo[i]) + __uint_as_float(rr4[i]);
                        vv = vmax(vv, __builtin_fmaf(vv, act_s, 0.f));
                        oo[i] = __float_as_uint(vv);
                    }
                } else
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float lo = __uint_as_float(oo[i] << 16) + __uint_as_float(rr4[i] << 16);
                    float hi = __uint_as_float(oo[i] & 0xffff0000u) + __uint_as_float(rr4[i] & 0xffff0000u);
                    lo = vmax(lo, __builtin_fmaf(lo, act_s, 0.f));          // act(v) = max(v, s v), s = 0 ReLU / alpha LeakyReLU / 1 none
                    hi = vmax(hi, __builtin_fmaf(hi, act_s, 0.f));
                    oo[i] = pack2bf(lo, hi);
                }
            }
            if constexpr (HAS_NBWD) {
                uint4 x4;
                if constexpr (PRE) x4 = mk[LINES_PIPE ? (PART & 1) * (NIT / NPARTS) + kk - PART * (NIT / NPARTS) : kk];
                else x4 = *reinterpret_cast<const uint4*>(gp(mask, kk, v, q));
                const unsigned xx[4] = {x4.x, x4.y, x4.z, x4.w};
                unsigned* const oo = reinterpret_cast<unsigned*>(&o4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float xlo = __uint_as_float(xx[i] << 16), xhi = __uint_as_float(xx[i] & 0xffff0000u);
                    float dlo = __uint_as_float(oo[i] << 16), dhi = __uint_as_float(oo[i] & 0xffff0000u);
                    dlo = __builtin_fmaf(xlo, nsc[2 * i], nsh[2 * i]) > 0.f ? dlo : dlo * act_s;          // act'(z): 1 | s (0 ReLU, alpha LeakyReLU, 1 none)
                    dhi = __builtin_fmaf(xhi, nsc[2 * i + 1], nsh[2 * i + 1]) > 0.f ? dhi : dhi * act_s;
                    oo[i] = pack2bf(dlo, dhi);
                    dlo = __uint_as_float(oo[i] << 16);                 // the sums run over the values as stored (what the apply pass reads)
                    dhi = __uint_as_float(oo[i] & 0xffff0000u);
                    ns_[2 * i] += dlo;
                    ns_[2 * i + 1] += dhi;
                    nq_[2 * i] = __builtin_fmaf(dlo, xlo, nq_[2 * i]);
                    nq_[2 * i + 1] = __builtin_fmaf(dhi, xhi, nq_[2 * i + 1]);
                }
            }
            if constexpr (HAS_STATS) {
                const unsigned oo[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float lo = __uint_as_float(oo[i] << 16), hi = __uint_as_float(oo[i] & 0xffff0000u);
                    ns_[2 * i] += lo;
                    ns_[2 * i + 1] += hi;
                    nq_[2 * i] = __builtin_fmaf(lo, lo, nq_[2 * i]);
                    nq_[2 * i + 1] = __builtin_fmaf(hi, hi, nq_[2 * i + 1]);
                }
            }
            *reinterpret_cast<uint4*>(const_cast<char*>(gp(y, kk, v, q))) = o4;
            if constexpr (MODE == 0 && !RES && HAS_LOGITS) {
                if (tail.logits) {
                    // final 1x1x1 conv to one label: the CPV lanes of a voxel hold its BN (= Cout) channels, 8 each
                    const float4 wa = w1a, wb = w1b;           // this lane's 8 weights (q = lane % CPV is the same in every iteration)
                    float part = __uint_as_float(o4.x << 16) * wa.x;
                    part = __builtin_fmaf(__uint_as_float(o4.x & 0xffff0000u), wa.y, part);
                    part = __builtin_fmaf(__uint_as_float(o4.y << 16), wa.z, part);
                    part = __builtin_fmaf(__uint_as_float(o4.y & 0xffff0000u), wa.w, part);
                    part = __builtin_fmaf(__uint_as_float(o4.z << 16), wb.x, part);
                    part = __builtin_fmaf(__uint_as_float(o4.z & 0xffff0000u), wb.y, part);
                    part = __builtin_fmaf(__uint_as_float(o4.w << 16), wb.z, part);
                    part = __builtin_fmaf(__uint_as_float(o4.w & 0xffff0000u), wb.w, part);
#pragma unroll
                    for (int m = 1; m < CPV; m <<= 1) part += __shfl_xor(part, m);
                    if (q == 0)
                        tail.logits[(((int64_t)it.n * D + it.d0 + tile_d(rt)) * H + it.h0 + tile_h(rt, rr)) * W + it.w0 + lane_w(rr)] = part + b1v;
                }
            }
            // four pieces at a time: left alone the scheduler hoists all NIT stage reads and addresses to the top (96+ live registers: the
            // producer path of the ASYNC kernel spilled 150 of them)
            if constexpr (NW == 4) {
                if ((kk & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (MODE == 0 && !RES && HAS_POOL && PART == NPARTS - 1) {
            if (tail.pool) {
                // 2x2x2 max pooling of the staged tile: 2 x 4 x 8 pooled voxels x CPV 16-byte pieces, one per thread and round
                constexpr int NPOOL = (64 * CPV + NW * 64 - 1) / (NW * 64);
#pragma unroll
                for (int pi = 0; pi < NPOOL; ++pi) {
                    const int idx = (pi * NW + w) * 64 + lane;
                    const bool live = idx < 64 * CPV;         // wave-uniform (64 * CPV is a multiple of 64)
                    const int pv = (live ? idx : 0) / CPV, q = idx % CPV;
                    const int pd = pv >> 5, ph = (pv >> 3) & 3, pw = pv & 7;
                    float mx[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const int d = 2 * pd + (c >> 2), h = 2 * ph + ((c >> 1) & 1), w_ = 2 * pw + (c & 1);
                        // inverse of (tile_d, tile_h, lane_w): column tile rt = d*4 + h/2, lane rr = (h&1)*16 + (w rotated by HW mod 16 on odd rows)
                        const int v = (d * 4 + (h >> 1)) * 32 + ((h & 1) ? 16 + ((w_ + (HW & 15)) & 15) : w_);
                        const int vs = BNB == 64 ? (v & 7) : ((v >> 2) & 3);
                        const uint4 p4 = *reinterpret_cast<const uint4*>(stage + v * (BNB * 2) + (((q ^ vs) & SWM) << 4));
                        const unsigned pp[4] = {p4.x, p4.y, p4.z, p4.w};
                        if constexpr (F32) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) mx[i] = c == 0 ? __uint_as_float(pp[i]) : vmax(mx[i], __uint_as_float(pp[i]));
                        } else
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float lo = __uint_as_float(pp[i] << 16), hi = __uint_as_float(pp[i] & 0xffff0000u);
                            mx[2 * i] = c == 0 ? lo : vmax(mx[2 * i], lo);
                            mx[2 * i + 1] = c == 0 ? hi : vmax(mx[2 * i + 1], hi);
                        }
                    }
                    uint4 o;
                    if constexpr (F32) o = make_uint4(__float_as_uint(mx[0]), __float_as_uint(mx[1]), __float_as_uint(mx[2]), __float_as_uint(mx[3]));
                    else {
                    o.x = (__float_as_uint(mx[0]) >> 16) | (__float_as_uint(mx[1]) & 0xffff0000u);
                    o.y = (__float_as_uint(mx[2]) >> 16) | (__float_as_uint(mx[3]) & 0xffff0000u);
                    o.z = (__float_as_uint(mx[4]) >> 16) | (__float_as_uint(mx[5]) & 0xffff0000u);
                    o.w = (__float_as_uint(mx[6]) >> 16) | (__float_as_uint(mx[7]) & 0xffff0000u);
                    }
                    const int64_t po = ((((int64_t)it.n * (D >> 1) + (it.d0 >> 1) + pd) * (H >> 1) + (it.h0 >> 1) + ph) * (W >> 1) + (it.w0 >> 1) + pw) * CoutB +
                                       (F32 ? 2 : 1) * it.co0 + q * 8;
                    if (live) *reinterpret_cast<uint4*>(tail.pool + po) = o;
                }
            }
        }
    };

    if (wv >= DW) {
        // ------------------------------------------------------------------------------------------------------------ producer
        const int dwv = wv - DW;
        unsigned f_voff[F_PER_WAVE];
#pragma unroll
        for (int k = 0; k < F_PER_WAVE; ++k) {
            const int instr = (dwv + DW * k) % F_INSTR;
            const int i = instr * 64 + lane;
            const int row = i >> 2, ps = i & 3;
            const int ls = ps ^ ((row >> 2) & 3);
            f_voff[k] = (unsigned)(((row / BN) * Cout + (row % BN)) * Krow + ls * 8) * 2u;
        }
        // halo piece ph: hd | hh<<4 | hw<<8 | ls<<13 | valid<<15 | edge<<16, edge = which faces of the halo box the row lies on
        // (bit 0 hd == 0, 1 hd == HD-1, 2 hh == 0, 3 hh == HH-1, 4 hw == 0, 5 hw == HW-1): the only rows a border tile can have outside the volume
        auto make_pack = [&](int ph, int ln) {
            const int i = (H_I0 + ph * DW + dwv) * 64 + ln;
            const int hv = i >> 2, ps = i & 3;
            const int hvc = hv < HVOX ? hv : 0;
            const int hw_ = hvc % HW, hq = hvc / HW;
            const int ls = ps ^ halo_key(hw_);
            const int hd_ = hq / HH, hh_ = hq % HH;
            // (bit 6: a row past the end of the box - "outside" whatever the tile)
            const int edge = (hd_ == 0 ? 1 : 0) | (hd_ == HD - 1 ? 2 : 0) | (hh_ == 0 ? 4 : 0) | (hh_ == HH - 1 ? 8 : 0) | (hw_ == 0 ? 16 : 0) | (hw_ == HW - 1 ? 32 : 0) |
                             (hv < HVOX ? 0 : 64);
            return hd_ | (hh_ << 4) | (hw_ << 8) | (ls << 13) | ((hv < HVOX ? 1 : 0) << 15) | (edge << 16);
        };
        // (EPI 5 / 6: no table either - a piece's descriptor is rebuilt from the lane index when the piece is issued, see KEEP_HP below)
        constexpr bool KEEP_PACK = EPI < 5;
        int h_pack[KEEP_PACK ? NPIECE : 1];
        if constexpr (KEEP_PACK) {
#pragma unroll
            for (int ph = 0; ph < NPIECE; ++ph) h_pack[ph] = make_pack(ph, lane);
        }
        auto pack_of = [&](int ph) {
            if constexpr (KEEP_PACK) return h_pack[ph];
            else {
                int ln = lane;
                asm volatile("" : "+v"(ln));         // (or hipcc hoists all NPIECE descriptors out of the tile loop again)
                return make_pack(ph, ln);
            }
        };
        auto issue_filter = [&](const FwdItem& it, int pl, int fb) {
            int64_t slab;
            int koff;
            if constexpr (MODE == 0) { slab = PH0 + pl; koff = it.ch << 5; }
            else if constexpr (MODE == 1) { slab = it.par * NROW + pl * RPP; koff = it.ch << 5; }
            else { slab = (it.ch / kpc) * NROW + pl * RPP; koff = (it.ch % kpc) << 5; }
            const bf16_t* const base = wt + ((slab * NKW * Cout + it.co0) * Krow + koff);
#pragma unroll
            for (int k = 0; k < F_PER_WAVE; ++k)
                if (k == 0 || dwv + DW * k < F_INSTR)
                    dma16_s(base, f_voff[k], __builtin_amdgcn_readfirstlane(ldsf0 + fb * FILT_BYTES + ((dwv + DW * k) % F_INSTR) * 1024));
        };
        // source address of every halo piece, kept across the chunks of a tile (+ 32 channels per chunk) - except in the instantiations
        // whose drain needs the registers (EPI 5 / 6: prefetched lines + partial sums): those rebuild a piece's address when they issue it
        constexpr bool KEEP_HP = EPI < 5;
        const bf16_t* hp[(KEEP_HP && !(FH && !PL && EPI < 5)) ? NPIECE : 1];
        unsigned hoff[(FH && !PL && EPI < 5) ? NPIECE : 1];      // FH: the per-lane byte offset of each piece inside the halo box (constant for the whole kernel)
        unsigned heff[(FH && !PL && EPI < 5) ? NPIECE : 1];      // ... as issued for the current tile: out of range (DMA_OOB) where the row leaves the volume
        auto halo_src = [&](const FwdItem& it, int pk) -> const bf16_t* {
            int od = 0, oh = 0, ow = 0;                            // TIGHT: the box starts at g - 1 + (parity of the taps) per axis
            if constexpr (TIGHT) {
                const int tp = MODE == 1 ? it.par : (NPAR - 1) - it.ch / kpc;
                od = tp >> 2; oh = (tp >> 1) & 1; ow = tp & 1;
            }
            const int gd = it.d0 - 1 + od + (pk & 15), gh = it.h0 - 1 + oh + ((pk >> 4) & 15), gw = it.w0 - 1 + ow + ((pk >> 8) & 31);
            const int ls = (pk >> 13) & 3;
            const bool ok = ((pk >> 15) & 1) && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
            if constexpr (MODE == 2) {
                const int p = it.ch / kpc, coff = (it.ch % kpc) << 5;
                const int sd = PL ? min(max(gd, 0), D - 1) : 2 * min(max(gd, 0), D - 1) + (p >> 2);
                const int sh = 2 * min(max(gh, 0), H - 1) + ((p >> 1) & 1), sw = 2 * min(max(gw, 0), W - 1) + (p & 1);
                const int64_t off = ((((int64_t)it.n * (PL ? D : 2 * D) + sd) * 2 * H + sh) * 2 * W + sw) * s.C0 + coff + ls * 8;
                return ok ? s.p0 + off : (const bf16_t*)zpage;
            } else {
                const int cc = it.ch << 5;
                const bool from0 = cc < s.C0;
                const bf16_t* sp = from0 ? s.p0 : s.p1;
                const int sC = from0 ? s.C0 : s.C1;
                const int coff = from0 ? cc : cc - s.C0;
                const int sh = (from0 && s.up0) ? 1 : 0;
                const int shd = sh & s.dsh;
                const int sD = D >> shd, sH = H >> sh, sW = W >> sh;
                const int gdc = min(max(gd, 0), D - 1) >> shd, ghc = min(max(gh, 0), H - 1) >> sh, gwc = min(max(gw, 0), W - 1) >> sh;
                const int off = ((gdc * sH + ghc) * sW + gwc) * sC + coff + ls * 8;
                const bf16_t* real = sp + (int64_t)it.n * sD * sH * sW * sC + off;
                return ok ? real : (const bf16_t*)zpage;
            }
        };
        static_assert(NPIECE * DW >= H_I1 - H_I0 && NPIECE * DW - (H_I1 - H_I0) < DW, "piece map");
        auto issue_halo = [&](int ph, int slot, const bf16_t* src) {
            const int instr = H_I0 + ph * DW + dwv;
            // every wave issues exactly NPIECE instructions per chunk (the counted s_waitcnt below relies on it): the few past the last live
            // instruction copy zeros into the dead rows behind it
            const bool dead = instr >= H_I1;
            dma16(dead ? (const void*)zpage : (const void*)src, __builtin_amdgcn_readfirstlane(lds0 + slot * HALO_STRIDE + instr * 1024));
        };
        // FAST halo addressing (round 4; plain single-source 3-D launches = every MODE 0 launch of the benchmarked step).  The producers'
        // per-phase instrumentation (tools/prof_phases.py) showed a piece whose address is worked out afresh (halo_src: clamps, three bounds
        // checks, 64-bit multiplies - ~50 VALU instructions) costing its wave ~600 cycles against ~150 for one that only advances its pointer:
        // on a one-chunk layer EVERY piece is fresh (17 per wave and tile) and the producers, not the MFMA waves, set the phase time
        // (enc0b: 3,300-cycle halo phases against 1,900 cycles of MFMA work).  Here a piece's address is
        //     [wave-uniform base of the tile's halo corner (n, d0-1, h0-1, w0-1), chunk included]  +  [per-lane constant of the piece]
        // The constant ((hd*H + hh)*W + hw)*C0 + 8*ls never changes during the kernel: it is worked out once and kept in hp[] (as a byte
        // offset).  Tiles and the volume are aligned, so a row can only be outside the volume if it lies on a face of the halo box (edge
        // bits of h_pack) on a side where the tile touches the volume's border (6 wave-uniform bits): interior tiles issue
        // `global_load_lds` with a scalar base + 32-bit lane offset and NO vector address arithmetic, border tiles select the zero page per lane.
        // The parity modes (TIGHT box; MODE 2 gathers voxel 2g + p of dy) have the same structure: the parity moves the box's origin (od, oh, ow) -
        // and with it which faces can leave the volume - and, MODE 2, the scalar base; the lane constant uses doubled strides there.
        constexpr bool FASTH_CT = FH && !PL && KEEP_HP && KEEP_PACK;
        // ... for launches whose pieces are mostly FRESH: with many chunks per tile (or, MODE 2, per parity class) the old scheme's "advance
        // every lane's pointer by 64 B" (two vector instructions per piece) beats one base computation per item + a per-lane border select per
        // piece (measured: dec2a / dec1a parity launches 3-7 % slower with the fast form, enc0b 10-15 % and dec0a's input gradient 7 % faster).
        // Its OWN instantiation (template parameter FH, chosen by the launcher for single-source launches with at most two chunks per fresh
        // address): compiled into one kernel behind a run-time switch, the second path cost the launches that do not take it 5-8 %
        // (profiles/r04_fast_halo_ab.log).
        constexpr bool fasth = FASTH_CT;
        auto fast_off = [&](int pk) -> unsigned {
            const int hd_ = pk & 15, hh_ = (pk >> 4) & 15, hw_ = (pk >> 8) & 31, ls = (pk >> 13) & 3;
            if constexpr (MODE == 2) return (unsigned)((((2 * hd_) * (2 * H) + 2 * hh_) * (2 * W) + 2 * hw_) * s.C0 + ls * 8) * 2u;
            else return (unsigned)(((hd_ * H + hh_) * W + hw_) * s.C0 + ls * 8) * 2u;
        };
        auto fast_base = [&](const FwdItem& it, unsigned& tmask) -> const char* {
            int od = 0, oh = 0, ow = 0;
            bool lo_d = it.d0 == 0, hi_d = it.d0 + TD == D, lo_h = it.h0 == 0, hi_h = it.h0 + TH == H, lo_w = it.w0 == 0, hi_w = it.w0 + TW == W;
            if constexpr (TIGHT) {
                const int tp = MODE == 1 ? it.par : (NPAR - 1) - it.ch / kpc;
                od = tp >> 2; oh = (tp >> 1) & 1; ow = tp & 1;
                // box rows g - 1 + o .. g + T - 1 + o per axis: row 0 is outside only for o = 0 on the low border, row T only for o = 1 on the high one
                lo_d = lo_d && od == 0; hi_d = hi_d && od == 1;
                lo_h = lo_h && oh == 0; hi_h = hi_h && oh == 1;
                lo_w = lo_w && ow == 0; hi_w = hi_w && ow == 1;
            }
            // (the item's fields come out of integer divisions by run-time values, i.e. out of the vector ALU: wave-uniform, but in vector
            // registers - the scalar base and the border mask are moved to scalar registers explicitly)
            tmask = __builtin_amdgcn_readfirstlane((lo_d ? 1u : 0u) | (hi_d ? 2u : 0u) | (lo_h ? 4u : 0u) | (hi_h ? 8u : 0u) | (lo_w ? 16u : 0u) | (hi_w ? 32u : 0u));
            const int bd = it.d0 - 1 + od, bh = it.h0 - 1 + oh, bw = it.w0 - 1 + ow;
            int64_t eoff;
            if constexpr (MODE == 2) {
                const int p = it.ch / kpc, coff = (it.ch % kpc) << 5;
                eoff = ((((int64_t)it.n * 2 * D + 2 * bd + (p >> 2)) * 2 * H + 2 * bh + ((p >> 1) & 1)) * 2 * W + 2 * bw + (p & 1)) * s.C0 + coff;
            } else {
                eoff = ((((int64_t)it.n * D + bd) * H + bh) * W + bw) * s.C0 + (it.ch << 5);
            }
            const unsigned long long a = reinterpret_cast<unsigned long long>(s.p0 + eoff);
            const unsigned alo = __builtin_amdgcn_readfirstlane((unsigned)a), ahi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
            return reinterpret_cast<const char*>(((unsigned long long)ahi << 32) | alo);
        };
        // (round 5: through a buffer descriptor whose base is the item's box corner.  Which rows of a piece lie outside the volume depends on
        // the tile only, so the offsets are fixed up when the stream moves to a new tile (`fresh_item`: three vector instructions per piece)
        // and every other chunk of the tile issues its pieces with NO vector arithmetic at all, border tiles included - the round-4 form
        // selected the zero page per lane for every piece of every chunk of a border tile, and the launches with many chunks per tile, whose
        // deep levels consist of border tiles, stayed on the pointer-advance scheme for it.)
        auto issue_halo_fast = [&](int ph, int slot, i32x4 rs, unsigned tmask, bool fresh_item) {
            const int instr = H_I0 + ph * DW + dwv;
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + slot * HALO_STRIDE + instr * 1024);
            if (fresh_item) {
                const bool bad = instr >= H_I1 || (((unsigned)h_pack[ph] >> 16) & (tmask | 64u)) != 0;
                heff[ph] = bad ? DMA_OOB : hoff[ph];
            }
            dma16_buf(rs, heff[ph], dst);
        };
        // Halo pieces of the NEXT chunk issued in phase pl: spread over the first NPH-1 phases (all of them when a chunk has one phase
        // only... it has at least two), so that the last phase's wait - everything landed - finds them a phase old.
        // (ASY: the first item of a tile issues none in phase 0 - the slot they go to is the staged tile the producers drain in that phase)
        auto pieces_from = [](int pl, bool drain) {
            if (drain) return pl <= DP ? 0 : (pl >= NPH - 1 ? NPIECE : (pl - DP) * NPIECE / (NPH - 1 - DP > 0 ? NPH - 1 - DP : 1));
            // two-phase chunks (the tight parity modes; round 4): a third of the pieces in the LAST phase as well - all twelve in phase 0 made that
            // phase producer-bound (3,300 cycles of issue against ~2,500 of MFMA work) and left phase 1 at 600; the pieces of the last phase
            // still land before the next chunk's full wait (dec0a / dec1a / dec2a forward and input gradient -1 ... -3.5 %, same-box A/B)
            if (NPH == 2) return pl >= NPH ? NPIECE : (pl * 2 * NPIECE) / 3;
            return pl >= NPH - 1 ? NPIECE : pl * NPIECE / (NPH - 1);
        };
        auto wait_newer = [](int n) {      // wait until at most n of this wave's vector-memory instructions are in flight (wave-uniform n)
#define FMRI_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
            switch (n) {
                FMRI_W(0) FMRI_W(1) FMRI_W(2) FMRI_W(3) FMRI_W(4) FMRI_W(5) FMRI_W(6) FMRI_W(7) FMRI_W(8) FMRI_W(9) FMRI_W(10) FMRI_W(11)
                FMRI_W(12) FMRI_W(13) FMRI_W(14) FMRI_W(15) FMRI_W(16) FMRI_W(17) FMRI_W(18) FMRI_W(19) FMRI_W(20) FMRI_W(21) FMRI_W(22)
                FMRI_W(23) FMRI_W(24) FMRI_W(25) FMRI_W(26) FMRI_W(27) FMRI_W(28) FMRI_W(29) FMRI_W(30) FMRI_W(31) FMRI_W(32) FMRI_W(33)
                FMRI_W(34) FMRI_W(35) FMRI_W(36) FMRI_W(37) FMRI_W(38) FMRI_W(39) FMRI_W(40)
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
#undef FMRI_W
        };
        // ASY: the finished tile waiting to be stored (staged by the consumers in halo slot `hb ^ 1` as seen from the next item), the mask
        // lines of this wave's share of it, and the number of vector-memory instructions the drain issues behind the phase-1 filter slab
        constexpr int NIT_ = 512 / (DW * VPI_);                  // store instructions (= lines) per producer wave and tile
        constexpr int LP = NIT_ / (DP > 0 ? DP : 1);              // ... per part of the drain
        constexpr int NDR = (ASY && HAS_LINES) ? (LINES_PIPE ? 2 * LP : NIT_) : 1;
        uint4 mk[NDR] = {};
        // the lines of part `part` of tile `it`'s drain -> buffer part & 1 (LINES_PIPE)
        auto fetch_lines = [&](const FwdItem& it, auto part_tag) {
            constexpr int PART = decltype(part_tag)::value;
            if constexpr (LINES_PIPE) {
#pragma unroll
                for (int i = 0; i < LP; ++i) mk[(PART & 1) * LP + i] = *reinterpret_cast<const uint4*>(line_ptr(mask, it, dwv, PART * LP + i));
            }
        };
        FwdItem done = cur;
        bool pending = false;
#ifdef FMRI_PROF
        unsigned long long pprof[12] = {};       // producers: [0] counted DMA wait, [11] barrier wait, [2] issue (filter slab, drain part, halo pieces)
        unsigned long long pph[36] = {};
#endif
        constexpr int DPN = DP > 0 ? DP : 1;
        int drain_vmops[DPN];
#pragma unroll
        for (int i = 0; i < DPN; ++i) drain_vmops[i] = ASY ? store_share_vmops(DW, i, DPN) + (LINES_PIPE && i + 1 < DPN ? LP : 0) : 0;
        if constexpr (FASTH_CT) {
            unsigned tm0;
            const i32x4 rs0 = dma_rsrc(fast_base(cur, tm0));
#pragma unroll
            for (int ph = 0; ph < NPIECE; ++ph) {
                hoff[ph] = fast_off(h_pack[ph]);
                issue_halo_fast(ph, 0, rs0, tm0, true);
            }
        } else {
#pragma unroll
            for (int ph = 0; ph < NPIECE; ++ph) {
                const bf16_t* const src = halo_src(cur, pack_of(ph));
                if constexpr (KEEP_HP) hp[ph] = src;
                issue_halo(ph, 0, src);
            }
        }
        issue_filter(cur, 0, 0);
        while (true) {
            bool has_next = true;
            FwdItem nxt = cur;
            int npair = pair;
            if (cur.ch + 1 < nch) nxt.ch = cur.ch + 1;
            else {
                npair = pair + gridDim.x;
                has_next = npair < npairs;
                if (has_next) nxt = decode(npair, 0);
            }
            const bool fresh = MODE == 2 ? (nxt.ch % kpc == 0) : (nxt.ch == 0 || (nxt.ch << 5) == s.C0);
            unsigned ntm = 0;
            const char* nsb = nullptr;
            i32x4 nrs = {0, 0, 0, 0};
            if constexpr (FASTH_CT) {
                if (has_next) nrs = dma_rsrc(fast_base(nxt, ntm));
            }
            // CHEAP_FRESH (tight parity modes outside FH: many chunks per fresh address): the pointers are kept and advanced as before, but a
            // FRESH pointer is formed the fast way - base of the box corner + lane constant, zero page where an edge row leaves the volume -
            // instead of halo_src's clamps and 64-bit multiplies (these launches always have one plain source: conv3d_api.hip).  dec2a / dec1a
            // input gradient -2.7 %, dec0a / dec1a forward -1 ... -2 % (same-box A/B)
            constexpr bool CHEAP_FRESH = TIGHT && KEEP_HP && KEEP_PACK && !FASTH_CT;
            if constexpr (CHEAP_FRESH) {
                if (has_next && fresh) nsb = fast_base(nxt, ntm);
            }
            // keep the packed piece descriptors packed: hipcc otherwise hoists the three bit-field extractions of every piece out of the loop
            // (51 more live registers) and spills them
#pragma unroll
            for (int ph = 0; ph < (KEEP_PACK ? NPIECE : 1); ++ph) asm volatile("" : "+v"(h_pack[ph]));
            // the phases of one item; DRN (ASY only): a staged tile is drained in phase 0 and the halo pieces start a phase later
            auto run_phases = [&](auto drn_tag) {
                constexpr bool DRN = decltype(drn_tag)::value;
                static_for<NPH>([&](auto pl_tag) {
                    constexpr int pl = decltype(pl_tag)::value;
                    PROF_T(pw0);
                    // This phase's filter slab was issued one phase ago, FOLLOWED by that phase's halo pieces (which belong to the next chunk)
                    // or by the stores of the drain: wait for the slab only and leave those in flight - a full vmcnt(0) here exposes the
                    // memory latency of every piece in every phase (measured: the producer chain issue + latency, not the MFMAs, set the
                    // phase time).  The first phase of a chunk needs its whole halo: everything must have landed, and the youngest piece is
                    // a phase old by then.
                    if constexpr (pl == 0) {
                        wait_newer(0);
                        if constexpr (DRN) {
                            // the mask lines requested in the previous tile's last phase are in their registers now: tell the compiler here,
                            // where it costs nothing (its own wait for them would otherwise sit behind the next filter slab's DMA)
                            if ((HAS_MASK && mask) || HAS_RESID || HAS_NBWD) {
#pragma unroll
                                for (int i = 0; i < (LINES_PIPE ? LP : NDR); ++i) asm volatile("" : "+v"(mk[i].x), "+v"(mk[i].y), "+v"(mk[i].z), "+v"(mk[i].w));
                            }
                        }
                    } else if constexpr (DRN && pl <= DP) {
                        wait_newer(drain_vmops[pl - 1]);
                        if constexpr (LINES_PIPE && pl < DP) {
                            // part pl's lines were requested a phase ago, in front of part pl - 1's stores: the compiler's own wait for them
                            // (it does not see the DMA instructions) belongs here, before this phase's filter slab is issued
#pragma unroll
                            for (int i = 0; i < LP; ++i) {
                                uint4& m = mk[(pl & 1) * LP + i];
                                asm volatile("" : "+v"(m.x), "+v"(m.y), "+v"(m.z), "+v"(m.w));
                            }
                        }
                    }
                    else {
                        int nfl = has_next ? pieces_from(pl, DRN) - pieces_from(pl - 1, DRN) : 0;
                        wait_newer(nfl);
                    }
                    PROF_T(pw1);
                    __builtin_amdgcn_s_barrier();                          // everybody's has landed; the previous phase is fully read
                    PROF_T(pw2);
                    if (pl < NPH - 1) issue_filter(cur, pl + 1, (g + 1) & 1);
                    else if (has_next) issue_filter(nxt, 0, (g + 1) & 1);
                    if constexpr (DRN) {
                        // part pl of the drain
                        if constexpr (pl < DP) {
                            if constexpr (LINES_PIPE && pl + 1 < DP) fetch_lines(done, std::integral_constant<int, pl + 1>{});
                            store_share(done, hb ^ 1, dwv, std::integral_constant<int, DW>{}, std::true_type{}, mk,
                                        std::integral_constant<int, pl>{}, std::integral_constant<int, DPN>{});
                        }
                    }
                    if (has_next) {
                        const int HP0 = pieces_from(pl, DRN), HPN = pieces_from(pl + 1, DRN) - HP0;
#pragma unroll
                        for (int q = 0; q < NPIECE; ++q) {
                            if (q < HPN) {
                                if constexpr (FASTH_CT) issue_halo_fast(HP0 + q, hb ^ 1, nrs, ntm, fresh);
                                else if constexpr (KEEP_HP) {
                                    if (fresh) {
                                        if constexpr (CHEAP_FRESH) {
                                            const int pk = h_pack[HP0 + q];
                                            const bool bad = !((pk >> 15) & 1) || (((unsigned)pk >> 16) & ntm) != 0;
                                            hp[HP0 + q] = bad ? (const bf16_t*)zpage : reinterpret_cast<const bf16_t*>(nsb + fast_off(pk));
                                        } else hp[HP0 + q] = halo_src(nxt, h_pack[HP0 + q]);
                                    } else hp[HP0 + q] += 32;
                                    issue_halo(HP0 + q, hb ^ 1, hp[HP0 + q]);
                                } else issue_halo(HP0 + q, hb ^ 1, halo_src(nxt, pack_of(HP0 + q)));
                            }
                        }
                    }
                    if constexpr (ASY) {
                        // last phase of a tile (it carries no halo pieces): request the mask lines of this wave's share of the tile's stores
                        if (pl == NPH - 1 && cur.ch == nch - 1 && ((HAS_MASK && mask) || HAS_RESID || HAS_NBWD)) {
                            const bf16_t* const lines = HAS_RESID ? residual : mask;      // (the residual of a tile is read before anyone stores to it)
                            if constexpr (LINES_PIPE) fetch_lines(cur, std::integral_constant<int, 0>{});
                            else {
#pragma unroll
                                for (int kk = 0; kk < NDR; ++kk) {
                                    const int v = dwv * (512 / DW) + kk * VPI_ + lane / CPV_;
                                    if constexpr (FASTD) mk[kk] = *reinterpret_cast<const uint4*>(line_ptr(lines, cur, dwv, kk));
                                    else mk[kk] = *reinterpret_cast<const uint4*>(piece_ptr(lines, cur, v, lane % CPV_));
                                }
                            }
                        }
                    }
#ifdef FMRI_PROF
                    { PROF_T(pw3); pprof[0] += pw1 - pw0; pprof[11] += pw2 - pw1; pprof[2] += pw3 - pw2;
                      if (pl < 9) { pph[(DRN ? 0 : 9) + pl] += pw3 - pw2; pph[18 + (DRN ? 0 : 9) + pl] += 1; } }
#endif
                    ++g;
                });
            };
            if constexpr (ASY) {
                if (pending) run_phases(std::true_type{});
                else run_phases(std::false_type{});
                pending = false;
            } else run_phases(std::false_type{});
            if (cur.ch == nch - 1) {
                __builtin_amdgcn_s_barrier();                          // the halo slot is fully read: the consumers stage the tile in it
                if constexpr (ASY) {
                    done = cur;
                    pending = true;
                    if (!has_next) {                                   // the last tile of this workgroup: nothing left to hide the stores under
                        __builtin_amdgcn_s_barrier();                  // staged
                        store_share(done, hb, dwv, std::integral_constant<int, DW>{}, std::integral_constant<bool, !LINES_PIPE>{}, mk, std::integral_constant<int, 0>{},
                                    std::integral_constant<int, 1>{});
                    }
                } else if constexpr (!RES) {
                    __builtin_amdgcn_s_barrier();                      // staged
                    store_share(cur, hb, wv, std::integral_constant<int, 8>{}, std::false_type{}, nullptr, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
                }
            }
            if (!has_next) break;
            cur = nxt;
            pair = npair;
            hb ^= 1;
        }
        nsum_flush();
#ifdef FMRI_PROF
        if (lane == 0) {
            atomicAdd(&g_prof[0], pprof[0]); atomicAdd(&g_prof[11], pprof[11]); atomicAdd(&g_prof[2], pprof[2]);
            for (int i = 0; i < 36; ++i) if (pph[i]) atomicAdd(&g_prof_ph[i], pph[i]);
        }
#endif
        return;
    }

    // ---------------------------------------------------------------------------------------------------------------- consumer
    const int cw = wv;
    if (tail.prio) __builtin_amdgcn_s_setprio(3);
    static_assert(!(S16 && PL), "the 16x16x32 form covers the 3-D launches");
    // S16: fragment index j = h-row of the wave's d-plane (16 voxels along w: lane & 15), c = 16-channel block; a lane's four accumulator
    // registers are channels 16 c + 4 kq .. + 3 of its voxel (kq = lane >> 4, also the 8-channel k-group the lane reads of both operands)
    constexpr int JN = S16 ? 8 : JT, CN = S16 ? 2 * NT : NT;
    typedef std::conditional_t<S16, f32x4, f32x16> acc_t;
    const int w16 = lane & 15, kq = lane >> 4;
    acc_t acc[JN][CN];
    // (S16 keeps only the first NT floats4 of bv: [c >> 2][c & 3] = the lane's 4 channels of 16-channel block c)
    auto load_bias = [&](int co0, float4 (&bv)[NT][4]) {
        if constexpr (S16) {
#pragma unroll
            for (int c = 0; c < CN; ++c)
                bv[c >> 2][c & 3] = bias ? *reinterpret_cast<const float4*>(bias + co0 + c * 16 + 4 * kq) : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
#pragma unroll
            for (int c = 0; c < NT; ++c)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    bv[c][gq] = bias ? *reinterpret_cast<const float4*>(bias + co0 + c * 32 + 8 * gq + 4 * hk) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto init_acc = [&](const float4 (&bv)[NT][4]) {
        if constexpr (S16) {
#pragma unroll
            for (int j = 0; j < JN; ++j)
#pragma unroll
                for (int c = 0; c < CN; ++c) {
                    acc[j][c] = __builtin_bit_cast(f32x4, bv[c >> 2][c & 3]);
                }
        } else {
#pragma unroll
            for (int j = 0; j < JT; ++j)
#pragma unroll
                for (int c = 0; c < NT; ++c)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        acc[j][c][4 * gq] = bv[c][gq].x;
                        acc[j][c][4 * gq + 1] = bv[c][gq].y;
                        acc[j][c][4 * gq + 2] = bv[c][gq].z;
                        acc[j][c][4 * gq + 3] = bv[c][gq].w;
                    }
        }
    };
    // RES with per-border-class bias (tail.bias27): accumulators of the tile `it` start from the bias of each lane's own output voxel.
    // Interior tiles (most) take the interior class 13 for every lane; tiles on a face of the volume look the class up per voxel.
    auto init_acc_b27 = [&](const FwdItem& it) {
        const bool border = it.d0 == 0 || it.d0 + TD == D || it.h0 == 0 || it.h0 + TH == H || it.w0 == 0 || it.w0 + TW == W;
        if constexpr (S16) {
#pragma unroll
            for (int j = 0; j < JN; ++j) {
                int cls = 13;
                if (border) {
                    const int d = it.d0 + cw, h = it.h0 + j, w = it.w0 + w16;
                    cls = ((d == 0 ? 0 : (d == D - 1 ? 2 : 1)) * 3 + (h == 0 ? 0 : (h == H - 1 ? 2 : 1))) * 3 + (w == 0 ? 0 : (w == W - 1 ? 2 : 1));
                }
                const float* const bp = tail.bias27 + (int64_t)cls * Cout + it.co0 + 4 * kq;
#pragma unroll
                for (int c = 0; c < CN; ++c) {
                    acc[j][c] = *reinterpret_cast<const f32x4*>(bp + c * 16);
                }
            }
        } else {
#pragma unroll
        for (int j = 0; j < JT; ++j) {
            int cls = 13;
            if (border) {
                const int rt = JT * cw + j;
                const int d = it.d0 + tile_d(rt), h = it.h0 + tile_h(rt, r), w = it.w0 + lane_w(r);
                cls = ((d == 0 ? 0 : (d == D - 1 ? 2 : 1)) * 3 + (h == 0 ? 0 : (h == H - 1 ? 2 : 1))) * 3 + (w == 0 ? 0 : (w == W - 1 ? 2 : 1));
            }
            const float* const bp = tail.bias27 + (int64_t)cls * Cout + it.co0 + 4 * hk;
#pragma unroll
            for (int c = 0; c < NT; ++c)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 b4 = *reinterpret_cast<const float4*>(bp + c * 32 + 8 * gq);
                    acc[j][c][4 * gq] = b4.x;
                    acc[j][c][4 * gq + 1] = b4.y;
                    acc[j][c][4 * gq + 2] = b4.z;
                    acc[j][c][4 * gq + 3] = b4.w;
                }
        }
        }
    };
    // lane r of a column tile: h-row r>>4, w rotated by HW mod 16 on the second row (conflict-free ds_read_b128 groups, see k_conv_fwd_mfma)
    // B-fragment (halo) addresses of this lane: [k-step][kw (+ the parity's first kw)] for column tile 0 of this wave, filter row (0, 0), in
    // the halo slot the NEXT request goes to (the slot's base is added / subtracted once per item); everything else is an immediate
    static_assert(JT == 4, "tile_d(JT * cw + j) = cw, tile_h = 2 j + (r >> 4)");
    // (S16: pre[0][kw] only - voxel w16 of h-row 0 of the wave's d-plane, slot kq: the chunk's 32 channels are one k-step)
    int pre[2][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int hwc = (S16 ? w16 : lane_w(r)) + k;
        pre[0][k] = S16 ? (cw * HH * HW + hwc) * 64 + ((kq ^ halo_key(hwc)) << 4) : ((cw * HH + (r >> 4)) * HW + hwc) * 64 + ((hk ^ halo_key(hwc)) << 4);
        pre[1][k] = pre[0][k] ^ 32;
    }
    const int fa[2] = {S16 ? swz64(w16, kq) : swz64(r, hk), swz64(r, hk) ^ 32};
    if ((RES || HAS_RESID) && tail.bias27) init_acc_b27(cur);
    else {
        float4 bv0[NT][4];
        load_bias(cur.co0, bv0);
        init_acc(bv0);
    }
#ifdef FMRI_PROF
    unsigned long long cprof[12] = {};
    PROF_T(ck0);
#endif
    // double-buffered filter / halo fragments (live across the phases of a tile).  S16: a halo fragment serves CN consecutive MFMAs and is
    // dead after them - the next step's is requested into the same registers (fb_[0] only)
    // (S16, NBR = 4: the halo fragments run FOUR 4-MFMA blocks - 256 cycles of matrix pipe, the look-ahead of the 32x32x16 form - ahead
    // of their use in a ring of four: a whole step's eight fragments held 16 registers more than the 64-wide instantiations have)
    constexpr int NBR = 4;
    bf16x8_t fa_[2][CN], fb_[2][S16 ? NBR : JN];
    // RES: the residual's 16-byte lines this lane adds in the epilogue, [column tile][store instruction] (see the RES epilogue)
    constexpr int RKK = RES ? 32 / (64 / (BN / 8)) : 1;
    // column tiles whose lines are requested a phase early: none - 16 registers per tile at BN = 64 held across the last phase's MFMAs made
    // hipcc spill inside the matrix loop (tools/spill_sites.py); all lines are requested at the top of the epilogue, in front of its barrier
    constexpr int RPRE = 0;
    constexpr int RAHEAD = 2;                  // column tiles whose lines are in flight ahead of the one being finished (16 registers each)
    uint4 res_q[RES ? RAHEAD : 1][RKK];
    auto res_line = [&](int64_t org, int j, int kk) {
        constexpr int LPV = BN / 8, VPI = 64 / LPV;
        const int rt = JT * cw + j, rr = kk * VPI + lane / LPV, q8 = lane % LPV;
        return *reinterpret_cast<const uint4*>(residual + org + ((tile_d(rt) * H + tile_h(rt, rr)) * W + lane_w(rr)) * Cout + q8 * 8);
    };
    // The consumers' stream: item_next() works out the item after `cur`, run_item() is the phases of `cur`, tile_epilogue() what follows a
    // tile's last chunk.  The 32x32x16 form drives them as ONE flat loop over items (rounds 1-4).  S16 drives them as a loop over tiles
    // around a loop over the tile's chunks: in the flat form hipcc (ROCm 7.2) failed to coalesce the accumulator phis of the conditional
    // epilogue for the 32 four-register accumulators - three copies of all of them alive at the merge, 545 spills at BN = 64.
    bool has_next = true;
    FwdItem nxt = cur;
    int npair = pair;
    auto item_next = [&]() {
        has_next = true;
        nxt = cur;
        npair = pair;
        if (cur.ch + 1 < nch) nxt.ch = cur.ch + 1;
        else {
            npair = pair + gridDim.x;
            has_next = npair < npairs;
            if (has_next) nxt = decode(npair, 0);
        }
    };
    auto run_item = [&]() {
#pragma unroll
        for (int k = 0; k < 3; ++k) asm volatile("" : "+v"(pre[0][k]), "+v"(pre[1][k]));
        // (halo slot, halo row offset of the (kd,kh) row, first kw) of phase `pl` of item `it` whose halo sits in slot `slot`
        auto phase_hoff = [&](const FwdItem& it, int pl, int& kw0) {
            kw0 = 0;
            if constexpr (!PAR) return (((PH0 + pl) / 3) * HH + ((PH0 + pl) % 3)) * HW;
            else {
                if constexpr (TIGHT) return pl * HH * HW;        // phase pl = filter rows (kd' = pl, kh' = 0, 1); the parity is in the box's origin
                const int p = MODE == 1 ? it.par : (NPAR - 1) - it.ch / kpc;
                kw0 = p & 1;
                if constexpr (PL) return (HH + pl + ((p >> 1) & 1)) * HW;
                else return (((pl >> 1) + (p >> 2)) * HH + ((pl & 1) + ((p >> 1) & 1))) * HW;
            }
        };
        constexpr int NST = RPP * NKW * (S16 ? 1 : 2);         // steps of a phase: (filter row, kw, k-step); S16: (filter row, kw)
        // One wave per SIMD feeds the MFMA pipe alone: the fragments of step st+1 (a (kw, k-step) pair) are requested before the MFMAs of
        // step st are issued, into the other half of a double register set, threaded between those MFMAs (left to the compiler the reads
        // sat right in front of their MFMAs).  The pipeline runs ACROSS the phase barrier inside a tile: once the fragments of a phase's
        // last step are in registers this wave is done reading the rings, so it passes the next phase's barrier BEFORE issuing that step's
        // MFMAs and requests the next phase's first fragments under them - only the first phase of a tile starts with an exposed LDS latency.
        auto load_a = [&](const unsigned char* lfp, int st, int buf) {
            const int t = S16 ? st : st >> 1, ks = S16 ? 0 : st & 1;          // t = filter row of the phase * NKW + kw: the slab holds them in this order
#pragma unroll
            for (int c = 0; c < CN; ++c) fa_[buf][c] = *reinterpret_cast<const bf16x8_t*>(lfp + fa[ks] + (t * BN + c * (S16 ? 16 : 32)) * 64);
        };
        auto load_b = [&](int hoffp, int kw0p, int st, int buf, int j) {
            const int t = S16 ? st : st >> 1, ks = S16 ? 0 : st & 1;
            const int kw = t % NKW, row = t / NKW;       // (row > 0 only with RPP = 2: the phase's second filter row = the next h-row of the halo)
            int base;
            if constexpr (PAR && !TIGHT) base = kw0p ? pre[ks][kw + 1] : pre[ks][kw];
            else base = pre[ks][kw];
            // a 32-voxel column tile is two h-rows (the lane's row is in `pre`), an S16 fragment is one
            fb_[S16 ? 0 : buf][S16 ? j % NBR : j] = *reinterpret_cast<const bf16x8_t*>(lds + base + (hoffp + (row + (S16 ? j : 2 * j)) * HW) * 64);
        };
#pragma unroll
        for (int pl = 0; pl < NPH; ++pl, ++g) {
            PROF_T(c0);
            const unsigned char* const lf = lds + HALO_SPAN + (g & 1) * FILT_BYTES;
            int kw0;
            const int hoff = phase_hoff(cur, pl, kw0);
            if (pl == 0 && cur.ch == 0) {          // first phase of a tile: nothing was primed across the epilogue
                __builtin_amdgcn_s_barrier();
                load_a(lf, 0, 0);
#pragma unroll
                for (int j = 0; j < (S16 ? NBR : JN); ++j) load_b(hoff, kw0, 0, 0, j);
            }
            PROF_T(c1);
            if constexpr (RES) {
                if (pl == NPH - 1 && cur.ch == nch - 1) {          // the tile's last phase: request the residual lines of the first column tile(s)
                    const int64_t org = ((((int64_t)cur.n * D + cur.d0) * H + cur.h0) * W + cur.w0) * Cout + cur.co0;
#pragma unroll
                    for (int j = 0; j < RPRE; ++j)
#pragma unroll
                        for (int kk = 0; kk < RKK; ++kk) res_q[j & (RAHEAD - 1)][kk] = res_line(org, j, kk);
                }
            }
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                const bool last = st + 1 == NST;
                // a next phase of the SAME tile follows (same chunk, or the next chunk of this tile in the other halo slot)
                const bool chain = last && (pl + 1 < NPH || cur.ch + 1 < nch);
                const unsigned char* lfn = lf;
                int hoffn = hoff, kw0n = kw0, stn = st + 1;
                if (last) {
                    stn = 0;
                    lfn = lds + HALO_SPAN + ((g + 1) & 1) * FILT_BYTES;
                    if (pl + 1 < NPH) hoffn = phase_hoff(cur, pl + 1, kw0n);
                    else {
                        // every later request is for the next item, whose halo sits in the other slot: move the six addresses there
                        hoffn = phase_hoff(nxt, 0, kw0n);
                        if constexpr (!S16) {
                            const int dlt = hb ? -HALO_STRIDE : HALO_STRIDE;
#pragma unroll
                            for (int k = 0; k < 3; ++k) { pre[0][k] += dlt; pre[1][k] += dlt; }
                        }
                    }
                }
                // the hand-over to the next phase: this phase's last fragments are in registers = this wave is done with the rings, so it
                // passes the next phase's barrier here, in front of the MFMAs that still use them
                auto hand_over = [&]() {
                    if (chain) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();                          // = the next phase's barrier
                    }
                };
                if constexpr (!S16) hand_over();
                // The two filter-fragment buffers alternate per step through the whole chunk (step pl * NST + st; a chunk's first step uses
                // buffer 0).  The plain S16 form has 27 steps per chunk: its last step sits in buffer 0 as well, so the next chunk's first
                // filter fragments are requested AFTER that step's MFMAs have been issued (one exposed LDS latency per chunk, ~1 %; a third
                // buffer costs 16 registers the 64-wide instantiations do not have: 41 spills, some inside this loop).
                const int ab = (pl * NST + st) & 1;
                const bool late = (NPH * NST) % 2 == 1 && pl == NPH - 1 && last;
                const int abn = (last && pl == NPH - 1) ? 0 : ab ^ 1;
                // (S16, last step: the next phase's filter fragments are requested behind the hand-over, inside the block loop)
                if ((!last || (chain && !S16)) && !late) load_a(lfn, stn, abn);
#pragma unroll
                for (int j = 0; j < JN; ++j) {
#pragma unroll
                    for (int c = 0; c < CN; ++c) {
                        if constexpr (S16) acc[j][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_[ab][c], fb_[0][j % NBR], acc[j][c], 0, 0, 0);
                        else if constexpr (F32) {
                            // the 16 bytes of a fragment are four fp32 k-values: lanes 0-31 hold channels 8 ks + i, lanes 32-63 channels
                            // 8 ks + 4 + i of the same row - the two k of MFMA i, in both operands alike
                            const f32x4 af = __builtin_bit_cast(f32x4, fa_[ab][c]), bf = __builtin_bit_cast(f32x4, fb_[ab][j]);
#pragma unroll
                            for (int i = 0; i < 4; ++i) acc[j][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[i], acc[j][c], 0, 0, 0);
                        } else acc[j][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[ab][c], fb_[ab][j], acc[j][c], 0, 0, 0);
                    }
                    if constexpr (S16) {
                        // the fragment NBR blocks on takes this one's place: of this step while it has them, else of the next one
                        if (j + NBR < JN) load_b(hoff, kw0, st, 0, j + NBR);
                        else {
                            if (last && j + NBR == JN) {
                                // a phase's last step: its last request into the rings went out a block ago.  (The chunk's last phase: the next
                                // item's halo sits in the other slot - the three addresses move there now, not before: the requests of
                                // blocks 0 .. JN - NBR - 1 still went to this chunk's slot, which the barrier below hands to the producers.)
                                hand_over();
                                if (pl + 1 == NPH) {
                                    const int dlt = hb ? -HALO_STRIDE : HALO_STRIDE;
#pragma unroll
                                    for (int k = 0; k < 3; ++k) pre[0][k] += dlt;
                                }
                                if (chain && !late) load_a(lfn, stn, abn);
                            }
                            if (!last || chain) load_b(hoffn, kw0n, stn, 0, j + NBR - JN);
                        }
                    } else if (!last || chain) load_b(hoffn, kw0n, stn, abn, j);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (late && chain) load_a(lfn, stn, abn);
            }
            PROF_T(c2);
#ifdef FMRI_PROF
            cprof[1] += c1 - c0; cprof[3] += c2 - c1; cprof[6] += 1;
#endif
        }
    };
    auto tile_epilogue = [&]() {
        PROF_T(ce0);
        if constexpr (RES) {
            // y = act(acc + residual): fp32 transposition through LDS, one 32-voxel column tile at a time (wave-private 8 KiB).  The
            // residual lines of the first RPRE column tiles were requested at the top of the tile's last phase (res_q), the next RAHEAD tiles' lines are
            // requested here, where the fragment registers have died, and each finished line's registers take the line RAHEAD tiles on: round 2 loaded each tile's lines inside its own iteration and waited four
            // global-memory latencies per tile in a row (MFMA pipe 43 % busy on the dec0a launch).
            constexpr int PPV = BN / 4, LPV = BN / 8, VPI = 64 / LPV;
            float4 bvn[NT][4];
            load_bias(has_next ? nxt.co0 : cur.co0, bvn);
            const int64_t org = ((((int64_t)cur.n * D + cur.d0) * H + cur.h0) * W + cur.w0) * Cout + cur.co0;
#pragma unroll
            for (int j = RPRE; j < RAHEAD; ++j)
#pragma unroll
                for (int kk = 0; kk < RKK; ++kk) res_q[j & (RAHEAD - 1)][kk] = res_line(org, j, kk);
            __builtin_amdgcn_s_barrier();
            unsigned char* const stage = lds + hb * HALO_BYTES + cw * (32 * BN * 4);
#pragma unroll
            for (int j = 0; j < JT; ++j) {
                if constexpr (S16) {
                    // column tile j = h-rows 2 j and 2 j + 1 of the plane; the lane's voxel sits at index rr of the tile (second row rotated, lane_w)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int rr = jj ? 16 + ((w16 + (HW & 15)) & 15) : w16;
#pragma unroll
                        for (int c = 0; c < CN; ++c) {
                            const int q = c * 4 + kq;
                            *reinterpret_cast<float4*>(stage + rr * (BN * 4) + (((q ^ rr) & (PPV - 1)) << 4)) =
                                make_float4(acc[2 * j + jj][c][0], acc[2 * j + jj][c][1], acc[2 * j + jj][c][2], acc[2 * j + jj][c][3]);
                        }
                    }
                } else {
#pragma unroll
                for (int c = 0; c < NT; ++c)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int q = c * 8 + 2 * gq + hk;
                        *reinterpret_cast<float4*>(stage + r * (BN * 4) + (((q ^ r) & (PPV - 1)) << 4)) =
                            make_float4(acc[j][c][4 * gq], acc[j][c][4 * gq + 1], acc[j][c][4 * gq + 2], acc[j][c][4 * gq + 3]);
                    }
                }
                const int rt = JT * cw + j;
#pragma unroll
                for (int kk = 0; kk < 32 / VPI; ++kk) {
                    const int rr = kk * VPI + lane / LPV, q8 = lane % LPV;
                    const float4 a0 = *reinterpret_cast<const float4*>(stage + rr * (BN * 4) + ((((2 * q8) ^ rr) & (PPV - 1)) << 4));
                    const float4 a1 = *reinterpret_cast<const float4*>(stage + rr * (BN * 4) + ((((2 * q8 + 1) ^ rr) & (PPV - 1)) << 4));
                    const int64_t ao = org + ((tile_d(rt) * H + tile_h(rt, rr)) * W + lane_w(rr)) * Cout + q8 * 8;
                    const uint4 r4 = res_q[j & (RAHEAD - 1)][kk];
                    // (the partial sum is rounded to bf16 before the residual is added - what the asynchronous form, EPI 3 of k_conv_fwd_ws,
                    // stages: every form of this launch gives the same bits, whichever one the grid size selects)
                    const unsigned p0 = pack2bf(a0.x, a0.y), p1 = pack2bf(a0.z, a0.w), p2 = pack2bf(a1.x, a1.y), p3 = pack2bf(a1.z, a1.w);
                    float o[8] = {__uint_as_float(p0 << 16) + __uint_as_float(r4.x << 16), __uint_as_float(p0 & 0xffff0000u) + __uint_as_float(r4.x & 0xffff0000u),
                                  __uint_as_float(p1 << 16) + __uint_as_float(r4.y << 16), __uint_as_float(p1 & 0xffff0000u) + __uint_as_float(r4.y & 0xffff0000u),
                                  __uint_as_float(p2 << 16) + __uint_as_float(r4.z << 16), __uint_as_float(p2 & 0xffff0000u) + __uint_as_float(r4.z & 0xffff0000u),
                                  __uint_as_float(p3 << 16) + __uint_as_float(r4.w << 16), __uint_as_float(p3 & 0xffff0000u) + __uint_as_float(r4.w & 0xffff0000u)};
#pragma unroll
                    for (int i = 0; i < 8; ++i) o[i] = vmax(o[i], __builtin_fmaf(o[i], act_s, 0.f));
                    *reinterpret_cast<uint4*>(y + ao) = make_uint4(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7]));
                    // this tile's line is consumed: request the one RAHEAD tiles on into the same registers
                    if (j + RAHEAD < JT) res_q[j & (RAHEAD - 1)][kk] = res_line(org, j + RAHEAD, kk);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (tail.bias27) init_acc_b27(has_next ? nxt : cur);
            else init_acc(bvn);
        }
        if constexpr (!RES) {
            // bias is in the accumulators; activation, bf16, half-wave exchange, wave-private LDS transposition (16 KiB per consumer wave of
            // the consumed halo slot), line-major stores
            constexpr int CPV = BN / 8;                  // 16-byte pieces per voxel
            constexpr int VPI = 64 / CPV;                // voxels per store instruction
            constexpr int SWM = BNB == 64 ? 7 : 3;
            float4 bvn[NT][4];
            load_bias(has_next ? nxt.co0 : cur.co0, bvn);
            PROF_T(e0);
            __builtin_amdgcn_s_barrier();
            PROF_T(e1);
            unsigned char* const stage = lds + hb * HALO_BYTES + cw * (32 * JT * BNB * 2);
            // the activation is a launch constant: none (every input-gradient launch) costs nothing, ReLU one v_max per value, LeakyReLU two
            // operations - the general max(v, alpha * v) form for all three spent 256 VALU instructions per tile and wave on the identity
            auto stage_tile = [&](auto act_tag) {
                constexpr int ACT = decltype(act_tag)::value;
                unsigned zero2_ = 0u;
                asm volatile("" : "+v"(zero2_));
                if constexpr (S16) {
                    // a lane holds 4 channels (two dwords of bf16) of its voxel per 16-channel block; v_permlane16_swap of blocks 2 p and 2 p + 1
                    // (rows 1 <-> 0 and 3 <-> 2 of the two operands) leaves every lane with one 16-byte piece: lane row kq gets piece
                    // 4 p + 2 (kq & 1) + (kq >> 1) of the voxel's BN / 8
#pragma unroll
                    for (int j = 0; j < JN; ++j) {
                        const int v = (j >> 1) * 32 + ((j & 1) ? 16 + ((w16 + (HW & 15)) & 15) : w16);      // place in the staged tile (lane_w's rotation on odd rows)
                        const int vs = NT == 2 ? (v & 7) : ((v >> 2) & 3);
#pragma unroll
                        for (int p = 0; p < NT; ++p) {
                            unsigned pk[2][2];
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                float o[4];
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const float vv = acc[j][2 * p + u][i];
                                    if constexpr (ACT == FMRI_ACT_LEAKY) o[i] = vmax(vv, __builtin_fmaf(vv, act_s, 0.f));
                                    else o[i] = vv;
                                }
                                pk[u][0] = pack2bf(o[0], o[1]);
                                pk[u][1] = pack2bf(o[2], o[3]);
                                if constexpr (ACT == FMRI_ACT_RELU) {
                                    pk[u][0] = relu2(pk[u][0], zero2_);
                                    pk[u][1] = relu2(pk[u][1], zero2_);
                                }
                            }
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                auto sw = __builtin_amdgcn_permlane16_swap(pk[0][q], pk[1][q], false, false);
                                pk[0][q] = sw[0];
                                pk[1][q] = sw[1];
                            }
                            const int q = p * 4 + ((kq & 1) << 1) + (kq >> 1);
                            *reinterpret_cast<uint4*>(stage + v * (BN * 2) + (((q ^ vs) & SWM) << 4)) = make_uint4(pk[0][0], pk[0][1], pk[1][0], pk[1][1]);
                        }
                    }
                } else if constexpr (F32) {
                    // a lane's registers 4 gq .. 4 gq + 3 are channels 8 gq + 4 hk .. + 3 of its voxel: one 16-byte piece (number 2 gq + hk of the
                    // voxel's eight) as they are - no packing, no half-wave exchange
#pragma unroll
                    for (int j = 0; j < JT; ++j) {
                        const int v = j * 32 + r;
                        const int vs = v & 7;
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            float o[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float vv = acc[j][0][4 * gq + i];
                                if constexpr (ACT == FMRI_ACT_LEAKY) o[i] = vmax(vv, __builtin_fmaf(vv, act_s, 0.f));
                                else if constexpr (ACT == FMRI_ACT_RELU) o[i] = vmax(vv, 0.f);
                                else o[i] = vv;
                            }
                            const int q = 2 * gq + hk;
                            *reinterpret_cast<float4*>(stage + v * (BNB * 2) + (((q ^ vs) & SWM) << 4)) = make_float4(o[0], o[1], o[2], o[3]);
                        }
                    }
                } else
#pragma unroll
                for (int j = 0; j < JT; ++j) {
                    const int v = j * 32 + r;
                    const int vs = NT == 2 ? (v & 7) : ((v >> 2) & 3);
#pragma unroll
                    for (int c = 0; c < NT; ++c) {
#pragma unroll
                        for (int pq = 0; pq < 2; ++pq) {
                            unsigned pk[2][2];
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                const int gq = 2 * pq + u;
                                float o[4];
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const float vv = acc[j][c][4 * gq + i];
                                    if constexpr (ACT == FMRI_ACT_LEAKY) o[i] = vmax(vv, __builtin_fmaf(vv, act_s, 0.f));
                                    else o[i] = vv;
                                }
                                pk[u][0] = pack2bf(o[0], o[1]);
                                pk[u][1] = pack2bf(o[2], o[3]);
                                if constexpr (ACT == FMRI_ACT_RELU) {
                                    // ReLU on the packed pair: max(int16(bits), 0) per half - a bf16 is negative (or -0) exactly when its bits are
                                    // a negative int16, and rounding never changes the sign: the same bits as bf16(max(v, 0)), half the instructions
                                    pk[u][0] = relu2(pk[u][0], zero2_);
                                    pk[u][1] = relu2(pk[u][1], zero2_);
                                }
                            }
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                auto sw = __builtin_amdgcn_permlane32_swap(pk[0][q], pk[1][q], false, false);
                                pk[0][q] = sw[0];
                                pk[1][q] = sw[1];
                            }
                            const int q = c * 4 + pq * 2 + hk;
                            *reinterpret_cast<uint4*>(stage + v * (BN * 2) + (((q ^ vs) & SWM) << 4)) =
                                make_uint4(pk[0][0], pk[0][1], pk[1][0], pk[1][1]);
                        }
                    }
                }
            };
            if (HAS_RESID || HAS_NBWD || act == FMRI_ACT_NONE) stage_tile(std::integral_constant<int, FMRI_ACT_NONE>{});     // (EPI 3 / 5 / 6: `act` is the producers')
            else if (act == FMRI_ACT_RELU) stage_tile(std::integral_constant<int, FMRI_ACT_RELU>{});
            else stage_tile(std::integral_constant<int, FMRI_ACT_LEAKY>{});
            PROF_T(e2);
            if (HAS_RESID && tail.bias27) init_acc_b27(has_next ? nxt : cur);
            else init_acc(bvn);
            if constexpr (ASY) {
                // the producers store the staged tile under the next tile's first phases; the "staged" barrier is that tile's phase-0 barrier
                if (!has_next) __builtin_amdgcn_s_barrier();           // (the workgroup's last tile: the producers store it right away)
            } else {
                __builtin_amdgcn_s_barrier();                          // the whole tile is staged: all eight waves store it
            }
            PROF_T(e3);
            if constexpr (!ASY) store_share(cur, hb, wv, std::integral_constant<int, 8>{}, std::false_type{}, nullptr, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
            PROF_T(e4);
#ifdef FMRI_PROF
            cprof[7] += e1 - e0; cprof[8] += e2 - e1; cprof[9] += e3 - e2; cprof[10] += e4 - e3;
#endif
        }
#ifdef FMRI_PROF
        { PROF_T(ce1); cprof[4] += ce1 - ce0; }
#endif
    };
    if constexpr (S16) {
        while (true) {
            while (true) {
                item_next();
                run_item();
                if (cur.ch == nch - 1) break;
                cur = nxt;
                hb ^= 1;
            }
            tile_epilogue();
            if (!has_next) break;
            cur = nxt;
            pair = npair;
            hb ^= 1;
        }
    } else {
        while (true) {
            item_next();
            run_item();
            if (cur.ch == nch - 1) tile_epilogue();
            if (!has_next) break;
            cur = nxt;
            pair = npair;
            hb ^= 1;
        }
    }
#ifdef FMRI_PROF
    PROF_T(ck1);
    cprof[5] = ck1 - ck0;
    if (lane == 0)
        for (int i = 0; i < 12; ++i) atomicAdd(&g_prof[i], cprof[i]);
#endif
}

}  // namespace

// --------------------------------------------------------------------------------------------------------- host dispatch
static int fwd_f32_mfma() {             // FMRI_F32_MFMA=0: fp32 tensors stay on the VALU kernels of conv3d_generic.hip (rounds 1-5)
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("FMRI_F32_MFMA");
        v = e ? atoi(e) : 1;
    }
    return v;
}
bool conv3d_fwd_mfma_ok(int C0, int C1, int Cout, int D, int H, int W, int dtype) {
    if (dtype == FMRI_F32) {
        // fp32 on v_mfma_f32_32x32x2_f32 (k_conv_fwd_ws<..., F32>): 16-channel chunks (64-byte rows), 32-wide Cout blocks, the 4x8x16 tiling
        if (!fwd_f32_mfma() || (C0 % 16) || (C1 % 16) || C0 + C1 < 16 || (Cout % 32)) return false;
        if ((D % fw::TD) || (H % fw::TH) || (W % fw::TW)) return false;
        C0 *= 2;                                         // everything below counts 2-byte units
        C1 *= 2;
    } else if (dtype != FMRI_BF16) return false;
    if ((C0 % 32) || (C1 % 32) || C0 + C1 < 32 || (Cout % 32)) return false;
    if (C0 + C1 > 4096) return false;                    // zero-page length (see g_zero_page)
    // buffer-descriptor LDS-DMA (dma16_buf): a lane's 32-bit byte offset from the item's box corner must stay below DMA_OOB = 2^31, or a
    // legitimate row would be zero-filled silently.  Largest offset: (tile depth + 2) planes, strides doubled in the space-to-depth mode.
    {
        const long long planes = (D < fw::TD ? D : fw::TD) + 2, cmax = C0 > C1 ? C0 : C1;
        if (planes * 2 * H * W * cmax * 2 >= (1ll << 31)) return false;
    }
    if ((D % fw::TD) || (H % fw::TH) || (W % fw::TW)) return !((D % 8) || (H % 8) || (W % 8));     // 8x8x8 tiles (CUBE variant)
    return true;
}
// true when only the 8x8x8 tiling fits: plain 3-D convs only (no planar, parity-form or residual launches)
bool conv3d_fwd_needs_cube(int D, int H, int W) { return (D % fw::TD) || (H % fw::TH) || (W % fw::TW); }

// mode 0: plain conv (residual != nullptr selects the RES epilogue); 1: up-forward (src0 = LOW-res tensor, D/H/W = low-res dims,
// y = [2D][2H][2W] partial sums); 2: up-backward (src0 = dy [2D][2H][2W][C0], y = gradient of the low-res tensor)
static int fwd_use_ws() {               // FMRI_FWD_WS=0: the symmetric kernel (every wave issues DMA and MFMAs) instead of the warp-specialised one
    static int use_ws = -1;
    if (use_ws < 0) {
        const char* e = getenv("FMRI_FWD_WS");
        use_ws = e ? atoi(e) : 1;
    }
    return use_ws;
}
static int fwd_fast_halo() {            // FMRI_FAST_HALO=0: every halo piece's address worked out per lane as in rounds 1-3 (A/B)
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("FMRI_FAST_HALO");
        v = e ? atoi(e) : 1;
    }
    return v;
}
static int fwd_async() {                // FMRI_FWD_ASYNC=0: the tile's stores by all eight waves behind a barrier (the round-2 epilogue)
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("FMRI_FWD_ASYNC");
        v = e ? atoi(e) : 1;
    }
    return v;
}
static int fwd_mfma16() {               // FMRI_MFMA16=0: v_mfma_f32_32x32x16_bf16 in the warp-specialised forward kernel (rounds 1-4), 1: 16x16x32 (S16)
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("FMRI_MFMA16");
        v = e ? atoi(e) : 0;
    }
    return v;
}
static bool fwd_wide(int mode, int planar, int ntile, int Cout) {
    // 64-wide Cout blocks halve the halo traffic per MFMA, but a launch with fewer (tile, block) pairs than CUs (the 8x16x16 bottleneck
    // level) leaves CUs idle: 32-wide blocks double the pairs there
    return Cout % 64 == 0 && (int64_t)ntile * (Cout / 64) * (mode == 1 ? (planar ? 4 : 8) : 1) >= fwd_cu_count();
}

// fp32 launches: 32-wide Cout blocks, asynchronous epilogue throughout (plain store / pooled copy, ReLU mask, residual), parity modes on the
// tight box; channel counts that are memory strides go in as 2-byte units (see k_conv_fwd_ws, F32)
static int conv3d_fwd_mfma_launch_f32(int mode, const void* src0, int C0, int up0, int planar, const void* src1, int C1, const void* w,
                                      const float* bias, const void* mask, const void* residual, void* y, int N, int D, int H, int W, int Cout,
                                      int act, float alpha, FwdTail tail, hipStream_t st) {
    if (planar || (D % fw::TD) || (H % fw::TH) || (W % fw::TW) || tail.logits || tail.nws) return FMRI_E_SHAPE;
    if (mode != 0 && (C1 != 0 || up0)) return FMRI_E_SHAPE;
    SrcB s{(const bf16_t*)src0, (const bf16_t*)src1, 2 * C0, 2 * C1, up0, 1, 0};
    const int ntile = N * (D / fw::TD) * (H / fw::TH) * (W / fw::TW);
    const int ncu = fwd_cu_count();
    const int np = ntile * (Cout / 32) * (mode == 1 ? 8 : 1);
    const int grid = np < ncu ? np : ncu;
    const bool fh = fwd_fast_halo() && C1 == 0 && !up0;
    tail.prio = 0;
#define FMRI_F32K(MODE_, A_, EPI_, FH_)                                                                                   \
    k_conv_fwd_ws<1, false, MODE_, false, A_, EPI_, FH_, false, true><<<grid, fw::NTHREADS, 0, st>>>(                     \
        s, (const bf16_t*)w, bias, (const bf16_t*)mask, (const bf16_t*)residual, (bf16_t*)y, N, D, H, W, Cout, act, alpha, tail)
    if (mode == 1) {
        if (mask || residual || tail.pool) return FMRI_E_SHAPE;
        FMRI_F32K(1, false, -1, true);
    } else if (mode == 2) {
        if (residual || tail.pool) return FMRI_E_SHAPE;
        FMRI_F32K(2, false, -1, true);
    } else if (residual) {
        if (mask || tail.pool) return FMRI_E_SHAPE;
        if (fh) FMRI_F32K(0, true, 3, true); else FMRI_F32K(0, true, 3, false);
    } else if (mask) {
        if (tail.pool) return FMRI_E_SHAPE;
        if (fh) FMRI_F32K(0, true, 1, true); else FMRI_F32K(0, true, 1, false);
    } else {
        if (fh) FMRI_F32K(0, true, 0, true); else FMRI_F32K(0, true, 0, false);
    }
#undef FMRI_F32K
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

static int conv3d_fwd_mfma_launch(int mode, const void* src0, int C0, int up0, int planar, const void* src1, int C1, const void* w,
                                  const float* bias, const void* mask, const void* residual, void* y, int N, int D, int H, int W, int Cout,
                                  int act, float alpha, FwdTail tail, int dtype, hipStream_t st) {
    if (dtype == FMRI_F32)
        return conv3d_fwd_mfma_launch_f32(mode, src0, C0, up0, planar, src1, C1, w, bias, mask, residual, y, N, D, H, W, Cout, act, alpha, tail, st);
    SrcB s{(const bf16_t*)src0, (const bf16_t*)src1, C0, C1, up0, planar ? 0 : 1, planar};
    // the parity modes form a fresh halo pointer as [one plain source] + box corner + lane constant (CHEAP_FRESH / FH): a second source or a
    // fused up-sampling there would read wrong addresses instead of failing
    if (mode != 0 && (C1 != 0 || up0)) return FMRI_E_SHAPE;
    const bool cube = (D % fw::TD) || (H % fw::TH) || (W % fw::TW);       // only the 8x8x8 tiling fits (conv3d_fwd_mfma_ok)
    const int ntile = cube ? N * (D / 8) * (H / 8) * (W / 8) : N * (D / fw::TD) * (H / fw::TH) * (W / fw::TW);
    const int ncu = fwd_cu_count();
    const int use_ws = fwd_use_ws();
    // fast halo addressing (k_conv_fwd_ws<..., FH = true>): single plain source, at most two 32-channel chunks per freshly addressed halo
    // (round 5: the warp-specialised kernel's form goes through a buffer descriptor and fixes a tile's out-of-volume offsets up once per
    // tile: it serves every single-source 3-D launch (same-box A/B, profiles/r05_buffer_halo_ab.log: forward family -1.9 %, step +0.9 %;
    // FMRI_FH_MAXCH = the most chunks per fresh address for which it is taken, 2 = the round-4 rule)
    static int fh_maxch = -1;
    if (fh_maxch < 0) {
        const char* e = getenv("FMRI_FH_MAXCH");
        fh_maxch = e ? atoi(e) : 1 << 20;
    }
    const int nchunk_fresh = mode == 2 ? C0 / 32 : (C0 + C1) / 32;
    const bool fh_any = fwd_fast_halo() && !cube && C1 == 0 && !up0 && nchunk_fresh <= fh_maxch;
    const bool fh = fh_any && !planar;                 // (the warp-specialised kernel's form covers the 3-D launches)
    const bool s16 = fwd_mfma16() && !planar;
    {
        static int prio = -1;
        if (prio < 0) {
            const char* e = getenv("FMRI_FWD_PRIO");
            prio = e ? atoi(e) : 0;
        }
        tail.prio = prio;
    }
#define FMRI_WS2(NT_, PL_, MODE_, RES_, A_, EPI_, FH_, GRID_)                                                              \
    do {                                                                                                                  \
        if (s16 && !(PL_))                                                                                                \
            k_conv_fwd_ws<NT_, PL_, MODE_, RES_, A_, EPI_, FH_, !(PL_)><<<GRID_, fw::NTHREADS, 0, st>>>(                  \
                s, (const bf16_t*)w, bias, (const bf16_t*)mask, (const bf16_t*)residual, (bf16_t*)y, N, D, H, W, Cout, act, alpha, tail); \
        else                                                                                                              \
            k_conv_fwd_ws<NT_, PL_, MODE_, RES_, A_, EPI_, FH_, false><<<GRID_, fw::NTHREADS, 0, st>>>(                   \
                s, (const bf16_t*)w, bias, (const bf16_t*)mask, (const bf16_t*)residual, (bf16_t*)y, N, D, H, W, Cout, act, alpha, tail); \
    } while (0)
#define FMRI_WS(NT_, PL_, MODE_, RES_, A_, EPI_, GRID_)                                                                    \
    do {                                                                                                                  \
        if (fh && !(PL_)) FMRI_WS2(NT_, PL_, MODE_, RES_, A_, EPI_, !(PL_), GRID_);                                       \
        else FMRI_WS2(NT_, PL_, MODE_, RES_, A_, EPI_, false, GRID_);                                                     \
    } while (0)
#define FMRI_LAUNCH_FWD(NT_, PL_, MODE_, RES_)                                                                             \
    do {                                                                                                                  \
        const int np = ntile * (Cout / (32 * NT_)) * (MODE_ == 1 ? (PL_ ? 4 : 8) : 1);                                    \
        constexpr bool A_ = !(PL_) && !(RES_) && (MODE_) == 0;   /* asynchronous epilogue: plain 3-D launches */         \
        const int epi_ = mask ? ((tail.pool || tail.logits) ? -1 : 1) : (tail.logits ? (tail.pool ? -1 : 2) : 0);                  \
        if (use_ws && A_ && fwd_async() && np > ncu && epi_ >= 0) {   /* a single pair per workgroup has nothing to hide the stores under */ \
            if (epi_ == 0) FMRI_WS(NT_, PL_, MODE_, RES_, A_, (A_ ? 0 : -1), (np < ncu ? np : ncu));                              \
            else if (epi_ == 1) FMRI_WS(NT_, PL_, MODE_, RES_, A_, (A_ ? 1 : -1), (np < ncu ? np : ncu));                         \
            else FMRI_WS(NT_, PL_, MODE_, RES_, A_, (A_ ? 2 : -1), (np < ncu ? np : ncu));                                        \
        } else if (use_ws && (!(PL_) || use_ws > 1))   /* planar: the producers are the bottleneck - symmetric kernel */ \
            FMRI_WS(NT_, PL_, MODE_, RES_, false, -1, (np < ncu ? np : ncu));                                             \
        else if (fh_any && (MODE_) == 0)                                                                                  \
            k_conv_fwd_mfma<NT_, PL_, MODE_, RES_, false, (MODE_) == 0><<<np < ncu ? np : ncu, fw::NTHREADS, 0, st>>>(    \
                s, (const bf16_t*)w, bias, (const bf16_t*)mask, (const bf16_t*)residual, (bf16_t*)y, N, D, H, W, Cout, act, alpha, tail); \
        else                                                                                                              \
            k_conv_fwd_mfma<NT_, PL_, MODE_, RES_><<<np < ncu ? np : ncu, fw::NTHREADS, 0, st>>>(                         \
                s, (const bf16_t*)w, bias, (const bf16_t*)mask, (const bf16_t*)residual, (bf16_t*)y, N, D, H, W, Cout, act, alpha, tail); \
    } while (0)
    const bool wide = fwd_wide(mode, planar, ntile, Cout);
    if (tail.nws) {
        // normalisation tails (EPI 4: statistics of the output, 6: the same behind the residual, 5: the backward reductions): asynchronous
        // epilogue only - conv3d_fwd_ntail_ok() tells the caller beforehand
        const int np_ = ntile * (Cout / (wide ? 64 : 32));
        if (cube || mode != 0 || planar || !use_ws || !fwd_async() || np_ <= ncu || tail.pool || tail.logits || tail.bias27) return FMRI_E_SHAPE;
        if (tail.nss ? (!mask || residual) : (mask != nullptr)) return FMRI_E_SHAPE;
#define FMRI_LAUNCH_NT(NT_, EPI_) FMRI_WS2(NT_, false, 0, false, true, EPI_, false, ncu)
        if (tail.nss) { if (wide) FMRI_LAUNCH_NT(2, 5); else FMRI_LAUNCH_NT(1, 5); }
        else if (residual) { if (wide) FMRI_LAUNCH_NT(2, 6); else FMRI_LAUNCH_NT(1, 6); }
        else { if (wide) FMRI_LAUNCH_NT(2, 4); else FMRI_LAUNCH_NT(1, 4); }
#undef FMRI_LAUNCH_NT
        FMRI_LAUNCH_CHECK();
        return FMRI_OK;
    }
    if (planar && (tail.pool || tail.logits)) {
        // 2-D slices: the pooled copy / the final conv's logits come out of the symmetric kernel's epilogue (TAIL instantiations)
        if (cube || mode != 0 || residual || mask || C1 != 0 || up0 || (tail.logits && Cout != (wide ? 64 : 32))) return FMRI_E_SHAPE;
        const int np = ntile * (Cout / (wide ? 64 : 32));
#define FMRI_PT(NT_, FH_)                                                                                                  \
    k_conv_fwd_mfma<NT_, true, 0, false, false, FH_, true><<<np < ncu ? np : ncu, fw::NTHREADS, 0, st>>>(                  \
        s, (const bf16_t*)w, bias, nullptr, nullptr, (bf16_t*)y, N, D, H, W, Cout, act, alpha, tail)
        if (wide) { if (fh_any) FMRI_PT(2, true); else FMRI_PT(2, false); }
        else { if (fh_any) FMRI_PT(1, true); else FMRI_PT(1, false); }
#undef FMRI_PT
        FMRI_LAUNCH_CHECK();
        return FMRI_OK;
    }
    if (cube) {
        if (mode != 0 || residual || planar) return FMRI_E_SHAPE;
        const int nt = wide ? 2 : 1;
        const int np = ntile * (Cout / (32 * nt));
        if (wide)
            k_conv_fwd_mfma<2, false, 0, false, true><<<np < ncu ? np : ncu, fw::NTHREADS, 0, st>>>(
                s, (const bf16_t*)w, bias, (const bf16_t*)mask, nullptr, (bf16_t*)y, N, D, H, W, Cout, act, alpha, tail);
        else
            k_conv_fwd_mfma<1, false, 0, false, true><<<np < ncu ? np : ncu, fw::NTHREADS, 0, st>>>(
                s, (const bf16_t*)w, bias, (const bf16_t*)mask, nullptr, (bf16_t*)y, N, D, H, W, Cout, act, alpha, tail);
    } else if (mode == 1 && planar) {
        if (wide) FMRI_LAUNCH_FWD(2, true, 1, false); else FMRI_LAUNCH_FWD(1, true, 1, false);
    } else if (mode == 1) {
        if (wide) FMRI_LAUNCH_FWD(2, false, 1, false); else FMRI_LAUNCH_FWD(1, false, 1, false);
    } else if (mode == 2 && planar) {
        if (wide) FMRI_LAUNCH_FWD(2, true, 2, false); else FMRI_LAUNCH_FWD(1, true, 2, false);
    } else if (mode == 2) {
        if (wide) FMRI_LAUNCH_FWD(2, false, 2, false); else FMRI_LAUNCH_FWD(1, false, 2, false);
    } else if (residual && planar) {
        if (wide) FMRI_LAUNCH_FWD(2, true, 0, true); else FMRI_LAUNCH_FWD(1, true, 0, true);
    } else if (residual) {
        // the residual added by the producers in the asynchronous drain (EPI 3) - FMRI_RES_ASYNC=0 keeps the RES epilogue on the MFMA waves.
        // Both add bf16(partial sum) + residual in fp32: the same bits from either, so a result does not depend on the grid size
        static int res_async = -1;
        if (res_async < 0) {
            const char* e = getenv("FMRI_RES_ASYNC");
            res_async = e ? atoi(e) : 1;
        }
        const int np_ = ntile * (Cout / (wide ? 64 : 32));
        if (res_async && use_ws && fwd_async() && np_ > ncu && !mask && !tail.pool && !tail.logits) {
            if (wide) FMRI_WS(2, false, 0, false, true, 3, ncu);
            else FMRI_WS(1, false, 0, false, true, 3, ncu);
        } else if (wide) FMRI_LAUNCH_FWD(2, false, 0, true); else FMRI_LAUNCH_FWD(1, false, 0, true);
    } else if (wide) {
        if (planar) FMRI_LAUNCH_FWD(2, true, 0, false); else FMRI_LAUNCH_FWD(2, false, 0, false);
    } else {
        if (planar) FMRI_LAUNCH_FWD(1, true, 0, false); else FMRI_LAUNCH_FWD(1, false, 0, false);
    }
#undef FMRI_LAUNCH_FWD
#undef FMRI_WS
#undef FMRI_WS2
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
int conv3d_fwd_mfma_ex(int mode, const void* src0, int C0, int up0, int planar, const void* src1, int C1, const void* w, const float* bias,
                       const void* mask, const void* residual, void* y, int N, int D, int H, int W, int Cout, int act, float alpha, int dtype,
                       hipStream_t st) {
    return conv3d_fwd_mfma_launch(mode, src0, C0, up0, planar, src1, C1, w, bias, mask, residual, y, N, D, H, W, Cout, act, alpha,
                                  FwdTail{nullptr, nullptr, nullptr, nullptr, nullptr}, dtype, st);
}
// bit 0: the 2x2x2 max-pooled copy can be produced by the conv's epilogue, bit 1: the final 1x1x1 conv to one label can (plain 3-D
// warp-specialised launch on the 4x8x16 tiling; the logits need the voxel's whole channel range in one workgroup: Cout == block width)
// planar: D slices of H x W, the pooled copy is MaxPooling2D(2) per slice (the symmetric kernel's TAIL instantiations)
int conv3d_fwd_tail_ok(int C0, int Cout, int N, int D, int H, int W, int dtype, int planar) {
    if (!conv3d_fwd_mfma_ok(C0, 0, Cout, D, H, W, dtype) || conv3d_fwd_needs_cube(D, H, W)) return 0;
    const int ntile = N * (D / fw::TD) * (H / fw::TH) * (W / fw::TW);
    if (planar) {
        if (dtype != FMRI_BF16) return 0;
        return 1 | (Cout == (fwd_wide(0, 1, ntile, Cout) ? 64 : 32) ? 2 : 0);
    }
    if (fwd_use_ws() == 0) return 0;
    if (dtype == FMRI_F32) return 1;                     // the pooled copy rides the fp32 drain; the logits need a 64-wide block
    const int bn = fwd_wide(0, 0, ntile, Cout) ? 64 : 32;
    return 1 | (Cout == bn ? 2 : 0);
}
int conv3d_fwd_mfma_tail(const void* src0, int C0, const void* w, const float* bias, void* y, void* pool, const float* w1, const float* b1,
                         float* logits, int N, int D, int H, int W, int Cout, int act, float alpha, int dtype, int planar, hipStream_t st) {
    const int ok = conv3d_fwd_tail_ok(C0, Cout, N, D, H, W, dtype, planar);
    if ((pool && !(ok & 1)) || (logits && (!(ok & 2) || !w1 || !b1))) return FMRI_E_SHAPE;
    return conv3d_fwd_mfma_launch(0, src0, C0, 0, planar, nullptr, 0, w, bias, nullptr, nullptr, y, N, D, H, W, Cout, act, alpha,
                                  FwdTail{(bf16_t*)pool, w1, b1, logits, nullptr}, dtype, st);
}
// can this plain 3-D launch carry a normalisation tail (statistics of its output / the backward reductions in its asynchronous epilogue)?
int conv3d_fwd_ntail_ok(int C0, int C1, int Cout, int N, int D, int H, int W, int dtype) {
    if (dtype != FMRI_BF16 || !conv3d_fwd_mfma_ok(C0, C1, Cout, D, H, W, dtype) || conv3d_fwd_needs_cube(D, H, W) || !fwd_use_ws() || !fwd_async()) return 0;
    const int ntile = N * (D / fw::TD) * (H / fw::TH) * (W / fw::TW);
    return ntile * (Cout / (fwd_wide(0, 0, ntile, Cout) ? 64 : 32)) > fwd_cu_count();
}
int conv3d_fwd_ntail_slots() { return fwd_cu_count(); }      // nws holds 1 + this many [G][Cout][2] blocks
// kind 0: y = act(conv(src) [+ residual]) and nws += {sum y, sum y^2};  kind 1: y = dz = conv(src) * act'(z(x)), nws += {sum dz, sum dz * x}
// (x = the normalised block's conv output, passed as `lines`; nss[g][Cout][2] = that block's {scale, shift}).  nws: [1 + slots][G][Cout][2],
// zero on entry; the sums are left in the slots 1.. (one per workgroup) for the caller to fold into block 0.
int conv3d_fwd_mfma_ntail(int kind, const void* src0, int C0, int up0, const void* src1, int C1, const void* w, const float* bias,
                          const void* lines, void* y, int N, int D, int H, int W, int Cout, int act, float alpha, double* nws, int per_instance,
                          const float* nss, hipStream_t st) {
    if (!nws || (kind == 1 && (!lines || !nss))) return FMRI_E_SHAPE;
    FwdTail t{};
    t.nws = nws;
    t.n_per = per_instance;
    t.n_grp = per_instance ? N : 1;
    if (kind == 1) t.nss = nss;
    return conv3d_fwd_mfma_launch(0, src0, C0, up0, 0, src1, C1, w, bias, kind == 1 ? lines : nullptr, kind == 1 ? nullptr : lines, y, N, D, H, W,
                                  Cout, act, alpha, t, FMRI_BF16, st);
}
// plain 3-D conv whose epilogue adds `residual` and whose bias depends on the output voxel's border class (FwdTail::bias27); only the
// warp-specialised kernel implements it
int conv3d_fwd_mfma_res_b27(const void* src, int C, const void* w, const float* bias27, const void* residual, void* y, int N, int D, int H,
                            int W, int Cout, int act, float alpha, int dtype, hipStream_t st) {
    if (!fwd_use_ws() || !bias27 || !residual) return FMRI_E_SHAPE;
    return conv3d_fwd_mfma_launch(0, src, C, 0, 0, nullptr, 0, w, bias27 + 13 * (int64_t)Cout, nullptr, residual, y, N, D, H, W, Cout, act, alpha,
                                  FwdTail{nullptr, nullptr, nullptr, nullptr, bias27}, dtype, st);
}
int conv3d_fwd_mfma(const void* src0, int C0, int up0, int planar, const void* src1, int C1, const void* w, const float* bias,
                    const void* mask, void* y, int N, int D, int H, int W, int Cout, int act, float alpha, int dtype, hipStream_t st) {
    return conv3d_fwd_mfma_ex(0, src0, C0, up0, planar, src1, C1, w, bias, mask, nullptr, y, N, D, H, W, Cout, act, alpha, dtype, st);
}
