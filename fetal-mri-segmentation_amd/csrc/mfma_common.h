// Shared device helpers of the bf16 MFMA conv kernels (conv3d_mfma.hip: forward / input gradient; conv3d_wgrad.hip: weight gradient):
// LDS-DMA forms, buffer descriptors, packed bf16 helpers, swizzles, XCD-aware numbering.  Everything lives in an anonymous namespace: each
// translation unit gets its own copy of the few __device__ objects (zero page, debug counters).
#pragma once
#include "common.h"
#include <type_traits>
#include <utility>
#include <cstdlib>

// debug builds (FMRI_CHECK / FMRI_PROF) export their read-back functions once per translation unit: the weight-gradient unit appends _wgrad
#ifdef FMRI_MFMA_TU_WGRAD
#define FMRI_DEBUG_NAME(x) x##_wgrad
#else
#define FMRI_DEBUG_NAME(x) x
#endif

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

struct SrcB {
    const bf16_t* p0; const bf16_t* p1; int C0, C1, up0, dsh, planar;   // dsh = 0: the fused x2 leaves D alone (2-D slices)
};

// ======================================================================================================== forward / dgrad
// Persistent workgroups (one per CU, 8 waves) walk a list of (spatial tile, Cout block) pairs.  For every 32-channel input
// chunk the 6x10x18 halo tile lives in LDS and is re-used by all 27 taps; the filters arrive as nine 3-tap slabs
// (one per (kd,kh)).  Everything is brought in by LDS-DMA (global_load_lds_dwordx4) one phase ahead of its use:
//   phase g:  s_waitcnt vmcnt(0) ; s_barrier ; issue DMA {filter slab of phase g+1, 1/9 of the NEXT chunk's halo} ;
//             24 MFMAs per wave on filter ring slot g&1 and halo ring slot item&1
// so one barrier per phase hands over both rings and no load latency is exposed after the prologue.  LDS is used to the
// last byte: 2 x 68 KiB halo + 2 x 12 KiB filter = 160 KiB.
namespace fw {
constexpr int TD = 4, TH = 8, TW = 16;                 // 512 output voxels per workgroup = 16 MFMA column tiles of 32
constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;   // halo 6 x 10 x 18
constexpr int HVOX = HD * HH * HW;                     // 1080
constexpr int H_INSTR = (HVOX * 4 + 63) / 64;          // 68 DMA wave-instructions per halo chunk
constexpr int HALO_BYTES = H_INSTR * 1024;             // 69,632 (1080 rows x 64 B + pad)
constexpr int NTHREADS = 512;
}  // namespace fw

// Zeros: the DMA source for out-of-volume halo rows.  The fwd kernel advances EVERY lane's source pointer by 64 B per channel chunk,
// zero-page lanes included, so the page must cover Cin * 2 bytes: 8 KiB + slack = Cin <= 4096 (checked in conv3d_fwd_mfma_ok).
__device__ uint4 g_zero_page[520];
// Its address, fetched ONCE per kernel into a scalar register pair the compiler cannot re-derive (round 5): written as `g_zero_page` at the
// point of use, hipcc re-materialised the address at EVERY DMA piece that may select it - s_getpc + s_load_dwordx2 from the GOT + a full
// s_waitcnt lgkmcnt(0) in front of the piece, i.e. one scalar-memory latency in the issuing wave's path per piece: every x-plane piece of the
// weight-gradient kernels (three per unit and wave) and every halo piece of a border tile of the forward kernels.
__device__ __forceinline__ const unsigned char* zero_page_addr() {
    const unsigned char* p = reinterpret_cast<const unsigned char*>(g_zero_page);
    asm volatile("" : "+s"(p));
    return p;
}
#ifdef FMRI_CHECK
__device__ long long g_chk[16];
extern "C" void FMRI_DEBUG_NAME(fmri_debug_chk)(long long* out) { hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chk), sizeof(g_chk)); }
#endif
#ifdef FMRI_PROF
__device__ unsigned long long g_prof[12];
// per phase index of an item (0..8), producers' issue time and count: [0..8] items that drain a staged tile, [9..17] other items, [18..35] counts
__device__ unsigned long long g_prof_ph[36];
extern "C" void FMRI_DEBUG_NAME(fmri_debug_prof)(unsigned long long* out, int reset) {
    if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(g_prof));
    if (reset) { unsigned long long z[12] = {}; hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)); }
}
extern "C" void FMRI_DEBUG_NAME(fmri_debug_prof_phases)(unsigned long long* out, int reset) {
    if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof_ph), sizeof(g_prof_ph));
    if (reset) { unsigned long long z[36] = {}; hipMemcpyToSymbol(HIP_SYMBOL(g_prof_ph), z, sizeof(z)); }
}
#define PROF_T(x) const unsigned long long x = __builtin_amdgcn_s_memtime()
#define PROF_ADD(i, a, b) prof[i] += (b) - (a)
#else
#define PROF_T(x)
#define PROF_ADD(i, a, b)
#endif                       // 128 B of zeros: DMA source for out-of-volume halo rows

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
// same with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset: no 64-bit VALU address arithmetic at all
__device__ __forceinline__ void dma16_s(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}
// LDS-DMA through a buffer descriptor (round 5): wave-uniform 128-bit resource (base, num_records = 2^31, raw) + one 32-bit byte offset per
// lane.  A lane whose offset fails the range check (>= num_records) gets ZEROS written to its 16 bytes of LDS (tools/probe/probe_bufdma.hip),
// so an out-of-volume halo row needs neither a zero page nor a per-lane pointer select: its offset is simply out of range.
typedef __attribute__((ext_vector_type(4))) int i32x4;
constexpr unsigned DMA_OOB = 0x80000000u;
__device__ __forceinline__ void dma16_buf(i32x4 rsrc, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ i32x4 dma_rsrc(const void* base, int records = (int)DMA_OOB) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xffff;      // stride 0, no swizzle
    r[2] = __builtin_amdgcn_readfirstlane(records);                            // every offset >= records is out of range: 0 = the whole piece
    r[3] = 0x00020000;
    return r;
}
// max of two non-NaN-critical floats as ONE v_max_f32 (fmaxf adds a canonicalising v_max x,x per operand)
__device__ __forceinline__ float vmax(float a, float b) {
    float o;
    asm("v_max_f32 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b));
    return o;
}
// two floats -> one dword of bf16 (one v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned pack2bf(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// per 16-bit half of a dword of two bf16 values: 0xffff where the value is > 0, else 0 - three packed 16-bit integer instructions (a bf16 is
// positive exactly when its bits, read as int16, are: +0 is 0, negative values and -0 have the sign bit; a NaN counts by its sign).  The
// float form (shift, compare, select per half) took eight instructions per dword in the drain of every input-gradient launch.
__device__ __forceinline__ unsigned pos_mask2(unsigned m, unsigned zero2, unsigned one2) {
    unsigned t;
    asm("v_pk_max_i16 %0, %1, %2\n\tv_pk_min_u16 %0, %0, %3\n\tv_pk_sub_u16 %0, %2, %0" : "=&v"(t) : "v"(m), "v"(zero2), "v"(one2));
    return t;
}
// max(x, 0) on two packed bf16 values through their int16 bit patterns
__device__ __forceinline__ unsigned relu2(unsigned m, unsigned zero2) {
    unsigned t;
    asm("v_pk_max_i16 %0, %1, %2" : "=v"(t) : "v"(m), "v"(zero2));
    return t;
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)p;
}

// 16-byte slot swizzle for 64-byte rows: 4 consecutive rows x 4 slots cover a 256-B bank row exactly once per slot index
__device__ __forceinline__ int swz64(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }

// XCD-aware workgroup numbering: the hardware deals consecutive workgroup ids round-robin over the 8 XCDs (each with its own L2), so ids that
// should share cached data - the (kd, Cout, Cin) workgroups reading the same planes, the Cout blocks / neighbouring tiles of one input tile -
// are renumbered such that each XCD owns a contiguous range of logical ids.
__device__ __forceinline__ int xcd_logical_id(int b, int nb) {
    const int full = nb & ~7;
    return b < full ? (b & 7) * (full >> 3) + (b >> 3) : b;
}

// compile-time loop: body(std::integral_constant<int, I>{}) for I = 0 .. N-1 (a `#pragma unroll` loop whose body instantiates several large
// lambdas can exceed the pragma-unroll size limit and silently stay a loop - its per-phase array indices then turn dynamic and the arrays
// move to scratch memory)
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}
}  // namespace

inline int fwd_cu_count() {             // CU count of the current device, queried once (persistent grid = one workgroup per CU)
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            ncu = v;
        else
            ncu = 256;
    }
    return ncu;
}
