// Probe (test infrastructure): does `buffer_load_dwordx4 ... offen lds` (LDS-DMA through a buffer descriptor) write ZEROS into LDS for a lane
// whose offset fails the descriptor's range check (voffset >= num_records), and is the range check blind to the descriptor base?
// The forward kernel's producers rely on it for out-of-volume halo rows (conv3d_mfma.hip, dma16_buf).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) int i32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ void dma16_buf(i32x4 rsrc, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}
__global__ void k(const unsigned* p, unsigned* out, long long base_shift) {
    __shared__ __attribute__((aligned(16))) unsigned lds[512];
    for (int i = threadIdx.x; i < 512; i += 64) lds[i] = 0xAAAAAAAAu;
    __syncthreads();
    // descriptor base = p + base_shift bytes (may point in front of the allocation: only in-range lanes are really fetched)
    const unsigned long long a = (unsigned long long)p + base_shift;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffff);
    r[2] = 0x80000000;
    r[3] = 0x00020000;
    const unsigned lane = threadIdx.x;
    const unsigned good = lane * 16;                      // offsets are unsigned: the descriptor base is the LOWEST address a lane may read
    dma16_buf(r, (lane & 1) ? 0x80000000u : good, (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)lds);
    dma16_buf(r, (lane & 2) ? 0xFFFFFFF0u : good, (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)lds + 1024);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
int main() {
    CK(hipSetDevice(0));
    std::vector<unsigned> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x1000000u + (unsigned)i;
    unsigned *d, *o;
    CK(hipMalloc(&d, h.size() * 4)); CK(hipMalloc(&o, 512 * 4));
    CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    int bad = 0;
    for (long long shift : {0ll, -4096ll, 65536ll}) {
        const unsigned* src = d + 32768;                         // room on both sides of the source for the shifted descriptor base
        k<<<1, 64>>>(src, o, shift);
        CK(hipDeviceSynchronize());
        std::vector<unsigned> r(512);
        CK(hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost));
        const unsigned first = 0x1000000u + 32768 + (unsigned)(shift / 4);
        int zeros = 0, kept = 0, wrong = 0;
        for (int half = 0; half < 2; ++half)
            for (int lane = 0; lane < 64; ++lane)
                for (int w = 0; w < 4; ++w) {
                    const unsigned v = r[half * 256 + lane * 4 + w];
                    const bool oob = half == 0 ? (lane & 1) : (lane & 2);
                    if (!oob) { if (v != first + lane * 4 + w) ++wrong; }
                    else if (v == 0) ++zeros;
                    else if (v == 0xAAAAAAAAu) ++kept;
                    else ++wrong;
                }
        printf("base shift %6lld: in-range words wrong %d | out-of-range lanes: %d words zero, %d words left untouched\n", shift, wrong, zeros, kept);
        bad += wrong + kept;
    }
    printf(bad ? "PROBE FAILED: out-of-range LDS-DMA lanes do not zero-fill\n" : "PROBE OK: out-of-range LDS-DMA lanes write zeros, the range check ignores the base\n");
    return bad ? 1 : 0;
}
