// Probe (test infrastructure): what does rocprofv3's FETCH_SIZE report for the forward kernel's OWN halo access pattern?
// MI355X_MICROARCH.md calibrates the counter for wide streaming reads only (16 B per lane over whole 128-byte lines: it reports exactly half)
// and says every other width must be calibrated.  A halo row of a 64-channel tensor is 64 B (one 32-channel chunk) of a 128-B voxel: four lanes
// x 16 B every 128 B.  Kernels over a 2 GiB buffer (beyond the 256 MiB Infinity Cache), each reading a known byte count through
// `buffer_load_dwordx4 ... lds` like the producers:
//   full   : every byte once, lines whole                     (the guide's case: expect FETCH = bytes / 2)
//   half0  : the first 64 B of every 128-B line                (1 GiB requested)
//   half01 : per 64 KiB block the first halves, then - behind a barrier - the second halves (the two chunks of one tile, microseconds apart)
//   quarter: 32 B (two lanes) of every 128-B line              (what a 16-channel tensor would look like)
// Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and under `--kernel-trace --stats` (durations); tools/r05/run21.sh prints the table.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) int i32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ void dma16_buf(i32x4 rsrc, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ i32x4 rsrc_of(const void* p) {
    const unsigned long long a = (unsigned long long)p;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffff);
    r[2] = 0x80000000;
    r[3] = 0x00020000;
    return r;
}
// MODE 0 full, 1 half0, 2 half01, 3 quarter.  One workgroup of 256 threads per 64 KiB block of the buffer (512 lines of 128 B).
template <int MODE>
__global__ void __launch_bounds__(256) k_read(const char* buf, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) unsigned lds[4 * 1024];                      // 16 KiB: 4 waves x 4 KiB landing area (overwritten)
    const char* blk = buf + (size_t)blockIdx.x * 65536;
    const i32x4 r = rsrc_of(blk);
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)lds + wv * 4096);
    if (MODE == 0) {
        for (int it = 0; it < 16; ++it) dma16_buf(r, (it * 4 + wv) * 1024 + lane * 16, dst);           // 64 x 1 KiB pieces: whole lines
    } else if (MODE == 1 || MODE == 2) {
        // a piece = 64 lanes x 16 B = 16 half-lines: lane l -> line (l >> 2), 16-byte slot (l & 3) of its first 64 B
        for (int it = 0; it < 8; ++it) dma16_buf(r, ((it * 4 + wv) * 16 + (lane >> 2)) * 128 + (lane & 3) * 16, dst);
        if (MODE == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            for (int it = 0; it < 8; ++it) dma16_buf(r, ((it * 4 + wv) * 16 + (lane >> 2)) * 128 + 64 + (lane & 3) * 16, dst);
        }
    } else {
        // 32 B of every line: lane l -> line (l >> 1), slot (l & 1)
        for (int it = 0; it < 4; ++it) dma16_buf(r, ((it * 4 + wv) * 32 + (lane >> 1)) * 128 + (lane & 1) * 16, dst);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && lds[5] == 0x12345u) sink[0] = 1;                              // keep the loads alive
}
int main() {
    CK(hipSetDevice(0));
    const size_t bytes = (size_t)2 << 30;
    char* d; unsigned* sink;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(d, 1, bytes));
    const int blocks = (int)(bytes / 65536);
    for (int rep = 0; rep < 2; ++rep) {
        k_read<0><<<blocks, 256>>>(d, sink);
        k_read<1><<<blocks, 256>>>(d, sink);
        k_read<2><<<blocks, 256>>>(d, sink);
        k_read<3><<<blocks, 256>>>(d, sink);
    }
    CK(hipDeviceSynchronize());
    printf("requested MiB per launch: full %zu, half0 %zu, half01 %zu, quarter %zu\n", bytes >> 20, bytes >> 21, bytes >> 20, bytes >> 22);
    return 0;
}
