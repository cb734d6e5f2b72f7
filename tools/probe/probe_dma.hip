// Throughput of the three ways a CU can pull 16-byte pieces out of L2 / HBM (tuning aid for the conv kernels' staging):
//   mode 0: global_load_lds_dwordx4 (LDS-DMA, what the kernels use)      mode 1: global_load_dwordx4 into VGPRs      mode 2: LDS-DMA + VGPR loads mixed 1:1
// Each 512-thread workgroup (one per CU) sweeps its own `span` bytes `reps` times; span = 64 KiB keeps it L2-resident, 64 MiB streams from HBM.
//   hipcc --offload-arch=gfx950 -O3 probe_dma.hip -o probe_dma && ./probe_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int MODE>
__global__ void __launch_bounds__(512) k_probe(const uint4* __restrict__ src, size_t span16, int reps, uint4* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 1024];
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint4* base = src + (size_t)blockIdx.x * span16;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int r = 0; r < reps; ++r) {
        for (size_t i = (size_t)wv * 64 + lane; i < span16; i += 512 * 4) {          // 4 instructions per wave per trip
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint4* p = base + (i + (size_t)u * 512) % span16;
                if (MODE == 0 || (MODE == 2 && (u & 1))) dma16(p, __builtin_amdgcn_readfirstlane(lds0 + ((wv * 4 + u) & 31) * 1024));
                else {
                    const uint4 v = *p;                    // the compiler keeps the four loads of a trip in flight and waits before the xor
                    acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
                }
            }
            if (MODE == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (acc.x == 0x12345678u || reinterpret_cast<volatile unsigned*>(lds)[threadIdx.x] == 0xdeadbeefu) sink[0] = acc;     // keeps the LDS array allocated
}

int main() {
    int ncu = 256;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    const size_t spans[2] = {64u << 10, 16u << 20};
    uint4 *src, *sink;
    if (hipMalloc(&src, (size_t)ncu * spans[1]) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    (void)hipMemset(src, 1, (size_t)ncu * spans[1]);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int si = 0; si < 2; ++si) {
        const size_t span16 = spans[si] / 16;
        const int reps = si == 0 ? 400 : 2;
        for (int mode = 0; mode < 3; ++mode) {
            for (int it = 0; it < 2; ++it) {
                hipEventRecord(e0);
                if (mode == 0) k_probe<0><<<ncu, 512>>>(src, span16, reps, sink);
                else if (mode == 1) k_probe<1><<<ncu, 512>>>(src, span16, reps, sink);
                else k_probe<2><<<ncu, 512>>>(src, span16, reps, sink);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (double)ncu * spans[si] * reps;
            printf("span %6zu KiB/CU  mode %d (%s): %.3f ms  %.2f TB/s  %.1f KB/us/CU\n", spans[si] >> 10, mode,
                   mode == 0 ? "lds-dma" : (mode == 1 ? "vgpr" : "mixed"), ms, bytes / ms / 1e9, bytes / ncu / ms / 1e3 / 1e3 * 1e0);
        }
    }
    return 0;
}
