// LDS read throughput per CU of the fragment-read instructions the conv kernels use (tuning aid):
//   mode 0: ds_read_b128 (forward / input-gradient fragments)      mode 1: ds_read_b64 (plain)      mode 2: ds_read_b64_tr_b16 (weight gradient:
//   the hardware-transposing read that gathers an MFMA operand whose k-dimension is strided in memory)
// 8 waves per workgroup, one workgroup per CU, every wave issues back-to-back reads from a conflict-free 64-KiB image; clock64() brackets
// the loop, so the result is bytes per shader clock per CU (LDS peak on CDNA: 128 B/clk/CU).
//   hipcc --offload-arch=gfx950 -O3 probe_lds_read.hip -o probe_lds_read && ./probe_lds_read
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <int MODE>
__global__ void __launch_bounds__(512) k_probe(int iters, unsigned long long* out, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 1024];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 1024 / 4; i += 512) reinterpret_cast<unsigned*>(lds)[i] = i;
    __syncthreads();
    unsigned acc = 0;
    // lane-linear addresses inside a wave-private 8-KiB window (conflict-free for plain reads); the transposing read uses the kernels' own
    // lane -> (row, slot) map on 128-byte rows with the half swizzle
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int row = 8 * (g >> 1) + q, slot = 2 * (g & 1) + (p >> 1);
    const int tr_off = row * 128 + ((slot ^ (((row >> 1) & 1) << 2)) << 4) + (p & 1) * 8;
    const unsigned char* base = lds + wv * 8192;
    // eight reads in flight per wave, one wait per batch (inline asm: the compiler serialises the transposing-read builtin otherwise)
    const unsigned a0 = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)base;
    const unsigned addr = a0 + (MODE == 0 ? lane * 16 : (MODE == 1 ? lane * 8 : tr_off));
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            u32x4 v[8];
            asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\tds_read_b128 %3, %8 offset:3072\n\t"
                         "ds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\tds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]) : "v"(addr));
#pragma unroll
            for (int u = 0; u < 8; ++u) acc ^= v[u][0];
        } else {
            u32x2 v[8];
            if (MODE == 1)
                asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:512\n\tds_read_b64 %2, %8 offset:1024\n\tds_read_b64 %3, %8 offset:1536\n\t"
                             "ds_read_b64 %4, %8 offset:2048\n\tds_read_b64 %5, %8 offset:2560\n\tds_read_b64 %6, %8 offset:3072\n\tds_read_b64 %7, %8 offset:3584\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]) : "v"(addr));
            else
                asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:512\n\tds_read_b64_tr_b16 %2, %8 offset:2048\n\tds_read_b64_tr_b16 %3, %8 offset:2560\n\t"
                             "ds_read_b64_tr_b16 %4, %8 offset:4096\n\tds_read_b64_tr_b16 %5, %8 offset:4608\n\tds_read_b64_tr_b16 %6, %8 offset:6144\n\tds_read_b64_tr_b16 %7, %8 offset:6656\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]) : "v"(addr));
#pragma unroll
            for (int u = 0; u < 8; ++u) acc ^= v[u][0];
        }
    }
    const unsigned long long t1 = clock64();
    if (lane == 0 && wv == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 0x12345u) sink[0] = acc;
}

int main() {
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned long long* out;
    unsigned* sink;
    if (hipMalloc(&out, ncu * 8) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    const int iters = 20000;
    const char* names[3] = {"ds_read_b128", "ds_read_b64", "ds_read_b64_tr_b16"};
    const int bytes_per_lane[3] = {16, 8, 8};
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) k_probe<0><<<ncu, 512>>>(iters, out, sink);
            else if (mode == 1) k_probe<1><<<ncu, 512>>>(iters, out, sink);
            else k_probe<2><<<ncu, 512>>>(iters, out, sink);
            (void)hipDeviceSynchronize();
        }
        unsigned long long h[1024];
        (void)hipMemcpy(h, out, ncu * 8, hipMemcpyDeviceToHost);
        double cyc = 0;
        for (int i = 0; i < ncu; ++i) cyc += (double)h[i];
        cyc /= ncu;
        const double bytes = (double)iters * 8 * 512 * bytes_per_lane[mode];
        printf("%-20s %8.1f B/clk/CU  (%.0f clocks for %d x 8 reads per wave, 8 waves)\n", names[mode], bytes / cyc, cyc, iters);
    }
    return 0;
}
