// Practical MFMA ceiling of this device on random bf16 data (test infrastructure): back-to-back v_mfma_f32_32x32x16_bf16 /
// 16x16x32 from registers, 1 or 2 waves per SIMD, all CUs.  Prints TFLOP/s and the in-kernel clock (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int SHAPE>
__global__ void k_peak(const uint4* __restrict__ in, float* __restrict__ out, int iters, unsigned long long* clk) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    uint4 ua = in[t % 4096], ub = in[(t + 77) % 4096];
    bf16x8_t a = __builtin_bit_cast(bf16x8_t, ua), b = __builtin_bit_cast(bf16x8_t, ub);
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float res = 0.f;
    if (SHAPE == 32) {
        f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
        }
        res = c0[0] + c1[1] + c2[2] + c3[3];
    } else {
        f32x4 c[8] = {};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) c[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[k], 0, 0, 0);
        }
        for (int k = 0; k < 8; ++k) res += c[k][k & 3];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[t] = res;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main() {
    CK(hipSetDevice(0));
    std::vector<unsigned short> h(4096 * 8);
    srand(3);
    for (auto& v : h) { float f = (rand() / (float)RAND_MAX) * 2.f - 1.f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
    uint4* din; float* dout; unsigned long long* dclk;
    CK(hipMalloc(&din, 4096 * 16)); CK(hipMalloc(&dout, 256 * 8 * 512 * 4)); CK(hipMalloc(&dclk, 16));
    CK(hipMemcpy(din, h.data(), 4096 * 16, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int shape : {32, 16}) for (int wps : {1, 2}) {
        const int threads = 256 * wps, blocks = 256, iters = 20000;
        const double flop_per_iter = (shape == 32 ? 4 * 2.0 * 32 * 32 * 16 : 8 * 2.0 * 16 * 16 * 32);
        for (int rep = 0; rep < 3; ++rep) {   // ~1 s of warm running before the measured launch
            if (shape == 32) k_peak<32><<<blocks, threads>>>(din, dout, iters * 4, dclk); else k_peak<16><<<blocks, threads>>>(din, dout, iters * 4, dclk);
        }
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        if (shape == 32) k_peak<32><<<blocks, threads>>>(din, dout, iters, dclk); else k_peak<16><<<blocks, threads>>>(din, dout, iters, dclk);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long clk[2]; CK(hipMemcpy(clk, dclk, 16, hipMemcpyDeviceToHost));
        double waves = (double)blocks * threads / 64;
        printf("mfma %dx%d waves/SIMD %d: %.0f TFLOP/s, in-kernel clock %.2f GHz\n", shape, shape, wps,
               waves * iters * flop_per_iter / (ms * 1e-3) / 1e12, (double)clk[0] / (double)clk[1] * 0.1);
    }
    return 0;
}
