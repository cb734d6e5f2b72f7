// Hardware-semantics probe for gfx950 (test infrastructure, not product code).
// Checks the MFMA operand/result lane maps and the ds_read_b64_tr_b16 gather that the
// conv kernels in fetal-mri-segmentation_amd/csrc rely on.  Prints PASS/FAIL per check.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float bf2f(unsigned short h) { unsigned u = ((unsigned)h) << 16; float f; memcpy(&f, &u, 4); return f; }

// ---- check 1: 32x32x16 bf16: D[i][j] = sum_k A[i][k] B[k][j] with the documented maps
__global__ void k_mfma32(const unsigned short* A /*[32][16]*/, const unsigned short* B /*[16][32]*/, float* D /*[32][32]*/) {
    int l = threadIdx.x, r = l & 31, h = l >> 5;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = A[r * 16 + 8 * h + j]; b[j] = B[(8 * h + j) * 32 + r]; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    for (int reg = 0; reg < 16; ++reg) {
        int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        D[row * 32 + r] = c[reg];
    }
}
// ---- check 2: 16x16x32
__global__ void k_mfma16(const unsigned short* A /*[16][32]*/, const unsigned short* B /*[32][16]*/, float* D /*[16][16]*/) {
    int l = threadIdx.x, r = l & 15, g = l >> 4;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = A[r * 32 + 8 * g + j]; b[j] = B[(8 * g + j) * 16 + r]; }
    f32x4 c = {0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    for (int reg = 0; reg < 4; ++reg) D[(g * 4 + reg) * 16 + r] = c[reg];
}
// ---- check 3: raw dump of ds_read_b64_tr_b16.  LDS image: [rows][64 cols] of u16 = row*256+col.
// lane 4q+p of each 16-lane group g supplies the address of row (4g+q), cols 4p..4p+3  (+16*cb)
__global__ void k_tr_dump(unsigned short* out /*[64 lanes][4]*/, int cb) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (unsigned short)(((i / 64) << 8) | (i % 64));
    __syncthreads();
    int l = threadIdx.x, g = l >> 4, q = (l & 15) >> 2, p = l & 3;
    int row = 4 * g + q, col = 16 * cb + 4 * p;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(&lds[row * 64 + col]));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
// ---- check 4: wgrad-style product through tr reads.  Xs[v][ci] (32 voxels x 32 ci), Ys[v][co] (32 voxels x 32 co) in LDS,
// D[co][ci] = sum_v Ys[v][co]*Xs[v][ci] using two 32x32x16 steps (k = voxel).
// For the 32x32x16 operand, lane l (r=l&31,h=l>>5) needs M[k=8h+j][r], j=0..7 from row-major [k][32] storage.
// 16-lane group g of the wave: lanes 16g..16g+15 -> r = 16*(g&1)+i, h = g>>1.  One tr read delivers 4 k-rows x 16 cols:
// read#0 rows 8h+0..3, read#1 rows 8h+4..7, cols 16*(g&1)..+15.
__device__ inline bf16x8 tr_frag32(const unsigned short* base /*[k][ld]*/, int ld, int k0, int c0, int l) {
    int g = l >> 4, q = (l & 15) >> 2, p = l & 3, h = g >> 1;
    int col = c0 + 16 * (g & 1) + 4 * p;
    const unsigned short* p0 = base + (k0 + 8 * h + q) * ld + col;
    const unsigned short* p1 = base + (k0 + 8 * h + 4 + q) * ld + col;
    s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
    s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
    bf16x8 f; for (int j = 0; j < 4; ++j) { f[j] = v0[j]; f[4 + j] = v1[j]; }
    return f;
}
__global__ void k_wgrad_tr(const unsigned short* X /*[32][32]*/, const unsigned short* Y /*[32][32]*/, float* D /*[co 32][ci 32]*/) {
    __shared__ __attribute__((aligned(16))) unsigned short xs[32 * 32];
    __shared__ __attribute__((aligned(16))) unsigned short ys[32 * 32];
    for (int i = threadIdx.x; i < 1024; i += 64) { xs[i] = X[i]; ys[i] = Y[i]; }
    __syncthreads();
    int l = threadIdx.x, r = l & 31, h = l >> 5;
    f32x16 c = {0};
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8 a = tr_frag32(ys, 32, 16 * ks, 0, l);  // A[row=co][k=voxel]
        bf16x8 b = tr_frag32(xs, 32, 16 * ks, 0, l);  // B[k=voxel][col=ci]
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
    for (int reg = 0; reg < 16; ++reg) {
        int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        D[row * 32 + r] = c[reg];
    }
}
// ---- check 5: global_load_lds 16B semantic: LDS dest = uniform base + lane*16
__global__ void k_glds(const unsigned* src /*[64*4]*/, unsigned* out /*[64*4]*/) {
    __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4];
    int l = threadIdx.x;
    // each lane loads the 16B of lane (63-l): source permutation, linear dest
    const unsigned* g = src + (63 - l) * 4;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = lds[l * 4 + j];
}

int main() {
    int dev = 0; CK(hipSetDevice(dev));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
    printf("device: %s arch %s CUs %d\n", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    srand(7);
    auto rnd = [](){ return (float)((rand() % 17) - 8); };
    // check 1
    {
        std::vector<unsigned short> A(32 * 16), B(16 * 32); std::vector<float> D(1024), R(1024, 0.f);
        for (auto& x : A) x = f2bf(rnd()); for (auto& x : B) x = f2bf(rnd());
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 16; ++k) R[i * 32 + j] += bf2f(A[i * 16 + k]) * bf2f(B[k * 32 + j]);
        unsigned short *dA, *dB; float* dD; CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dD, 4096));
        CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
        k_mfma32<<<1, 64>>>(dA, dB, dD); CK(hipDeviceSynchronize()); CK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
        int bad = 0; for (int i = 0; i < 1024; ++i) if (D[i] != R[i]) ++bad;
        printf("check1 mfma_32x32x16 maps: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
    }
    {
        std::vector<unsigned short> A(16 * 32), B(32 * 16); std::vector<float> D(256), R(256, 0.f);
        for (auto& x : A) x = f2bf(rnd()); for (auto& x : B) x = f2bf(rnd());
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 32; ++k) R[i * 16 + j] += bf2f(A[i * 32 + k]) * bf2f(B[k * 16 + j]);
        unsigned short *dA, *dB; float* dD; CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dD, 1024));
        CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
        k_mfma16<<<1, 64>>>(dA, dB, dD); CK(hipDeviceSynchronize()); CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
        int bad = 0; for (int i = 0; i < 256; ++i) if (D[i] != R[i]) ++bad;
        printf("check2 mfma_16x16x32 maps: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
    }
    {
        unsigned short* dO; CK(hipMalloc(&dO, 64 * 4 * 2)); std::vector<unsigned short> O(256);
        for (int cb = 0; cb < 2; ++cb) {
            k_tr_dump<<<1, 64>>>(dO, cb); CK(hipDeviceSynchronize()); CK(hipMemcpy(O.data(), dO, 512, hipMemcpyDeviceToHost));
            // expectation: lane i of group g gets column (16cb + i), rows 4g+0..3 in elements 0..3
            int bad = 0;
            for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
                int g = l >> 4, i = l & 15; unsigned short e = (unsigned short)(((4 * g + j) << 8) | (16 * cb + i));
                if (O[l * 4 + j] != e) ++bad;
            }
            printf("check3 ds_read_tr16_b64 (cb=%d): %s (%d mismatches)\n", cb, bad ? "FAIL" : "PASS", bad);
            if (bad) { for (int l = 0; l < 64; ++l) { printf("  lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" r%dc%d", O[l * 4 + j] >> 8, O[l * 4 + j] & 255); printf("\n"); } }
        }
    }
    {
        std::vector<unsigned short> X(1024), Y(1024); std::vector<float> D(1024), R(1024, 0.f);
        for (auto& x : X) x = f2bf(rnd()); for (auto& x : Y) x = f2bf(rnd());
        for (int co = 0; co < 32; ++co) for (int ci = 0; ci < 32; ++ci) for (int v = 0; v < 32; ++v) R[co * 32 + ci] += bf2f(Y[v * 32 + co]) * bf2f(X[v * 32 + ci]);
        unsigned short *dX, *dY; float* dD; CK(hipMalloc(&dX, 2048)); CK(hipMalloc(&dY, 2048)); CK(hipMalloc(&dD, 4096));
        CK(hipMemcpy(dX, X.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(dY, Y.data(), 2048, hipMemcpyHostToDevice));
        k_wgrad_tr<<<1, 64>>>(dX, dY, dD); CK(hipDeviceSynchronize()); CK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
        int bad = 0; for (int i = 0; i < 1024; ++i) if (D[i] != R[i]) ++bad;
        printf("check4 wgrad via tr reads: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
    }
    {
        std::vector<unsigned> S(256), O(256); for (int i = 0; i < 256; ++i) S[i] = i;
        unsigned *dS, *dO; CK(hipMalloc(&dS, 1024)); CK(hipMalloc(&dO, 1024));
        CK(hipMemcpy(dS, S.data(), 1024, hipMemcpyHostToDevice));
        k_glds<<<1, 64>>>(dS, dO); CK(hipDeviceSynchronize()); CK(hipMemcpy(O.data(), dO, 1024, hipMemcpyDeviceToHost));
        int bad = 0; for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) if (O[l * 4 + j] != (unsigned)((63 - l) * 4 + j)) ++bad;
        printf("check5 global_load_lds 16B linear dest: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
    }
    return 0;
}
