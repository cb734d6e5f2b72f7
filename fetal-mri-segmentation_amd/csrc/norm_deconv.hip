// Normalisation (+activation) and 2x2x2 stride-2 transposed convolution — the optional pieces of the reference's conv
// block and up-convolution (reference fetal_net/model/unet3d/unet.py:102-115 BatchNormalization(axis=1) /
// keras-contrib InstanceNormalization(axis=1) + activation; unet.py:132-138 Deconvolution3D(k=2, s=2)).
// All HBM-bound; channels-last, fp32 statistics accumulated in double.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------------- statistics
// ws[g][c][0..1] (double) += sum(x), sum(x^2) over the voxels of group g (g = sample for instance norm, 0 for batch norm).
// Block = 256 threads = (256/CL) voxel lanes x CL channel lanes, CL = min(C,64) rounded to a power of two.
template <typename T>
__global__ void k_norm_reduce(const T* __restrict__ x, double* __restrict__ ws, int64_t V, int C, int per_instance, int vchunk) {
    const int c = blockIdx.y * 64 + (threadIdx.x & 63);
    const int g = blockIdx.z;
    const int vl = threadIdx.x >> 6;                       // 0..3
    const int64_t v0 = (int64_t)blockIdx.x * vchunk;
    const int64_t v1 = min(V, v0 + vchunk);
    float s = 0.f, q = 0.f;
    if (c < C) {
        const T* base = x + (int64_t)g * V * C + c;
        for (int64_t v = v0 + vl; v < v1; v += 4) {
            const float t = to_f<T>(base[v * C]);
            s += t;
            q = fmaf(t, t, q);
        }
    }
    __shared__ float red[2][4][64];
    red[0][vl][threadIdx.x & 63] = s;
    red[1][vl][threadIdx.x & 63] = q;
    __syncthreads();
    if (threadIdx.x < 64 && c < C) {
        double ds = (double)red[0][0][threadIdx.x] + red[0][1][threadIdx.x] + red[0][2][threadIdx.x] + red[0][3][threadIdx.x];
        double dq = (double)red[1][0][threadIdx.x] + red[1][1][threadIdx.x] + red[1][2][threadIdx.x] + red[1][3][threadIdx.x];
        const int gi = per_instance ? g : 0;
        atomicAdd(&ws[((int64_t)gi * C + c) * 2 + 0], ds);
        atomicAdd(&ws[((int64_t)gi * C + c) * 2 + 1], dq);
    }
}
// Same reductions with 16-byte loads (bf16 x 8 channels per lane): C/8 lanes cover one voxel's channels, 256/(C/8) voxels per pass.
// Used when C/8 is a power of two <= 256 (the scalar kernels above move 2 bytes per lane and reach ~1/4 of the HBM rate).
template <typename T, int VEC>
__global__ void __launch_bounds__(256) k_norm_reduce_v(const T* __restrict__ x, double* __restrict__ ws, int64_t V, int C, int per_instance, int vchunk) {
    const int CG = C / VEC, VL = 256 / CG;
    const int cg = threadIdx.x % CG, vl = threadIdx.x / CG;
    const int g = blockIdx.z;
    const int64_t v0 = (int64_t)blockIdx.x * vchunk, v1 = min(V, v0 + vchunk);
    float s[VEC], q[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) s[k] = q[k] = 0.f;
    const T* base = x + (int64_t)g * V * C + cg * VEC;
#pragma unroll 4
    for (int64_t v = v0 + vl; v < v1; v += VL) {
        float xv[VEC];
        ldv<T, VEC>(base + v * C, xv);
#pragma unroll
        for (int k = 0; k < VEC; ++k) { s[k] += xv[k]; q[k] = fmaf(xv[k], xv[k], q[k]); }
    }
    __shared__ float red[2][256][VEC + 1];
#pragma unroll
    for (int k = 0; k < VEC; ++k) { red[0][threadIdx.x][k] = s[k]; red[1][threadIdx.x][k] = q[k]; }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {          // (C up to 512: 64 channel groups x 4 voxel lanes)
        double ds = 0, dq = 0;
        for (int l = 0; l < VL; ++l) { ds += red[0][l * CG + c / VEC][c % VEC]; dq += red[1][l * CG + c / VEC][c % VEC]; }
        const int gi = per_instance ? g : 0;
        atomicAdd(&ws[((int64_t)gi * C + c) * 2 + 0], ds);
        atomicAdd(&ws[((int64_t)gi * C + c) * 2 + 1], dq);
    }
}
// FROMX: the sign of the block's output is recomputed from x exactly as k_norm_apply formed it (z = fma(x, inv * gamma, beta - mean * inv *
// gamma): the same operands in the same operations, so the same z bit for bit) instead of reading the stored y - one tensor read less in each
// of the two backward passes (round 3; the normalisation passes are HBM-bound)
template <typename T, int VEC, bool FROMX>
__global__ void __launch_bounds__(256) k_norm_bwd_reduce_v(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ dy, const float* __restrict__ stats,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    double* __restrict__ ws, int64_t V, int C, int per_instance, int act, float alpha, int vchunk) {
    const int CG = C / VEC, VL = 256 / CG;
    const int cg = threadIdx.x % CG, vl = threadIdx.x / CG;
    const int g = blockIdx.z, gi = per_instance ? g : 0;
    const int64_t v0 = (int64_t)blockIdx.x * vchunk, v1 = min(V, v0 + vchunk);
    // per lane: sum dz and the RAW second sum, sum dz * x (round 3: the centred form kept mean and 1/s of the lane's 8 channels in registers
    // as well - 128 registers with spills inside the loop, 2.2-3.5 TB/s); the workgroup centres its totals in double before they leave
    float s[VEC], q[VEC], sc[VEC], sh[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        s[k] = q[k] = 0.f;
        if constexpr (FROMX) {
            const float* const st = stats + ((int64_t)gi * C + cg * VEC + k) * 3;
            sc[k] = st[1] * gamma[cg * VEC + k];
            sh[k] = fmaf(-st[0], sc[k], beta[cg * VEC + k]);
        }
    }
    const int64_t base = (int64_t)g * V * C + cg * VEC;
#pragma unroll 4
    for (int64_t v = v0 + vl; v < v1; v += VL) {
        float xv[VEC], yv[VEC], dv[VEC];
        ldv<T, VEC>(x + base + v * C, xv);
        if constexpr (!FROMX) ldv<T, VEC>(y + base + v * C, yv);
        ldv<T, VEC>(dy + base + v * C, dv);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float d = dv[k];
            if constexpr (FROMX) yv[k] = fmaf(xv[k], sc[k], sh[k]);
            if (act == FMRI_ACT_RELU) d = yv[k] > 0.f ? d : 0.f;
            else if (act == FMRI_ACT_LEAKY) d = yv[k] > 0.f ? d : alpha * d;
            s[k] += d;
            q[k] = fmaf(d, xv[k], q[k]);
        }
    }
    __shared__ float red[2][256][VEC + 1];
#pragma unroll
    for (int k = 0; k < VEC; ++k) { red[0][threadIdx.x][k] = s[k]; red[1][threadIdx.x][k] = q[k]; }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        double ds = 0, dq = 0;
        for (int l = 0; l < VL; ++l) { ds += red[0][l * CG + c / VEC][c % VEC]; dq += red[1][l * CG + c / VEC][c % VEC]; }
        const float* const st = stats + ((int64_t)gi * C + c) * 3;
        atomicAdd(&ws[((int64_t)gi * C + c) * 2 + 0], ds);
        atomicAdd(&ws[((int64_t)gi * C + c) * 2 + 1], (dq - (double)st[0] * ds) * (double)st[1]);      // sum dz * xhat, xhat = (x - mean) * inv
    }
}
// voxels per reduction workgroup.  Every workgroup ends with one fp64 atomic per channel into its group's accumulators, so the number of
// workgroups PER GROUP (instance norm: per sample; batch norm: the whole batch) is what has to stay moderate (<= 512 / <= 1024: measured,
// 2048 on one group halves the rate), while small tensors of several samples need a finer cut than the former fixed 4096 to fill the
// chip (4 x 32 x 64 x 64 x 128: 128 workgroups = 2.5 TB/s, 1024 workgroups = 4.6 TB/s)
static inline int norm_vchunk(int64_t V, int N, int per_instance, int C) {
    int64_t c = per_instance ? (V + 511) / 512 : (V * (int64_t)N + 1023) / 1024;
    const int64_t floor_c = (65536 + C - 1) / C;               // and at least ~64 K elements per workgroup: below that its LDS reduction
    if (c < floor_c) c = floor_c;                              // and atomics outweigh the loads (8 x 32 x 64 x 128 x 32: 4096 small workgroups ran 30 % slower)
    // ... unless that leaves the chip mostly empty: the 8x16x16 level of BASELINE configs[1] (4 x 2048 voxels x 256 / 512 channels) ran on 16-32
    // workgroups of 128 serial iterations each - 70 us per call for 4-8 MB (rocprofv3, round 3).  Aim at >= 256 workgroups, 64 voxels or more each.
    const int64_t total = per_instance ? V : V * (int64_t)N;
    if ((total + c - 1) / c * (per_instance ? N : 1) < 256) {
        c = (total * (per_instance ? N : 1) + 255) / 256;
        c = ((c + 63) / 64) * 64;
        return (int)(c < 64 ? 64 : (c > 4096 ? 4096 : c));
    }
    c = ((c + 255) / 256) * 256;
    return (int)(c < 512 ? 512 : (c > 4096 ? 4096 : c));
}
static inline bool norm_vec_ok(int C, int dtype) {
    const int cg = C / 8;
    return dtype == FMRI_BF16 && C % 8 == 0 && cg >= 1 && cg <= 64 && (cg & (cg - 1)) == 0;      // 256 threads = cg channel groups x 256 / cg voxel lanes
}

// stats[g][c] = {mean, 1/s, 1/sigma}: s = sqrt(var+eps) (batch norm, Keras) or sqrt(var)+eps (keras-contrib instance norm)
__global__ void k_norm_finalize(double* __restrict__ ws, float* __restrict__ stats, int G, int C, double M, float eps,
                                int eps_on_std) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G * C) return;
    const double mean = ws[i * 2] / M;
    double var = ws[i * 2 + 1] / M - mean * mean;
    if (var < 0) var = 0;
    const double sigma = sqrt(var);
    const double s = eps_on_std ? sigma + (double)eps : sqrt(var + (double)eps);
    stats[i * 3 + 0] = (float)mean;
    stats[i * 3 + 1] = (float)(1.0 / s);
    // a constant channel (sigma = 0, e.g. channel padding) has xhat = 0 everywhere, so the sigma-term of the gradient vanishes: 1/sigma := 0
    stats[i * 3 + 2] = (float)(eps_on_std ? (sigma > 0 ? 1.0 / sigma : 0.0) : 1.0 / s);
    // leave the accumulators zero for the next reduction (see "ws" in include/fmri_hip.h): one launch less per layer than zeroing in front
    ws[i * 2] = 0.0;
    ws[i * 2 + 1] = 0.0;
}
__global__ void k_zero_d(double* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0;
}
// nss[g][c] = {scale, shift} of the apply pass, z = fma(x, scale, shift): the very operations of k_norm_apply, so that a kernel that
// recomputes z from x (the normalisation tail of the input-gradient launch, conv3d_mfma.hip EPI 5) gets the forward's z bit for bit
__global__ void k_norm_scale_shift(const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ nss, int G, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G * C) return;
    const int c = i % C;
    const float sc = stats[i * 3 + 1] * gamma[c];
    nss[i * 2] = sc;
    nss[i * 2 + 1] = fmaf(-stats[i * 3], sc, beta[c]);
}
// Keras BatchNormalization moving statistics (momentum m): moving = m * moving + (1 - m) * batch value, the variance fed to the average
// sample-size corrected (x M / (M - 1 - eps), as Keras' fused batch norm does).  One launch instead of the half-dozen element-wise
// launches per layer the host expression took (56 per step of the batch-norm variant, each serialised between two conv launches).
__global__ void k_norm_moving_update(const float* __restrict__ stats, float* __restrict__ mmean, float* __restrict__ mvar, int C, float momentum,
                                     float eps, float corr) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float mean = stats[c * 3], inv = stats[c * 3 + 1];
    const float var = (1.0f / (inv * inv) - eps) * corr;
    mmean[c] = mmean[c] * momentum + mean * (1.0f - momentum);
    mvar[c] = mvar[c] * momentum + var * (1.0f - momentum);
}
// ws[g][c] = {sum dz, sum dz * x} (raw, from the conv epilogue) -> {sum dz, sum dz * xhat}, xhat = (x - mean) * inv: in double, the
// cancellation mean * sum dz against sum dz * x happens once, on the totals
__global__ void k_norm_bwd_center(double* __restrict__ ws, const float* __restrict__ stats, int G, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G * C) return;
    ws[i * 2 + 1] = (ws[i * 2 + 1] - (double)stats[i * 3] * ws[i * 2]) * (double)stats[i * 3 + 1];
}

// y = act((x - mean) * inv * gamma + beta).  A thread keeps its channel group for the whole grid-stride loop whenever the stride is a
// multiple of C/VEC (always, for power-of-two channel counts), so the per-channel scale / shift live in registers and are only rebuilt
// when the loop crosses into another instance (the four scalar loads per element they replace made the kernel instruction-bound).
template <typename T, int VEC>
__global__ void __launch_bounds__(256) k_norm_apply(const T* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
                             const float* __restrict__ beta, T* __restrict__ y, int64_t V, int C, int per_instance, int act, float alpha,
                             int64_t total) {
    const int CG = C / VEC;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const bool fixed = stride % CG == 0;
    float sc[VEC], sh[VEC];
    int have_g = -1, have_cg = -1;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += stride) {
        const int cg = (int)(i % CG);
        const int64_t v = i / CG;
        const int g = per_instance ? (int)(v / V) : 0;
        if (g != have_g || cg != have_cg || !fixed) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const int c = cg * VEC + k;
                const float* st = stats + ((int64_t)g * C + c) * 3;
                sc[k] = st[1] * gamma[c];
                sh[k] = fmaf(-st[0], sc[k], beta[c]);        // one explicit fma: the backward recomputes this shift and must get the same bits
            }
            have_g = g;
            have_cg = cg;
        }
        float xv[VEC], o[VEC];
        ldv<T, VEC>(x + v * C + cg * VEC, xv);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float z = fmaf(xv[k], sc[k], sh[k]);
            if (act == FMRI_ACT_RELU) z = fmaxf(z, 0.f);
            else if (act == FMRI_ACT_LEAKY) z = z > 0.f ? z : alpha * z;
            o[k] = z;
        }
        stv<T, VEC>(y + v * C + cg * VEC, o);
    }
}

// backward reduction: ws[g][c] += { sum dz, sum dz*xhat },  dz = dy * act'(y)
template <typename T>
__global__ void k_norm_bwd_reduce(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ dy, const float* __restrict__ stats,
                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                  double* __restrict__ ws, int64_t V, int C, int per_instance, int act, float alpha, int vchunk) {
    const int c = blockIdx.y * 64 + (threadIdx.x & 63);
    const int g = blockIdx.z;
    const int vl = threadIdx.x >> 6;
    const int64_t v0 = (int64_t)blockIdx.x * vchunk;
    const int64_t v1 = min(V, v0 + vchunk);
    float s = 0.f, q = 0.f;
    if (c < C) {
        const int gi = per_instance ? g : 0;
        const float mean = stats[((int64_t)gi * C + c) * 3], inv = stats[((int64_t)gi * C + c) * 3 + 1];
        const float sc = beta ? inv * gamma[c] : 0.f, sh = beta ? fmaf(-mean, sc, beta[c]) : 0.f;
        const int64_t base = (int64_t)g * V * C + c;
        for (int64_t v = v0 + vl; v < v1; v += 4) {
            float d = to_f<T>(dy[base + v * C]);
            const float yy = beta ? fmaf(to_f<T>(x[base + v * C]), sc, sh) : to_f<T>(y[base + v * C]);
            if (act == FMRI_ACT_RELU) d = yy > 0.f ? d : 0.f;
            else if (act == FMRI_ACT_LEAKY) d = yy > 0.f ? d : alpha * d;
            const float xh = (to_f<T>(x[base + v * C]) - mean) * inv;
            s += d;
            q = fmaf(d, xh, q);
        }
    }
    __shared__ float red[2][4][64];
    red[0][vl][threadIdx.x & 63] = s;
    red[1][vl][threadIdx.x & 63] = q;
    __syncthreads();
    if (threadIdx.x < 64 && c < C) {
        double ds = (double)red[0][0][threadIdx.x] + red[0][1][threadIdx.x] + red[0][2][threadIdx.x] + red[0][3][threadIdx.x];
        double dq = (double)red[1][0][threadIdx.x] + red[1][1][threadIdx.x] + red[1][2][threadIdx.x] + red[1][3][threadIdx.x];
        const int gi = per_instance ? g : 0;
        atomicAdd(&ws[((int64_t)gi * C + c) * 2 + 0], ds);
        atomicAdd(&ws[((int64_t)gi * C + c) * 2 + 1], dq);
    }
}
// dgamma[c] += sum_g ws[g][c][1], dbeta[c] += sum_g ws[g][c][0]; the last reader of the backward sums (launched behind the apply pass)
// leaves them zero for the next reduction
__global__ void k_norm_bwd_params(double* __restrict__ ws, float* __restrict__ dgamma, float* __restrict__ dbeta, int G, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double a = 0, b = 0;
    for (int g = 0; g < G; ++g) {
        double* const w = ws + ((int64_t)g * C + c) * 2;
        b += w[0];
        a += w[1];
        w[0] = 0.0;
        w[1] = 0.0;
    }
    if (dgamma) dgamma[c] += (float)a;
    if (dbeta) dbeta[c] += (float)b;
}
// dx = gamma * [ (dz - mean(dz)) * inv_s - xhat * mean(dz*xhat) * inv_sigma ]  =  a*dz + b*x + c  with per-(instance, channel)
// coefficients a = gamma*inv, b = -gamma*inv*m2*invsig, c = -a*m1 - b*mean, kept in registers as in k_norm_apply
template <typename T, int VEC>
__global__ void k_norm_bwd_apply(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ dy, const float* __restrict__ stats,
                                 const float* __restrict__ gamma, const float* __restrict__ beta, const double* __restrict__ ws,
                                 T* __restrict__ dx, int64_t V, int C, int per_instance, int act, float alpha, double M, int64_t total) {
    const int CG = C / VEC;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const bool fixed = stride % CG == 0;
    const bool fromx = beta != nullptr;                 // sign of the output recomputed from x (see k_norm_bwd_reduce_v)
    float ca[VEC], cb[VEC], cc[VEC], zs[VEC], zh[VEC];
    int have_g = -1, have_cg = -1;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += stride) {
        const int cg = (int)(i % CG);
        const int64_t v = i / CG;
        const int g = per_instance ? (int)(v / V) : 0;
        if (g != have_g || cg != have_cg || !fixed) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const int c = cg * VEC + k;
                const int64_t sc = (int64_t)g * C + c;
                const float mean = stats[sc * 3], inv = stats[sc * 3 + 1], invsig = stats[sc * 3 + 2];
                const float m1 = (float)(ws[sc * 2] / M), m2 = (float)(ws[sc * 2 + 1] / M);
                ca[k] = gamma[c] * inv;
                cb[k] = -gamma[c] * inv * m2 * invsig;
                cc[k] = -ca[k] * m1 - cb[k] * mean;
                zs[k] = inv * gamma[c];
                zh[k] = fromx ? fmaf(-mean, zs[k], beta[c]) : 0.f;
            }
            have_g = g;
            have_cg = cg;
        }
        float xv[VEC], yv[VEC], dv[VEC], o[VEC];
        ldv<T, VEC>(x + v * C + cg * VEC, xv);
        if (!fromx && act != FMRI_ACT_NONE) ldv<T, VEC>(y + v * C + cg * VEC, yv);      // (act none: dy is taken as it is - the `pre` form)
        ldv<T, VEC>(dy + v * C + cg * VEC, dv);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float d = dv[k];
            if (fromx) yv[k] = fmaf(xv[k], zs[k], zh[k]);
            if (act == FMRI_ACT_RELU) d = yv[k] > 0.f ? d : 0.f;
            else if (act == FMRI_ACT_LEAKY) d = yv[k] > 0.f ? d : alpha * d;
            o[k] = fmaf(ca[k], d, fmaf(cb[k], xv[k], cc[k]));
        }
        stv<T, VEC>(dx + v * C + cg * VEC, o);
    }
}

// The same apply pass, sample-local (round 3): workgroup = a voxel range of ONE sample, so the per-(instance, channel) coefficients are
// built once per workgroup - by its threads together, one channel each, through LDS - instead of by every thread for its 8 channels
// whenever its grid-stride walk crossed into another sample (instance norm, 4 x 32x64x64 x 128: every second item; 16 fp64 divisions and
// 48 loads each time - the pass ran at 1.5 TB/s there, 3.4 at level 0).  SRC: 0 the block's output y gives the activation's derivative,
// 1 it is recomputed from x (z = fma(x, a, zh): the scale of z IS the coefficient a), 2 dy is already dz.
template <typename T, int VEC, int SRC>
__global__ void __launch_bounds__(256) k_norm_bwd_apply_s(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ dy,
                                                          const float* __restrict__ stats, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const double* __restrict__ ws, T* __restrict__ dx,
                                                          int64_t V, int C, int per_instance, int act, float alpha, double M, int vchunk) {
    const int CG = C / VEC, VL = 256 / CG;
    const int cg = threadIdx.x % CG, vl = threadIdx.x / CG;
    const int g = blockIdx.z, gi = per_instance ? g : 0;
    const int64_t v0 = (int64_t)blockIdx.x * vchunk, v1 = min(V, v0 + vchunk);
    __shared__ float coef[4][512];
    for (int c = threadIdx.x; c < C; c += 256) {
        const int64_t sc = (int64_t)gi * C + c;
        const float mean = stats[sc * 3], inv = stats[sc * 3 + 1], invsig = stats[sc * 3 + 2];
        const float m1 = (float)(ws[sc * 2] / M), m2 = (float)(ws[sc * 2 + 1] / M);
        const float a = gamma[c] * inv, b = -gamma[c] * inv * m2 * invsig;
        coef[0][c] = a;
        coef[1][c] = b;
        coef[2][c] = -a * m1 - b * mean;
        coef[3][c] = SRC == 1 ? fmaf(-mean, inv * gamma[c], beta[c]) : 0.f;
    }
    __syncthreads();
    float ca[VEC], cb[VEC], cc[VEC], zh[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        ca[k] = coef[0][cg * VEC + k];
        cb[k] = coef[1][cg * VEC + k];
        cc[k] = coef[2][cg * VEC + k];
        zh[k] = coef[3][cg * VEC + k];
    }
    const int64_t base = (int64_t)g * V * C + cg * VEC;
#pragma unroll 4
    for (int64_t v = v0 + vl; v < v1; v += VL) {
        float xv[VEC], yv[VEC], dv[VEC], o[VEC];
        ldv<T, VEC>(x + base + v * C, xv);
        if constexpr (SRC == 0) ldv<T, VEC>(y + base + v * C, yv);
        ldv<T, VEC>(dy + base + v * C, dv);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float d = dv[k];
            if constexpr (SRC == 1) yv[k] = fmaf(xv[k], ca[k], zh[k]);       // (inv * gamma and gamma * inv are the same product)
            if constexpr (SRC != 2) {
                if (act == FMRI_ACT_RELU) d = yv[k] > 0.f ? d : 0.f;
                else if (act == FMRI_ACT_LEAKY) d = yv[k] > 0.f ? d : alpha * d;
            }
            o[k] = fmaf(ca[k], d, fmaf(cb[k], xv[k], cc[k]));
        }
        stv<T, VEC>(dx + base + v * C, o);
    }
}

// ---------------------------------------------------------------------------------------------------- deconvolution k2 s2
// y[n, 2i+a, co] = b[co] + sum_ci x[n,i,ci] * w[a][co][ci]   (a = ad*4+ah*2+aw; 'valid', stride 2: the 8 taps never overlap)
template <typename T>
__global__ void k_deconv_fwd(const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ b, T* __restrict__ y, int N, int D,
                             int H, int W, int Cin, int Cout, int pd) {
    const int D2 = D << pd, H2 = 2 * H, W2 = 2 * W;
    const int64_t total = (int64_t)N * D2 * H2 * W2 * Cout;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % Cout);
        int64_t v = i / Cout;
        const int ww = (int)(v % W2); v /= W2;
        const int hh = (int)(v % H2); v /= H2;
        const int dd = (int)(v % D2);
        const int n = (int)(v / D2);
        const int a = ((pd ? (dd & 1) : 0) << 2) | ((hh & 1) << 1) | (ww & 1);
        const T* xp = x + ((((int64_t)n * D + (dd >> pd)) * H + (hh >> 1)) * W + (ww >> 1)) * Cin;
        const T* wp = w + ((int64_t)a * Cout + co) * Cin;
        float acc = b ? b[co] : 0.f;
        for (int ci = 0; ci < Cin; ++ci) acc = fmaf(to_f<T>(xp[ci]), to_f<T>(wp[ci]), acc);
        y[i] = from_f<T>(acc);
    }
}
// dx[n,i,ci] = sum_a sum_co dy[n,2i+a, dy_off+co] * w[a][co][ci]  (optionally masked by xmask > 0)
template <typename T>
__global__ void k_deconv_dgrad(const T* __restrict__ dy, int dy_ld, int dy_off, const T* __restrict__ w, const T* __restrict__ xmask,
                               T* __restrict__ dx, int N, int D, int H, int W, int Cin, int Cout, int pd) {
    const int D2 = D << pd, H2 = 2 * H, W2 = 2 * W;
    const int64_t total = (int64_t)N * D * H * W * Cin;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin);
        int64_t v = i / Cin;
        const int ww = (int)(v % W); v /= W;
        const int hh = (int)(v % H); v /= H;
        const int dd = (int)(v % D);
        const int n = (int)(v / D);
        float acc = 0.f;
        for (int a = 0; a < (4 << pd); ++a) {
            const int ad = pd ? (a >> 2) : 0, ah = (a >> 1) & 1, aw = a & 1;
            const int wa = (ad << 2) | (ah << 1) | aw;
            const T* gp = dy + ((((int64_t)n * D2 + (dd << pd) + ad) * H2 + 2 * hh + ah) * W2 + 2 * ww + aw) * dy_ld + dy_off;
            const T* wp = w + (int64_t)wa * Cout * Cin + ci;
            for (int co = 0; co < Cout; ++co) acc = fmaf(to_f<T>(gp[co]), to_f<T>(wp[(int64_t)co * Cin]), acc);
        }
        if (xmask && !(to_f<T>(xmask[i]) > 0.f)) acc = 0.f;
        dx[i] = from_f<T>(acc);
    }
}
// dw[a][co][ci] += sum_{n,i} dy[n,2i+a,co] * x[n,i,ci];  db[co] += sum dy.  One block per (a, co): threads over ci, loop voxels.
template <typename T>
__global__ void k_deconv_wgrad(const T* __restrict__ x, const T* __restrict__ dy, int dy_ld, int dy_off, float* __restrict__ dw,
                               float* __restrict__ db, int N, int D, int H, int W, int Cin, int Cout, int pd, int nsplit) {
    const int a = blockIdx.x, co = blockIdx.y, sp = blockIdx.z;
    const int ad = pd ? (a >> 2) : 0, ah = (a >> 1) & 1, aw = a & 1;
    const int D2 = D << pd, H2 = 2 * H, W2 = 2 * W;
    const int64_t nvox = (int64_t)N * D * H * W;
    const int64_t v0 = nvox * sp / nsplit, v1 = nvox * (sp + 1) / nsplit;
    float bsum = 0.f;
    for (int ci0 = 0; ci0 < Cin; ci0 += blockDim.x) {
        const int ci = ci0 + threadIdx.x;
        float acc = 0.f;
        for (int64_t v = v0; v < v1; ++v) {
            int64_t q = v;
            const int ww = (int)(q % W); q /= W;
            const int hh = (int)(q % H); q /= H;
            const int dd = (int)(q % D);
            const int n = (int)(q / D);
            const float g = to_f<T>(dy[((((int64_t)n * D2 + (dd << pd) + ad) * H2 + 2 * hh + ah) * W2 + 2 * ww + aw) * dy_ld + dy_off + co]);
            if (ci < Cin) acc = fmaf(g, to_f<T>(x[v * Cin + ci]), acc);
            if (ci0 == 0 && threadIdx.x == 0) bsum += g;
        }
        if (ci < Cin) atomicAdd(&dw[((int64_t)a * Cout + co) * Cin + ci], acc);
    }
    if (db && threadIdx.x == 0) atomicAdd(&db[co], bsum);
}

}  // namespace

int norm_ws_zero(double* ws, int n, hipStream_t s) {
    if (!ws || n <= 0) return FMRI_E_SHAPE;
    k_zero_d<<<(n + 255) / 256, 256, 0, s>>>(ws, n);
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
// ws[0][i] = sum over the slots s = 1 .. nslot of ws[s][i], i < n (the per-workgroup partial sums of a conv launch's normalisation tail)
__global__ void k_ws_fold(double* __restrict__ ws, int nslot, int n) {
    const int i = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;       // 4 slot ranges per entry
    __shared__ double red[4][64];
    double a = 0.0;
    if (i < n)
        for (int s = 1 + part; s <= nslot; s += 4) {
            a += ws[(int64_t)s * n + i];
            ws[(int64_t)s * n + i] = 0.0;            // the scratch is left zero (the slots overlap other layers' [G][C][2] blocks)
        }
    red[part][threadIdx.x & 63] = a;
    __syncthreads();
    if (part == 0 && i < n) ws[i] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
int norm_ws_fold(double* ws, int nslot, int n, hipStream_t s) {
    if (!ws || n <= 0 || nslot <= 0) return FMRI_E_SHAPE;
    k_ws_fold<<<(n + 63) / 64, 256, 0, s>>>(ws, nslot, n);
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

// -------------------------------------------------------------------------------------------------------------- C ABI
// have_sums: ws already holds {sum x, sum x^2} per (group, channel) - the conv in front summed its own output (fmri_conv3d_fwd_stats)
static int norm_act_fwd_impl(const void* x, const float* gamma, const float* beta, void* y, float* stats, double* ws, int N,
                             int64_t V, int C, int per_instance, float eps, int eps_on_std, int act, float alpha, int dtype,
                             bool have_sums, fmri_stream_t stream) {
    if (N <= 0 || V <= 0 || C <= 0 || !stats || !ws) return FMRI_E_SHAPE;
    hipStream_t s = as_stream(stream);
    if (dtype != FMRI_F32 && dtype != FMRI_BF16) return FMRI_E_DTYPE;
    if (per_instance >= 0) {            // per_instance < 0: inference with the statistics already in `stats` (moving averages)
        const int G = per_instance ? N : 1;
        if (!have_sums) {
        const int vchunk = norm_vchunk(V, N, per_instance, C);
        dim3 grid((unsigned)ceil_div64(V, vchunk), (C + 63) / 64, N);
        if (norm_vec_ok(C, dtype))
            k_norm_reduce_v<bf16_t, 8><<<dim3(grid.x, 1, N), 256, 0, s>>>((const bf16_t*)x, ws, V, C, per_instance, vchunk);
        else if (dtype == FMRI_F32) k_norm_reduce<float><<<grid, 256, 0, s>>>((const float*)x, ws, V, C, per_instance, vchunk);
        else k_norm_reduce<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)x, ws, V, C, per_instance, vchunk);
        }
        const double M = per_instance ? (double)V : (double)V * N;
        k_norm_finalize<<<(G * C + 255) / 256, 256, 0, s>>>(ws, stats, G, C, M, eps, eps_on_std);
    } else {
        per_instance = 0;
    }
    const int vec = pick_vec(C);
    const int64_t total = (int64_t)N * V * (C / vec);
    const int g2 = grid_for(total);
    if (dtype == FMRI_F32)
        LAUNCH_TV(k_norm_apply, float, vec, g2, 256, s, (const float*)x, stats, gamma, beta, (float*)y, V, C, per_instance, act, alpha, total);
    else
        LAUNCH_TV(k_norm_apply, bf16_t, vec, g2, 256, s, (const bf16_t*)x, stats, gamma, beta, (bf16_t*)y, V, C, per_instance, act, alpha, total);
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
extern "C" int fmri_norm_act_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, double* ws, int N,
                                 int64_t V, int C, int per_instance, float eps, int eps_on_std, int act, float alpha, int dtype,
                                 fmri_stream_t stream) {
    return norm_act_fwd_impl(x, gamma, beta, y, stats, ws, N, V, C, per_instance, eps, eps_on_std, act, alpha, dtype, false, stream);
}
// as fmri_norm_act_fwd with the reduction pass already done: ws[g][C][2] = {sum x, sum x^2} as left by fmri_conv3d_fwd_stats /
// fmri_conv3d_upcat_fwd_stats (training statistics only: per_instance 0 | 1)
extern "C" int fmri_norm_act_fwd_pre(const void* x, const float* gamma, const float* beta, void* y, float* stats, double* ws, int N,
                                     int64_t V, int C, int per_instance, float eps, int eps_on_std, int act, float alpha, int dtype,
                                     fmri_stream_t stream) {
    if (per_instance < 0) return FMRI_E_SHAPE;
    return norm_act_fwd_impl(x, gamma, beta, y, stats, ws, N, V, C, per_instance, eps, eps_on_std, act, alpha, dtype, true, stream);
}
extern "C" int fmri_norm_moving_update(const float* stats, float* moving_mean, float* moving_var, int C, double M, float momentum, float eps,
                                       fmri_stream_t stream) {
    if (!stats || !moving_mean || !moving_var || C <= 0 || M <= 0) return FMRI_E_SHAPE;
    const double den = M - (1.0 + (double)eps);
    k_norm_moving_update<<<(C + 255) / 256, 256, 0, as_stream(stream)>>>(stats, moving_mean, moving_var, C, momentum, eps, (float)(M / (den > 1.0 ? den : 1.0)));
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
extern "C" int fmri_norm_scale_shift(const float* stats, const float* gamma, const float* beta, float* nss, int G, int C, fmri_stream_t stream) {
    if (!stats || !gamma || !beta || !nss || G <= 0 || C <= 0) return FMRI_E_SHAPE;
    k_norm_scale_shift<<<(G * C + 255) / 256, 256, 0, as_stream(stream)>>>(stats, gamma, beta, nss, G, C);
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}

// pre: dy is already dz = dy * act'(z) and ws holds the raw sums {sum dz, sum dz * x} (fmri_conv3d_dgrad_norm)
static int norm_act_bwd_impl(const void* x, const void* y, const void* dy, const float* gamma, const float* beta, const float* stats, void* dx,
                             float* dgamma, float* dbeta, double* ws, int N, int64_t V, int C, int per_instance, int act,
                             float alpha, int dtype, fmri_stream_t stream, bool pre = false) {
    if (N <= 0 || V <= 0 || C <= 0 || !stats || !ws || (!y && !beta && !pre)) return FMRI_E_SHAPE;
    hipStream_t s = as_stream(stream);
    const int G = per_instance ? N : 1;
    const int vchunk = norm_vchunk(V, N, per_instance, C);
    dim3 grid((unsigned)ceil_div64(V, vchunk), (C + 63) / 64, N);
    if (pre) {
        if (dtype != FMRI_F32 && dtype != FMRI_BF16) return FMRI_E_DTYPE;
        k_norm_bwd_center<<<(G * C + 255) / 256, 256, 0, s>>>(ws, stats, G, C);
        act = FMRI_ACT_NONE;           // the apply pass takes dz as it is
        beta = nullptr;
        y = x;                         // (never read with act == none)
    }
    if (pre) {
    } else if (norm_vec_ok(C, dtype)) {
        if (beta)
            k_norm_bwd_reduce_v<bf16_t, 8, true><<<dim3(grid.x, 1, N), 256, 0, s>>>((const bf16_t*)x, (const bf16_t*)y, (const bf16_t*)dy, stats, gamma,
                                                                                      beta, ws, V, C, per_instance, act, alpha, vchunk);
        else
            k_norm_bwd_reduce_v<bf16_t, 8, false><<<dim3(grid.x, 1, N), 256, 0, s>>>((const bf16_t*)x, (const bf16_t*)y, (const bf16_t*)dy, stats, gamma,
                                                                                       beta, ws, V, C, per_instance, act, alpha, vchunk);
    } else if (dtype == FMRI_F32)
        k_norm_bwd_reduce<float><<<grid, 256, 0, s>>>((const float*)x, (const float*)y, (const float*)dy, stats, gamma, beta, ws, V, C, per_instance, act, alpha, vchunk);
    else if (dtype == FMRI_BF16)
        k_norm_bwd_reduce<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)x, (const bf16_t*)y, (const bf16_t*)dy, stats, gamma, beta, ws, V, C, per_instance, act, alpha, vchunk);
    else return FMRI_E_DTYPE;
    const double M = per_instance ? (double)V : (double)V * N;
    const int vec = pick_vec(C);
    const int64_t total = (int64_t)N * V * (C / vec);
    const int g2 = grid_for(total);
    if (norm_vec_ok(C, dtype)) {
        const dim3 ga((unsigned)ceil_div64(V, vchunk), 1, N);
#define FMRI_APPLY_S(SRC_) k_norm_bwd_apply_s<bf16_t, 8, SRC_><<<ga, 256, 0, s>>>((const bf16_t*)x, (const bf16_t*)y, (const bf16_t*)dy, stats, gamma, \
                                                                                beta, ws, (bf16_t*)dx, V, C, per_instance, act, alpha, M, vchunk)
        if (pre || act == FMRI_ACT_NONE) FMRI_APPLY_S(2);
        else if (beta) FMRI_APPLY_S(1);
        else FMRI_APPLY_S(0);
#undef FMRI_APPLY_S
    } else if (dtype == FMRI_F32)
        LAUNCH_TV(k_norm_bwd_apply, float, vec, g2, 256, s, (const float*)x, (const float*)y, (const float*)dy, stats, gamma, beta, ws, (float*)dx, V, C, per_instance, act, alpha, M, total);
    else
        LAUNCH_TV(k_norm_bwd_apply, bf16_t, vec, g2, 256, s, (const bf16_t*)x, (const bf16_t*)y, (const bf16_t*)dy, stats, gamma, beta, ws, (bf16_t*)dx, V, C, per_instance, act, alpha, M, total);
    k_norm_bwd_params<<<(C + 255) / 256, 256, 0, s>>>(ws, dgamma, dbeta, G, C);          // parameter gradients; zeroes ws behind its last reader
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
extern "C" int fmri_norm_act_bwd(const void* x, const void* y, const void* dy, const float* gamma, const float* stats, void* dx,
                                 float* dgamma, float* dbeta, double* ws, int N, int64_t V, int C, int per_instance, int act,
                                 float alpha, int dtype, fmri_stream_t stream) {
    if (!y) return FMRI_E_SHAPE;
    return norm_act_bwd_impl(x, y, dy, gamma, nullptr, stats, dx, dgamma, dbeta, ws, N, V, C, per_instance, act, alpha, dtype, stream);
}
extern "C" int fmri_norm_act_bwd_x(const void* x, const void* dy, const float* gamma, const float* beta, const float* stats, void* dx,
                                   float* dgamma, float* dbeta, double* ws, int N, int64_t V, int C, int per_instance, int act,
                                   float alpha, int dtype, fmri_stream_t stream) {
    if (!beta) return FMRI_E_SHAPE;
    return norm_act_bwd_impl(x, nullptr, dy, gamma, beta, stats, dx, dgamma, dbeta, ws, N, V, C, per_instance, act, alpha, dtype, stream);
}
// the rest of the normalisation's backward pass behind fmri_conv3d_dgrad_norm: dz (activation derivative applied) and ws = {sum dz, sum dz * x}
// come from that launch's epilogue; parameter gradients and dx = gamma * [(dz - mean(dz)) / s - xhat * mean(dz * xhat) / sigma] remain
extern "C" int fmri_norm_act_bwd_pre(const void* x, const void* dz, const float* gamma, const float* stats, void* dx, float* dgamma, float* dbeta,
                                     double* ws, int N, int64_t V, int C, int per_instance, int dtype, fmri_stream_t stream) {
    if (!x || !dz || !dx) return FMRI_E_SHAPE;
    return norm_act_bwd_impl(x, nullptr, dz, gamma, nullptr, stats, dx, dgamma, dbeta, ws, N, V, C, per_instance, FMRI_ACT_NONE, 0.f, dtype, stream, true);
}

extern "C" int fmri_deconv3d_k2s2_fwd(const void* x, const void* w, const float* b, void* y, int N, int D, int H, int W, int Cin, int Cout,
                                      int dtype, int planar, fmri_stream_t stream) {
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return FMRI_E_SHAPE;
    const int pd = planar ? 0 : 1;
    const int grid = grid_for((int64_t)N * (D << pd) * 2 * H * 2 * W * Cout, 256, 8192);
    hipStream_t s = as_stream(stream);
    if (dtype == FMRI_F32) k_deconv_fwd<float><<<grid, 256, 0, s>>>((const float*)x, (const float*)w, b, (float*)y, N, D, H, W, Cin, Cout, pd);
    else if (dtype == FMRI_BF16) k_deconv_fwd<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)x, (const bf16_t*)w, b, (bf16_t*)y, N, D, H, W, Cin, Cout, pd);
    else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
extern "C" int fmri_deconv3d_k2s2_bwd(const void* x, const void* w, const void* dy, int dy_ld, int dy_off, const void* xmask, void* dx,
                                      float* dw, float* db, int N, int D, int H, int W, int Cin, int Cout, int dtype, int planar,
                                      fmri_stream_t stream) {
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || dy_ld < dy_off + Cout) return FMRI_E_SHAPE;
    const int pd = planar ? 0 : 1;
    hipStream_t s = as_stream(stream);
    const int grid = grid_for((int64_t)N * D * H * W * Cin, 256, 8192);
    const int nsplit = 16;
    dim3 gw(4 << pd, Cout, nsplit);
    const int bt = Cin >= 256 ? 256 : (Cin >= 128 ? 128 : 64);
    if (dtype == FMRI_F32) {
        if (dx) k_deconv_dgrad<float><<<grid, 256, 0, s>>>((const float*)dy, dy_ld, dy_off, (const float*)w, (const float*)xmask, (float*)dx, N, D, H, W, Cin, Cout, pd);
        if (dw) k_deconv_wgrad<float><<<gw, bt, 0, s>>>((const float*)x, (const float*)dy, dy_ld, dy_off, dw, db, N, D, H, W, Cin, Cout, pd, nsplit);
    } else if (dtype == FMRI_BF16) {
        if (dx) k_deconv_dgrad<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)dy, dy_ld, dy_off, (const bf16_t*)w, (const bf16_t*)xmask, (bf16_t*)dx, N, D, H, W, Cin, Cout, pd);
        if (dw) k_deconv_wgrad<bf16_t><<<gw, bt, 0, s>>>((const bf16_t*)x, (const bf16_t*)dy, dy_ld, dy_off, dw, db, N, D, H, W, Cin, Cout, pd, nsplit);
    } else return FMRI_E_DTYPE;
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
