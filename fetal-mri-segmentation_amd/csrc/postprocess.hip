// Post-processing of a predicted probability volume on the device (reference fetal_net/postprocess.py:7-19):
//   scipy.ndimage.gaussian_filter -> "> threshold" -> binary_fill_holes -> largest connected component (scipy.ndimage.label).
// The production flow of the reference (prod/predict_nifti2.py:77-95) runs this once per stage on a whole 160x256x256-class volume; with
// the sliding-window result already in HBM it costs four small HBM-bound passes and two label-propagation loops instead of a download
// plus ~1 s of single-threaded scipy.  All kernels are byte / index movers: one thread per voxel, z (contiguous) fastest.
//
// Exactness: the 1-D correlation sums in scipy's own order (centre tap first, then the symmetric pairs from the outermost inwards,
// ni_filters.c NI_Correlate1D symmetric branch) in fp64 with host-computed weights, so the smoothed volume - and therefore the
// thresholded mask - is bit-identical to scipy's; hole filling and labelling are integer algorithms with scipy's default 6-connectivity,
// and ties between equally large components go to the one scipy numbers first (smallest linear index of its first voxel).
#include "common.h"

namespace {

__device__ __forceinline__ int reflect_idx(int i, int n) {          // scipy mode='reflect': d c b a | a b c d | d c b a
    while (i < 0 || i >= n) i = i < 0 ? -i - 1 : 2 * n - i - 1;
    return i;
}

// T = double: the post-processing volume (bit-identical to scipy).  T = float: a training patch (skimage.filters.gaussian of the
// augmentation chain, reference augment.py:113-114): the same sums in fp64, the result rounded once to the patch's fp32.
// NEAREST = scipy mode 'nearest' (a a a a | a b c d | d d d d), what skimage.filters.gaussian passes; else 'reflect'.
template <typename T, bool NEAREST>
__global__ void k_correlate1d_sym(const T* __restrict__ src, T* __restrict__ dst, int X, int Y, int Z, int axis,
                                  const double* __restrict__ w, int radius) {
#pragma clang fp contract(off)      // separately rounded multiply and add: scipy's C loop is compiled without fused operations
    const int64_t total = (int64_t)X * Y * Z;
    const int n = axis == 0 ? X : (axis == 1 ? Y : Z);
    const int64_t stride = axis == 0 ? (int64_t)Y * Z : (axis == 1 ? Z : 1);
    auto at = [&](int i) { return NEAREST ? min(max(i, 0), n - 1) : reflect_idx(i, n); };
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int z = (int)(t % Z);
        const int64_t q = t / Z;
        const int y = (int)(q % Y), x = (int)(q / Y);
        const int l = axis == 0 ? x : (axis == 1 ? y : z);
        const int64_t base = t - (int64_t)l * stride;
        double acc = (double)src[t] * w[radius];
        for (int jj = -radius; jj < 0; ++jj)
            acc += ((double)src[base + (int64_t)at(l + jj) * stride] + (double)src[base + (int64_t)at(l - jj) * stride]) * w[jj + radius];
        dst[t] = (T)acc;
    }
}

__global__ void k_threshold(const double* __restrict__ src, uint8_t* __restrict__ dst, int64_t n, double thr) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i] > thr ? 1 : 0;
}

// ---- binary_fill_holes = NOT (background reachable from outside the volume, 6-connectivity)
__global__ void k_flood_init(const uint8_t* __restrict__ mask, uint8_t* __restrict__ reached, int X, int Y, int Z) {
    const int64_t total = (int64_t)X * Y * Z;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int z = (int)(t % Z);
        const int64_t q = t / Z;
        const int y = (int)(q % Y), x = (int)(q / Y);
        const bool border = x == 0 || y == 0 || z == 0 || x == X - 1 || y == Y - 1 || z == Z - 1;
        reached[t] = (!mask[t] && border) ? 1 : 0;
    }
}
// one sweep: every unreached background voxel with a reached 6-neighbour becomes reached; then the front runs on along +z / -z inside
// the thread's own row as far as it can (rows are contiguous: the long axis costs one sweep instead of Z)
__global__ void k_flood_sweep(const uint8_t* __restrict__ mask, uint8_t* __restrict__ reached, int X, int Y, int Z, int* __restrict__ changed) {
    const int64_t total = (int64_t)X * Y * Z;
    bool any = false;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        if (mask[t] || reached[t]) continue;
        const int z = (int)(t % Z);
        const int64_t q = t / Z;
        const int y = (int)(q % Y), x = (int)(q / Y);
        const int64_t sx = (int64_t)Y * Z;
        const bool hit = (x > 0 && reached[t - sx]) || (x < X - 1 && reached[t + sx]) || (y > 0 && reached[t - Z]) || (y < Y - 1 && reached[t + Z]) ||
                         (z > 0 && reached[t - 1]) || (z < Z - 1 && reached[t + 1]);
        if (!hit) continue;
        reached[t] = 1;
        any = true;
        for (int k = z + 1; k < Z && !mask[t - z + k] && !reached[t - z + k]; ++k) reached[t - z + k] = 1;
        for (int k = z - 1; k >= 0 && !mask[t - z + k] && !reached[t - z + k]; --k) reached[t - z + k] = 1;
    }
    if (any) *changed = 1;
}
__global__ void k_fill_from_reached(const uint8_t* __restrict__ reached, uint8_t* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = reached[i] ? 0 : 1;
}

// ---- connected components (6-connectivity) by min-label propagation with root chasing: label = 1 + linear index of the component's
// first voxel once converged - the order scipy.ndimage.label numbers the components in
__global__ void k_cc_init(const uint8_t* __restrict__ mask, int32_t* __restrict__ lab, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) lab[i] = mask[i] ? (int32_t)(i + 1) : 0;
}
__global__ void k_cc_sweep(int32_t* __restrict__ lab, int X, int Y, int Z, int* __restrict__ changed) {
    const int64_t total = (int64_t)X * Y * Z;
    bool any = false;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int32_t mine = lab[t];
        if (!mine) continue;
        const int z = (int)(t % Z);
        const int64_t q = t / Z;
        const int y = (int)(q % Y), x = (int)(q / Y);
        const int64_t sx = (int64_t)Y * Z;
        int32_t m = mine;
        auto look = [&](int64_t o) { const int32_t v = lab[o]; if (v && v < m) m = v; };
        if (x > 0) look(t - sx);
        if (x < X - 1) look(t + sx);
        if (y > 0) look(t - Z);
        if (y < Y - 1) look(t + Z);
        if (z > 0) look(t - 1);
        if (z < Z - 1) look(t + 1);
        // chase the chain of representatives: the label of voxel (m - 1) is never larger than m and belongs to the same component
        for (int hop = 0; hop < 8; ++hop) {
            const int32_t up = lab[m - 1];
            if (up == m) break;
            m = up;
        }
        if (m < mine) {
            atomicMin(&lab[t], m);
            atomicMin(&lab[mine - 1], m);          // hand the better label to the old representative too (union by smaller index)
            any = true;
        }
    }
    if (any) *changed = 1;
}
__global__ void k_cc_count(const int32_t* __restrict__ lab, int32_t* __restrict__ counts, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (lab[i]) atomicAdd(&counts[lab[i]], 1);
}
// best = (largest count, smallest label among equals), packed as (count << 32) | (0xffffffff - label) so that one 64-bit max does both
__global__ void k_cc_best(const int32_t* __restrict__ counts, int64_t n, unsigned long long* __restrict__ best) {
    unsigned long long mine = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; i <= n; i += (int64_t)gridDim.x * blockDim.x)
        if (counts[i] > 0) {
            const unsigned long long key = ((unsigned long long)(unsigned)counts[i] << 32) | (unsigned long long)(0xffffffffu - (unsigned)i);
            if (key > mine) mine = key;
        }
    if (mine) atomicMax(best, mine);
}
__global__ void k_cc_select(const int32_t* __restrict__ lab, const unsigned long long* __restrict__ best, uint8_t* __restrict__ out, int64_t n) {
    const unsigned long long b = *best;
    const int32_t want = b ? (int32_t)(0xffffffffu - (unsigned)(b & 0xffffffffull)) : -1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = lab[i] == want ? 1 : 0;
}

}  // namespace

extern "C" int fmri_correlate1d_f64(const double* src, double* dst, int X, int Y, int Z, int axis, const double* weights, int radius,
                                    fmri_stream_t stream) {
    if (!src || !dst || !weights || src == dst || X <= 0 || Y <= 0 || Z <= 0 || axis < 0 || axis > 2 || radius < 0) return FMRI_E_SHAPE;
    k_correlate1d_sym<double, false><<<grid_for((int64_t)X * Y * Z, 256, 8192), 256, 0, as_stream(stream)>>>(src, dst, X, Y, Z, axis, weights, radius);
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
extern "C" int fmri_correlate1d_f32(const float* src, float* dst, int X, int Y, int Z, int axis, const double* weights, int radius, int mode,
                                    fmri_stream_t stream) {
    if (!src || !dst || !weights || src == dst || X <= 0 || Y <= 0 || Z <= 0 || axis < 0 || axis > 2 || radius < 0 || (mode != 0 && mode != 1))
        return FMRI_E_SHAPE;
    const int grid = grid_for((int64_t)X * Y * Z, 256, 8192);
    if (mode == 1) k_correlate1d_sym<float, true><<<grid, 256, 0, as_stream(stream)>>>(src, dst, X, Y, Z, axis, weights, radius);
    else k_correlate1d_sym<float, false><<<grid, 256, 0, as_stream(stream)>>>(src, dst, X, Y, Z, axis, weights, radius);
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
extern "C" int fmri_threshold_f64(const double* src, uint8_t* dst, int64_t n, double threshold, fmri_stream_t stream) {
    if (!src || !dst || n <= 0) return FMRI_E_SHAPE;
    k_threshold<<<grid_for(n, 256, 8192), 256, 0, as_stream(stream)>>>(src, dst, n, threshold);
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
// phase 0: reached := background voxels on the volume's border; phase 1: `sweeps` propagation sweeps (sets *changed when a sweep still
// grew the region - the caller repeats until it stays 0); phase 2: out := NOT reached (the filled mask)
extern "C" int fmri_fill_holes_step(const uint8_t* mask, uint8_t* reached, uint8_t* out, int X, int Y, int Z, int phase, int sweeps, int* changed,
                                    fmri_stream_t stream) {
    if (!mask || !reached || X <= 0 || Y <= 0 || Z <= 0) return FMRI_E_SHAPE;
    const int64_t n = (int64_t)X * Y * Z;
    hipStream_t st = as_stream(stream);
    const int grid = grid_for(n, 256, 8192);
    if (phase == 0) k_flood_init<<<grid, 256, 0, st>>>(mask, reached, X, Y, Z);
    else if (phase == 1) {
        if (!changed) return FMRI_E_SHAPE;
        for (int i = 0; i < sweeps; ++i) k_flood_sweep<<<grid, 256, 0, st>>>(mask, reached, X, Y, Z, changed);
    } else {
        if (!out) return FMRI_E_SHAPE;
        k_fill_from_reached<<<grid, 256, 0, st>>>(reached, out, n);
    }
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
// phase 0: labels := 1 + linear index on the foreground; phase 1: `sweeps` propagation sweeps (*changed as above); phase 2: out := the
// largest component (counts: int32 [n + 1] scratch, best: uint64 scratch - both zeroed here)
extern "C" int fmri_largest_component_step(const uint8_t* mask, int32_t* labels, int32_t* counts, unsigned long long* best, uint8_t* out, int X,
                                           int Y, int Z, int phase, int sweeps, int* changed, fmri_stream_t stream) {
    if (!labels || X <= 0 || Y <= 0 || Z <= 0) return FMRI_E_SHAPE;
    const int64_t n = (int64_t)X * Y * Z;
    if (n >= 0x7fffffff) return FMRI_E_SHAPE;
    hipStream_t st = as_stream(stream);
    const int grid = grid_for(n, 256, 8192);
    if (phase == 0) {
        if (!mask) return FMRI_E_SHAPE;
        k_cc_init<<<grid, 256, 0, st>>>(mask, labels, n);
    } else if (phase == 1) {
        if (!changed) return FMRI_E_SHAPE;
        for (int i = 0; i < sweeps; ++i) k_cc_sweep<<<grid, 256, 0, st>>>(labels, X, Y, Z, changed);
    } else {
        if (!counts || !best || !out) return FMRI_E_SHAPE;
        if (hipMemsetAsync(counts, 0, (size_t)(n + 1) * sizeof(int32_t), st) != hipSuccess) return FMRI_E_LAUNCH;
        if (hipMemsetAsync(best, 0, sizeof(unsigned long long), st) != hipSuccess) return FMRI_E_LAUNCH;
        k_cc_count<<<grid, 256, 0, st>>>(labels, counts, n);
        k_cc_best<<<grid, 256, 0, st>>>(counts, n, best);
        k_cc_select<<<grid, 256, 0, st>>>(labels, best, out, n);
    }
    FMRI_LAUNCH_CHECK();
    return FMRI_OK;
}
