// ao_amd/csrc/gva_common.h -- shared pieces of the fused grouped-vector-attention kernels.
#pragma once
#include "common.h"

namespace gva {

constexpr int TPB = 256;
constexpr int WPB = TPB / WAVE;  // waves per block

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

constexpr int MAX_BLOCKS = 256 * 8;   // grid cap of the row-parallel stages
constexpr int MAX_PARAM_BLOCKS = 256; // grid cap of the channel-parallel parameter-gradient stage

// floats of per-block partial sums any stage may write (the workspace's first region)
inline size_t part_floats(int c, int g) {
    size_t logits_fwd = (size_t)MAX_BLOCKS * 2 * g;
    size_t logits_bwd = (size_t)MAX_BLOCKS * g + (size_t)MAX_PARAM_BLOCKS * c * (g + 4);
    size_t agg_bwd = (size_t)MAX_BLOCKS * (3 * (size_t)g + (size_t)g * g + 4 * (size_t)c);
    size_t m = logits_fwd > logits_bwd ? logits_fwd : logits_bwd;
    m = m > agg_bwd ? m : agg_bwd;
    return m < 9 * (size_t)MAX_BLOCKS ? 9 * (size_t)MAX_BLOCKS : m;
}
inline size_t rows_offset_bytes(int c, int g) { return align_up(sizeof(float) * part_floats(c, g)); }

// masked relative position of neighbour slot (n, s)
struct Rel {
    float x, y, z;
    int src;  // neighbour index or -1
};

__device__ __forceinline__ Rel rel_pos(const float *__restrict__ coord, const int *__restrict__ idx, long long row,
                                       int n) {
    Rel r;
    r.src = idx[row];
    r.x = r.y = r.z = 0.f;
    if (r.src >= 0) {
        r.x = coord[3 * (long long)r.src] - coord[3 * (long long)n];
        r.y = coord[3 * (long long)r.src + 1] - coord[3 * (long long)n + 1];
        r.z = coord[3 * (long long)r.src + 2] - coord[3 * (long long)n + 2];
    }
    return r;
}

// P = ReLU(a . pos + b): the folded Linear(3,C) -> BatchNorm -> ReLU of linear_p_bias
__device__ __forceinline__ float pe_act(float ax, float ay, float az, float b, float px, float py, float pz) {
    return fmaxf(__builtin_fmaf(az, pz, __builtin_fmaf(ay, py, __builtin_fmaf(ax, px, b))), 0.f);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

// Sum `len` per-block partial vectors (float) into double or float outputs: out[j] = sum_b part[b*len + j].
template <typename OutT>
__global__ void reduce_partials_kernel(const float *__restrict__ part, int nblk, int len, OutT *__restrict__ out) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= len) return;
    double acc = 0.0;
    for (int b = 0; b < nblk; ++b) acc += (double)part[(size_t)b * len + j];
    out[j] = (OutT)acc;
}

// Same, split into two outputs: columns [0,len1) -> out1, [len1, len1+len2) -> out2.
template <typename OutT>
__global__ void reduce_partials2_kernel(const float *__restrict__ part, int nblk, int len1, int len2,
                                        OutT *__restrict__ out1, OutT *__restrict__ out2) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int len = len1 + len2;
    if (j >= len) return;
    double acc = 0.0;
    for (int b = 0; b < nblk; ++b) acc += (double)part[(size_t)b * len + j];
    if (j < len1) out1[j] = (OutT)acc; else out2[j - len1] = (OutT)acc;
}

}  // namespace gva
