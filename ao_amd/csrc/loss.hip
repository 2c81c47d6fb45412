// ao_amd/csrc/loss.hip -- softmax cross-entropy of the segmentation head (pointcept/models/default.py:239-251 wraps
// nn.CrossEntropyLoss(ignore_index=-1), mean over the labelled points) for (N, C <= 64) fp32 logits.
// The stock path is log_softmax + a single-workgroup nll reduction (142 us forward + 87 us backward at N = 120 k,
// C = 13); here: one lane per point, logits row in registers, per-block partial (sum, count) finished by the last
// block to arrive; the backward writes (softmax - onehot) * g / count in one pass.
#include "gva_common.h"

namespace {

constexpr int CE_TPB = 256;

__global__ __launch_bounds__(CE_TPB) void ce_forward_kernel(int n, int c, const float *__restrict__ logits,
                                                            const long long *__restrict__ label, int ignore_index,
                                                            float *__restrict__ lse, float *part, unsigned *counter,
                                                            float *__restrict__ loss, float *__restrict__ count_out,
                                                            float *__restrict__ bad_out) {
    __shared__ float s_sum[CE_TPB / 64], s_cnt[CE_TPB / 64], s_bad[CE_TPB / 64];
    float sum = 0.f, cnt = 0.f, bad = 0.f;
    for (long long i = (long long)blockIdx.x * CE_TPB + threadIdx.x; i < n; i += (long long)gridDim.x * CE_TPB) {
        const float *row = logits + i * c;
        float mx = row[0];
        for (int j = 1; j < c; ++j) mx = fmaxf(mx, row[j]);
        float se = 0.f;
        for (int j = 0; j < c; ++j) se += expf(row[j] - mx);
        const float l = mx + logf(se);
        lse[i] = l;
        const long long y = label[i];
        if (y != ignore_index) {
            if (y >= 0 && y < c) { sum += l - row[y]; cnt += 1.f; }
            else bad += 1.f;  // torch device-asserts on such a label; here it poisons the loss (below)
        }
    }
    sum = gva::wave_sum(sum);
    cnt = gva::wave_sum(cnt);
    bad = gva::wave_sum(bad);
    if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = sum; s_cnt[threadIdx.x >> 6] = cnt; s_bad[threadIdx.x >> 6] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, b = 0.f, d = 0.f;
        for (int w = 0; w < CE_TPB / 64; ++w) { a += s_sum[w]; b += s_cnt[w]; d += s_bad[w]; }
        gva::part_store(part + 3 * blockIdx.x, a);
        gva::part_store(part + 3 * blockIdx.x + 1, b);
        gva::part_store(part + 3 * blockIdx.x + 2, d);
    }
    if (gva::last_block_arrives(counter)) {
        // all 256 threads of the last block share the records (one thread alone walked ~470 x 3 dependent loads: 45 of this
        // kernel's 56 us at 120 k points); strided partial sums, then a fixed-shape tree in LDS: bitwise reproducible
        __shared__ double s_a[CE_TPB], s_b[CE_TPB], s_d[CE_TPB];
        double a = 0.0, b = 0.0, d = 0.0;
        for (unsigned k = threadIdx.x; k < gridDim.x; k += CE_TPB) {
            a += (double)gva::part_load(part + 3 * k); b += (double)gva::part_load(part + 3 * k + 1);
            d += (double)gva::part_load(part + 3 * k + 2);
        }
        s_a[threadIdx.x] = a; s_b[threadIdx.x] = b; s_d[threadIdx.x] = d;
        __syncthreads();
        for (int s = CE_TPB / 2; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) {
                s_a[threadIdx.x] += s_a[threadIdx.x + s]; s_b[threadIdx.x] += s_b[threadIdx.x + s];
                s_d[threadIdx.x] += s_d[threadIdx.x + s];
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            a = s_a[0]; b = s_b[0]; d = s_d[0];
            // 0 / 0 = nan for a batch without labelled points, as torch; a label outside [0, c) that is not ignore_index
            // makes the loss nan too: the run fails visibly at the first loss read instead of training on fewer points
            *loss = d > 0.0 ? __builtin_nanf("") : (float)(a / b);
            *count_out = (float)b;
            *bad_out = (float)d;
        }
    }
}

__global__ __launch_bounds__(CE_TPB) void ce_backward_kernel(int n, int c, const float *__restrict__ logits,
                                                             const long long *__restrict__ label, int ignore_index,
                                                             const float *__restrict__ lse, const float *__restrict__ g_loss,
                                                             const float *__restrict__ count, float *__restrict__ g_logits) {
    const float cnt = *count;
    const float scale = cnt > 0.f ? *g_loss / cnt : 0.f;
    for (long long i = (long long)blockIdx.x * CE_TPB + threadIdx.x; i < n; i += (long long)gridDim.x * CE_TPB) {
        const float *row = logits + i * c;
        float *out = g_logits + i * c;
        const long long y = label[i];
        const bool on = y != ignore_index && y >= 0 && y < c;
        const float l = lse[i];
        for (int j = 0; j < c; ++j) out[j] = on ? (expf(row[j] - l) - (j == y ? 1.f : 0.f)) * scale : 0.f;
    }
}

}  // namespace

extern "C" size_t cross_entropy_workspace_bytes(int n) {
    return sizeof(float) * 3 * (size_t)std::min<long long>(((long long)n + CE_TPB - 1) / CE_TPB, 1024) + 256;
}

// loss (device scalar) = mean over labelled rows of -log softmax(logits)[label]; lse (n) and count (device scalar)
// are kept for the backward.  logits (n,c) fp32 row-major, label (n) int64.  bad_labels (device scalar) = number of
// labels that are neither ignore_index nor in [0, c); when it is not 0 the loss is nan.
extern "C" int cross_entropy_forward_hip_launcher(int n, int c, const float *logits, const long long *label, int ignore_index,
                                                  float *lse, float *loss, float *count, float *bad_labels, void *workspace,
                                                  size_t workspace_bytes, void *stream) {
    if (n < 1 || c < 1 || c > 1024 || !logits || !label || !lse || !loss || !count || !bad_labels) return PTV2_ERR_ARG;
    if (!workspace || workspace_bytes < cross_entropy_workspace_bytes(n)) return PTV2_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    unsigned *cnt = ptv2_stream_counters(st);
    if (!cnt) return PTV2_ERR_LAUNCH;
    const int nblk = (int)std::min<long long>(((long long)n + CE_TPB - 1) / CE_TPB, 1024);
    hipLaunchKernelGGL(ce_forward_kernel, dim3(nblk), dim3(CE_TPB), 0, st, n, c, logits, label, ignore_index, lse,
                       (float *)workspace, cnt + CNT_CE, loss, count, bad_labels);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int cross_entropy_backward_hip_launcher(int n, int c, const float *logits, const long long *label, int ignore_index,
                                                   const float *lse, const float *g_loss, const float *count, float *g_logits,
                                                   void *stream) {
    if (n < 1 || c < 1 || !logits || !label || !lse || !g_loss || !count || !g_logits) return PTV2_ERR_ARG;
    const int nblk = (int)std::min<long long>(((long long)n + CE_TPB - 1) / CE_TPB, 4096);
    hipLaunchKernelGGL(ce_backward_kernel, dim3(nblk), dim3(CE_TPB), 0, (hipStream_t)stream, n, c, logits, label, ignore_index,
                       lse, g_loss, count, g_logits);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}
