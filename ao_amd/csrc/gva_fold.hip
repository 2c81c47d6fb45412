// ao_amd/csrc/gva_fold.hip -- the parameter-sized algebra around the fused GVA stages, as single launches.
//
// Folding the two BatchNorms of GroupedVectorAttention into affine maps (ao_amd/ptv2/gva.py) is a few
// dozen elementwise ops on (C,3) / (C,) / (G,) tensors; as eager torch ops (with their autograd) they were
// ~130 launches per attention block and made the step host-bound (profiles/r01_fused_v4_*).  Each fold and
// its hand-derived backward is one tiny kernel here:
//
//  fold_p  (linear_p_bias[0..1]: Linear(3,C) -> BatchNorm over all N*K neighbour slots)
//     mean_c = Wp1[c].mu + bp1[c],  var_c = Wp1[c]^T Cov Wp1[c]      (closed form from pos moments; training)
//     s_c = gamma_c / sqrt(var_c + eps);  a[c,:] = Wp1[c,:] s_c;  b[c] = (bp1[c] - mean_c) s_c + beta_c
//     eval: mean / var are the running statistics.  Training also updates the running statistics.
//  fold_w  (weight_encoding[1]: BatchNorm over the (N*K, G) logits, from their column sums T1, T2)
//     mean = T1/R, var = T2/R - mean^2, sc = gamma / sqrt(var + eps), sh = beta - mean sc
#include "gva_common.h"
#include "gva_fold_p.h"

namespace gva {

__global__ void fold_p_fwd_kernel(FoldPFwdArgs A) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch < A.c) fold_p_fwd_channel(A, ch);
}

__global__ void fold_p_bwd_kernel(FoldPBwdArgs A) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch < A.c) fold_p_bwd_channel(A, ch);
}

__global__ void fold_w_fwd_kernel(int g, const double *__restrict__ T1, const double *__restrict__ T2, FoldWFwdArgs A) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < g) fold_w_fwd_channel(A, j, T1[j], T2[j]);
}

__global__ void fold_w_bwd_kernel(int g, const float *__restrict__ gamma, const double *__restrict__ mean_in,
                                  const double *__restrict__ rstd_in, int training, double rows,
                                  const float *__restrict__ gsc, const float *__restrict__ gsh,
                                  double *__restrict__ gT1, double *__restrict__ gT2, float *__restrict__ ggamma,
                                  float *__restrict__ gbeta) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= g) return;
    const FoldWBwdArgs A{gamma, mean_in, rstd_in, training, rows, gsc, gsh, ggamma, gbeta};
    fold_w_bwd_channel(A, j, gT1[j], gT2[j], ggamma[j], gbeta[j]);
}

}  // namespace gva

using namespace gva;

extern "C" int gva_fold_p_forward_hip_launcher(int c, const float *Wp1, const float *bp1, const float *gamma,
                                               const float *beta, const double *mu, const double *cov,
                                               float *running_mean, float *running_var,
                                               long long *num_batches_tracked, int training, double rows, float eps,
                                               float momentum, float *a, float *b, float *rstd, void *stream) {
    if (c < 1) return PTV2_ERR_ARG;
    hipLaunchKernelGGL(fold_p_fwd_kernel, dim3(divup(c, 128)), dim3(128), 0, (hipStream_t)stream,
                       FoldPFwdArgs{c, Wp1, bp1, gamma, beta, mu, cov, running_mean, running_var, num_batches_tracked, training,
                                    rows, eps, momentum, a, b, rstd});
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int gva_fold_p_backward_hip_launcher(int c, const float *Wp1, const float *bp1, const float *gamma,
                                                const double *mu, const double *cov, const float *running_mean,
                                                const float *rstd, int training, const float *ga, const float *gb,
                                                float *gWp1, float *gbp1, float *ggamma, float *gbeta, void *stream) {
    if (c < 1) return PTV2_ERR_ARG;
    hipLaunchKernelGGL(fold_p_bwd_kernel, dim3(divup(c, 128)), dim3(128), 0, (hipStream_t)stream,
                       FoldPBwdArgs{c, Wp1, bp1, gamma, mu, cov, running_mean, rstd, training, ga, gb, nullptr, nullptr, gWp1,
                                    gbp1, ggamma, gbeta});
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int gva_fold_w_forward_hip_launcher(int g, const double *T1, const double *T2, const float *gamma,
                                               const float *beta, float *running_mean, float *running_var,
                                               long long *num_batches_tracked, int training, double rows, float eps,
                                               float momentum, float *sc, float *sh, double *mean, double *rstd,
                                               void *stream) {
    if (g < 1) return PTV2_ERR_ARG;
    hipLaunchKernelGGL(fold_w_fwd_kernel, dim3(divup(g, 64)), dim3(64), 0, (hipStream_t)stream, g, T1, T2,
                       FoldWFwdArgs{gamma, beta, running_mean, running_var, num_batches_tracked, training, rows, eps, momentum, sc,
                                    sh, mean, rstd});
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}

extern "C" int gva_fold_w_backward_hip_launcher(int g, const float *gamma, const double *mean, const double *rstd,
                                                int training, double rows, const float *gsc, const float *gsh,
                                                double *gT1, double *gT2, float *ggamma, float *gbeta, void *stream) {
    if (g < 1) return PTV2_ERR_ARG;
    hipLaunchKernelGGL(fold_w_bwd_kernel, dim3(divup(g, 64)), dim3(64), 0, (hipStream_t)stream, g, gamma, mean, rstd,
                       training, rows, gsc, gsh, gT1, gT2, ggamma, gbeta);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}
