// ao_amd/csrc/gva_fold_p.h -- per-channel bodies of the folded BN_p algebra (see gva_fold.hip), shared by the
// stand-alone fold kernels and by the block runtime, which runs them inside its fold_m launches (gva_block.hip).
#pragma once
#include "gva_common.h"

namespace gva {

struct FoldPFwdArgs {
    int c;
    const float *Wp1, *bp1, *gamma, *beta;
    const double *mu, *cov;
    float *run_mean, *run_var;
    long long *batches;
    int training;
    double rows;
    float eps, momentum;
    float *a, *b, *rstd_out;
};

__device__ inline void fold_p_fwd_channel(const FoldPFwdArgs &A, int ch) {
    const float *Wp1 = A.Wp1, *bp1 = A.bp1;
    const double *mu = A.mu, *cov = A.cov;
    const double w0 = Wp1[3 * ch], w1 = Wp1[3 * ch + 1], w2 = Wp1[3 * ch + 2];
    double mean, rstd;
    if (A.training) {
        mean = w0 * mu[0] + w1 * mu[1] + w2 * mu[2] + (double)bp1[ch];
        const double t0 = cov[0] * w0 + cov[1] * w1 + cov[2] * w2;
        const double t1 = cov[3] * w0 + cov[4] * w1 + cov[5] * w2;
        const double t2 = cov[6] * w0 + cov[7] * w1 + cov[8] * w2;
        double var = w0 * t0 + w1 * t1 + w2 * t2;
        var = var > 0.0 ? var : 0.0;
        rstd = 1.0 / sqrt(var + (double)A.eps);
        if (A.run_mean) {
            const double unb = A.rows > 1.0 ? var * (A.rows / (A.rows - 1.0)) : var;
            A.run_mean[ch] = (float)((1.0 - A.momentum) * (double)A.run_mean[ch] + A.momentum * mean);
            A.run_var[ch] = (float)((1.0 - A.momentum) * (double)A.run_var[ch] + A.momentum * unb);
            if (ch == 0 && A.batches) *A.batches += 1;
        }
    } else {
        mean = (double)A.run_mean[ch];
        rstd = 1.0 / sqrt((double)A.run_var[ch] + (double)A.eps);
    }
    const double s = (double)A.gamma[ch] * rstd;
    A.a[3 * ch] = (float)(w0 * s);
    A.a[3 * ch + 1] = (float)(w1 * s);
    A.a[3 * ch + 2] = (float)(w2 * s);
    A.b[ch] = (float)(((double)bp1[ch] - mean) * s + (double)A.beta[ch]);
    A.rstd_out[ch] = (float)rstd;
}

struct FoldPBwdArgs {
    int c;
    const float *Wp1, *bp1, *gamma;
    const double *mu, *cov;
    const float *run_mean, *rstd_in;
    int training;
    const float *ga, *gb, *ga2, *gb2;  // (ga2, gb2) may be NULL: second contribution to the gradient of (a, b)
    float *gWp1, *gbp1, *ggamma, *gbeta;
};

__device__ inline void fold_p_bwd_channel(const FoldPBwdArgs &A, int ch) {
    const double *mu = A.mu, *cov = A.cov;
    const double w[3] = {A.Wp1[3 * ch], A.Wp1[3 * ch + 1], A.Wp1[3 * ch + 2]};
    double g[3] = {A.ga[3 * ch], A.ga[3 * ch + 1], A.ga[3 * ch + 2]};
    double gbv = A.gb[ch];
    if (A.ga2) {  // (a, b) feed two stages (logits and aggregation): their gradients are summed here
        g[0] += A.ga2[3 * ch]; g[1] += A.ga2[3 * ch + 1]; g[2] += A.ga2[3 * ch + 2];
        gbv += A.gb2[ch];
    }
    const double rstd = A.rstd_in[ch], gam = A.gamma[ch];
    const double s = gam * rstd;
    A.gbeta[ch] = (float)gbv;
    if (A.training) {
        const double wmu = w[0] * mu[0] + w[1] * mu[1] + w[2] * mu[2];  // b = -wmu * s + beta
        const double gs = g[0] * w[0] + g[1] * w[1] + g[2] * w[2] - gbv * wmu;
        A.ggamma[ch] = (float)(gs * rstd);
        const double gvar = gs * gam * (-0.5) * rstd * rstd * rstd;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const double cw = cov[3 * d] * w[0] + cov[3 * d + 1] * w[1] + cov[3 * d + 2] * w[2];
            A.gWp1[3 * ch + d] = (float)(g[d] * s - gbv * mu[d] * s + gvar * 2.0 * cw);
        }
        A.gbp1[ch] = 0.f;  // the batch mean removes the bias
    } else {
        const double dm = (double)A.bp1[ch] - (double)A.run_mean[ch];
        A.ggamma[ch] = (float)((g[0] * w[0] + g[1] * w[1] + g[2] * w[2] + gbv * dm) * rstd);
#pragma unroll
        for (int d = 0; d < 3; ++d) A.gWp1[3 * ch + d] = (float)(g[d] * s);
        A.gbp1[ch] = (float)(gbv * s);
    }
}

}  // namespace gva
