// ao_amd/csrc/optim.hip -- AdamW over ONE flat fp32 buffer.
// torch's fused AdamW walks the 840 parameter tensors of PT-v2m2 in 24 multi-tensor launches (0.49 ms per step,
// 1.3 ms of host time building the tensor lists); with the parameters living in one flat buffer
// (ao_amd/ptv2/optim.py) the update is a single streaming pass: 16 bytes read + 12 written per parameter.
// Same arithmetic as torch.optim.AdamW (decoupled weight decay, bias correction, eps added after the corrected
// sqrt):  p *= 1 - lr wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= (lr / c1) * m / (sqrt(v) / sqrt(c2) + eps)
#include "common.h"

namespace {
__global__ __launch_bounds__(256) void adamw_flat_kernel(long long n4, float4 *__restrict__ p, const float4 *__restrict__ g,
                                                         float4 *__restrict__ m, float4 *__restrict__ v, float decay,
                                                         float b1, float b2, float step_size, float inv_sqrt_c2, float eps,
                                                         float grad_scale) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 pp = p[i], gg = g[i], mm = m[i], vv = v[i];
#define UPD(c)                                                    \
    {                                                             \
        const float gr = gg.c * grad_scale;                       \
        pp.c *= decay;                                            \
        mm.c = b1 * mm.c + (1.f - b1) * gr;                       \
        vv.c = b2 * vv.c + (1.f - b2) * gr * gr;                  \
        pp.c -= step_size * (mm.c / (sqrtf(vv.c) * inv_sqrt_c2 + eps)); \
    }
        UPD(x) UPD(y) UPD(z) UPD(w)
#undef UPD
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
}
}  // namespace

// n must be a multiple of 4 (pad the flat buffers); step = 1, 2, ... (after increment); grad_scale multiplies the
// gradient first (1 / world size when g holds a sum over ranks)
extern "C" int adamw_flat_hip_launcher(long long n, float *p, const float *g, float *m, float *v, float lr, float beta1,
                                       float beta2, float eps, float weight_decay, int step, float grad_scale, void *stream) {
    if (n < 0 || n % 4 != 0 || !p || !g || !m || !v || step < 1) return PTV2_ERR_ARG;
    if (n == 0) return PTV2_OK;
    const double c1 = 1.0 - pow((double)beta1, step), c2 = 1.0 - pow((double)beta2, step);
    const long long n4 = n / 4;
    const int nblk = (int)((n4 + 255) / 256 < 256 * 16 ? (n4 + 255) / 256 : 256 * 16);
    hipLaunchKernelGGL(adamw_flat_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, n4, (float4 *)p, (const float4 *)g,
                       (float4 *)m, (float4 *)v, 1.f - lr * weight_decay, beta1, beta2, (float)((double)lr / c1),
                       (float)(1.0 / sqrt(c2)), eps, grad_scale);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}
