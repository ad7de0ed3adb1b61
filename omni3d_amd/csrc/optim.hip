// optim.hip -- fused SGD-with-momentum step over a flat fp32 parameter bucket and a fused
// non-finite scan of the flat gradient bucket.
//
// Reference: torch.optim.SGD built by /root/reference/cubercnn/solver/build.py:6-69 (momentum 0.9,
// weight decay per parameter group, no nesterov by default) stepped at tools/train_net.py:250, and the
// per-parameter isnan/isinf gradient scan of tools/train_net.py:222-233 (~230 params x 2 reductions
// with a host sync each) which collapses into ONE pass over the flat bucket here.
// HBM-bound: 4 streams (p, g, m read; p, m written) of 16 B per lane.
#include <device_rt.h>

namespace {

__global__ void __launch_bounds__(256) sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                  long n, float lr, float momentum, float dampening, float wd, int nesterov,
                                                  int first_step, float gscale, const float* __restrict__ skip_flag) {
    if (skip_flag != nullptr && skip_flag[0] != 0.f) return;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 mv = reinterpret_cast<float4*>(m)[i];
        float pe[4] = {pv.x, pv.y, pv.z, pv.w}, ge[4] = {gv.x, gv.y, gv.z, gv.w}, me[4] = {mv.x, mv.y, mv.z, mv.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float d = ge[k] * gscale + wd * pe[k];
            if (momentum != 0.f) {
                me[k] = first_step ? d : momentum * me[k] + (1.f - dampening) * d;
                d = nesterov ? d + momentum * me[k] : me[k];
            }
            pe[k] -= lr * d;
        }
        reinterpret_cast<float4*>(p)[i] = make_float4(pe[0], pe[1], pe[2], pe[3]);
        reinterpret_cast<float4*>(m)[i] = make_float4(me[0], me[1], me[2], me[3]);
    }
    // tail
    for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float d = g[i] * gscale + wd * p[i];
        if (momentum != 0.f) {
            m[i] = first_step ? d : momentum * m[i] + (1.f - dampening) * d;
            d = nesterov ? d + momentum * m[i] : m[i];
        }
        p[i] -= lr * d;
    }
}

__global__ void __launch_bounds__(256) nonfinite_kernel(const float* __restrict__ g, long n, float* __restrict__ flag) {
    int bad = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = g[i];
        bad |= !(fabsf(v) <= 3.402823466e+38f);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) flag[0] = 1.f;
}

inline int grid_for(long n) {
    long b = (n / 4 + 255) / 256;
    if (b > 2048) b = 2048;
    return (int)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" {

// In-place SGD step over n contiguous fp32 elements (one weight-decay group).  skip_flag [nullable]:
// device float; when != 0 the step is skipped (divergence guard decided on the device).  grad_scale multiplies the
// gradient as it is read: the data-parallel exchange sums over ranks and hands 1/world here instead of rescaling the
// 191.6 MB bucket in a separate pass.
int omni_sgd_step(float* param, const float* grad, float* momentum_buf, long long n, float lr, float momentum,
                  float dampening, float weight_decay, int nesterov, int first_step, float grad_scale, const float* skip_flag,
                  void* stream) {
    if (n < 0) return OMNI_ERR_ARG;
    if (n == 0) return OMNI_OK;
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, momentum_buf, (long)n,
                       lr, momentum, dampening, weight_decay, nesterov, first_step, grad_scale, skip_flag);
    return omni_launch_status();
}

// flag[0] = 1 if any element of grad is NaN / +-Inf (flag is NOT cleared here).
int omni_nonfinite_any(const float* grad, long long n, float* flag, void* stream) {
    if (n < 0) return OMNI_ERR_ARG;
    if (n == 0) return OMNI_OK;
    hipLaunchKernelGGL(nonfinite_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, grad, (long)n, flag);
    return omni_launch_status();
}

}  // extern "C"
