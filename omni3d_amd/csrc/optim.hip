// optim.hip -- fused SGD-with-momentum and Adam / AdamW (+ amsgrad) steps over a flat fp32 parameter bucket and a fused
// non-finite scan of the flat gradient bucket.
//
// Reference: torch.optim.SGD built by /root/reference/cubercnn/solver/build.py:6-69 (momentum 0.9,
// weight decay per parameter group, no nesterov by default) stepped at tools/train_net.py:250, and the
// per-parameter isnan/isinf gradient scan of tools/train_net.py:222-233 (~230 params x 2 reductions
// with a host sync each) which collapses into ONE pass over the flat bucket here.
// HBM-bound: 4 streams (p, g, m read; p, m written) of 16 B per lane.
// SOLVER.TYPE adam / adam+amsgrad / adamw / adamw+amsgrad (build.py:58-65: torch.optim.Adam / AdamW, eps 1e-2, betas (0.9, 0.999)):
// one pass over p, g, exp_avg, exp_avg_sq (+ max_exp_avg_sq); the step count lives on the device because the divergence guard's
// skip decision does (a skipped iteration must not advance the bias correction, as the reference never calls step() then).
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

// torch.optim.Adam / AdamW, single-tensor form (torch/optim/adam.py _single_tensor_adam, adamw.py): L2 decay into the gradient
// (Adam) or p *= 1 - lr wd (AdamW); exp_avg.lerp_(g, 1 - b1); exp_avg_sq = b2 v + (1 - b2) g g; bias corrections from the step
// count in double, like the Python scalars of the reference; denom = sqrt(v or max v) / sqrt(bc2) + eps; p -= lr / bc1 * m / denom.
struct AdamCoef {
    float beta2, eps, wd, w1, w2, shrink, step_size, bc2s, gscale;
    int decoupled;
};
__device__ __forceinline__ void adam_elem(float& pv, float gv, float& mv, float& vv, float* vmax_elem, const AdamCoef& c) {
    gv *= c.gscale;
    if (c.wd != 0.f) {
        if (c.decoupled) pv *= c.shrink;
        else gv += c.wd * pv;
    }
    mv += (gv - mv) * c.w1;
    vv = vv * c.beta2 + (c.w2 * gv) * gv;
    float second = vv;
    if (vmax_elem != nullptr) {
        second = fmaxf(*vmax_elem, vv);
        *vmax_elem = second;
    }
    pv -= c.step_size * (mv / (sqrtf(second) / c.bc2s + c.eps));
}
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, float* __restrict__ vmax, long n, float lr, float beta1,
                                                   float beta2, float eps, float wd, int decoupled, const float* __restrict__ step,
                                                   float gscale, const float* __restrict__ skip_flag) {
    if (skip_flag != nullptr && skip_flag[0] != 0.f) return;
    const double t = (double)step[0];
    const double bc1 = 1.0 - pow((double)beta1, t), bc2 = 1.0 - pow((double)beta2, t);
    AdamCoef c;
    c.beta2 = beta2; c.eps = eps; c.wd = wd; c.w1 = 1.f - beta1; c.w2 = 1.f - beta2; c.shrink = 1.f - lr * wd;
    c.step_size = (float)((double)lr / bc1); c.bc2s = (float)sqrt(bc2); c.gscale = gscale; c.decoupled = decoupled;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 pv = reinterpret_cast<float4*>(p)[i], mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 xv = vmax != nullptr ? reinterpret_cast<float4*>(vmax)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        adam_elem(pv.x, gv.x, mv.x, vv.x, vmax != nullptr ? &xv.x : nullptr, c);
        adam_elem(pv.y, gv.y, mv.y, vv.y, vmax != nullptr ? &xv.y : nullptr, c);
        adam_elem(pv.z, gv.z, mv.z, vv.z, vmax != nullptr ? &xv.z : nullptr, c);
        adam_elem(pv.w, gv.w, mv.w, vv.w, vmax != nullptr ? &xv.w : nullptr, c);
        reinterpret_cast<float4*>(p)[i] = pv;
        reinterpret_cast<float4*>(m)[i] = mv;
        reinterpret_cast<float4*>(v)[i] = vv;
        if (vmax != nullptr) reinterpret_cast<float4*>(vmax)[i] = xv;
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float pv = p[i], mv = m[i], vv = v[i];
        adam_elem(pv, g[i], mv, vv, vmax != nullptr ? vmax + i : nullptr, c);
        p[i] = pv; m[i] = mv; v[i] = vv;
    }
}
// step[0] += 1 unless the guard asked to skip this iteration
__global__ void adam_tick_kernel(float* __restrict__ step, const float* __restrict__ skip_flag) {
    if (threadIdx.x == 0 && !(skip_flag != nullptr && skip_flag[0] != 0.f)) step[0] += 1.f;
}

__global__ void __launch_bounds__(256) nonfinite_kernel(const float* __restrict__ g, long n, float* __restrict__ flag) {
    int bad = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = g[i];
        bad |= !(fabsf(v) <= 3.402823466e+38f);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) flag[0] = 1.f;
}

// The training loop's safety logic (tools/train_net.py:157-285) as two one-wave kernels around the ONE small all-reduce:
//   guard_pre:  vec[n] = sum of the n losses (vec[n+1] already holds the non-finite-gradient flag of this rank)
//   guard_post: averages over `world`, rolling-loss divergence test (:194-215), success / explode counters, retry decision
//               (:258-259), writes the skip flag the fused SGD kernel reads and the record the host may read back.
// state (3): [recent loss (NaN = unset), iterations_success, iterations_explode]
__global__ void guard_pre_kernel(float* __restrict__ vec, int n) {
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += vec[i];
        vec[n] = s;
    }
}

__global__ void guard_post_kernel(float* __restrict__ vec, int n, float inv_world, float stabilize, float half_period, float tolerance,
                                  float gamma, float* __restrict__ state, float* __restrict__ skip, float* __restrict__ out) {
    if (threadIdx.x != 0) return;
    for (int i = 0; i <= n; ++i) vec[i] *= inv_world;                    // allreduce_dict(average=True), :496
    const float total = vec[n];
    bool bad_grad = vec[n + 1] > 0.f;
    float recent = state[0];
    if (recent != recent) recent = total * 2.0f;                         // :194-196
    bool loss_div = !(fabsf(total) <= 3.402823466e+38f) || total > recent * tolerance;
    if (!(stabilize > 0.f)) { loss_div = false; bad_grad = false; }
    if (!loss_div) recent = recent * (1.f - gamma) + total * gamma;      // :205-215 (before the gradient scan of :222)
    const bool diverging = loss_div || bad_grad;
    state[0] = recent;
    state[1] += diverging ? 0.f : 1.f;
    state[2] += diverging ? 1.f : 0.f;
    const float tot = state[1] + state[2];
    const bool retry = stabilize > 0.f && (state[2] / tot) >= stabilize && tot > half_period;
    skip[0] = diverging ? 1.f : 0.f;
    out[0] = skip[0];
    out[1] = retry ? 1.f : 0.f;
    out[2] = total;
    for (int i = 0; i < n; ++i) out[3 + i] = vec[i];
    vec[n + 1] = 0.f;                                                    // re-arm the gradient scan's flag
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

// In-place Adam (decoupled = 0) / AdamW (decoupled = 1) step over n contiguous fp32 elements of one (lr, weight decay) group.
// max_exp_avg_sq: null, or the amsgrad running maximum.  step: device float holding the number of this update (1 for the first;
// advance it with omni_adam_tick once per optimizer step, before the groups).  grad_scale / skip_flag as in omni_sgd_step.
int omni_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq, long long n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int decoupled, const float* step, float grad_scale,
                   const float* skip_flag, void* stream) {
    if (n < 0 || step == nullptr) return OMNI_ERR_ARG;
    if (n == 0) return OMNI_OK;
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad,
                       exp_avg, exp_avg_sq, max_exp_avg_sq, (long)n, lr, beta1, beta2, eps, weight_decay, decoupled, step, grad_scale,
                       skip_flag);
    return omni_launch_status();
}
int omni_adam_tick(float* step, const float* skip_flag, void* stream) {
    if (step == nullptr) return OMNI_ERR_ARG;
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step, skip_flag);
    return omni_launch_status();
}

// flag[0] = 1 if any element of grad is NaN / +-Inf (flag is NOT cleared here).
int omni_nonfinite_any(const float* grad, long long n, float* flag, void* stream) {
    if (n < 0) return OMNI_ERR_ARG;
    if (n == 0) return OMNI_OK;
    hipLaunchKernelGGL(nonfinite_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, grad, (long)n, flag);
    return omni_launch_status();
}


// vec (n + 2 floats): [n losses | their sum | non-finite-gradient flag].  omni_guard_pre fills the sum; the caller all-reduces
// vec (sum) when world > 1; omni_guard_post decides (see guard_post_kernel).  state (3), skip (1), out (n + 3).
int omni_guard_pre(float* vec, int n, void* stream) {
    if (n <= 0) return OMNI_ERR_ARG;
    hipLaunchKernelGGL(guard_pre_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, vec, n);
    return omni_launch_status();
}

int omni_guard_post(float* vec, int n, int world, float stabilize, float half_period, float tolerance, float gamma, float* state,
                    float* skip, float* out, void* stream) {
    if (n <= 0 || world <= 0) return OMNI_ERR_ARG;
    hipLaunchKernelGGL(guard_post_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, vec, n, 1.f / (float)world, stabilize, half_period,
                       tolerance, gamma, state, skip, out);
    return omni_launch_status();
}

}  // extern "C"
