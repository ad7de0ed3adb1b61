// glue.hip -- the last scalar-sized launches of the training step as kernels of this library (round 6).
//
// After round 5 about thirty ATen launches were left on the critical stream of the replayed step -- loss scaling and summing, clamps,
// slice copies, counter bumps, the zero-fill of the gradient bucket, the stacking of the guard's loss vector -- each one a dependent
// ~4.8 us slot (profiles/r05_timeline_one_step.txt).  None of them is work: they exist because the reference's Python is written
// that way (fast_rcnn.py:189-194, roi_heads.py:745-768, tools/train_net.py:180-186).  The entry points below do each of those jobs in
// one launch of a few threads; arrays of device pointers / coefficients travel in the kernel arguments (nothing is staged).
#include <device_rt.h>

namespace {

constexpr int GLUE_MAXV = 16;      // vectors / scalars / coefficients per launch
constexpr int GLUE_MAXP = 64;      // counters per launch

struct ScaleArgs {
    double coef[GLUE_MAXV];
};

// out[i] = (float)( src[i * stride] * coef[i] / max(denom, denom_min) ), computed in double and rounded once
template <typename T>
__global__ void scale_vec_kernel(const T* __restrict__ src, int stride, ScaleArgs a, const T* __restrict__ denom, double denom_min, int n,
                                 float* __restrict__ out) {
    const int i = threadIdx.x;
    if (i >= n) return;
    double v = (double)src[(long)i * stride] * a.coef[i];
    if (denom != nullptr) {
        const double d = (double)denom[0];
        v /= d < denom_min ? denom_min : d;
    }
    out[i] = (float)v;
}

struct VecList {
    const float* p[GLUE_MAXV];
    int len[GLUE_MAXV];
};

// out[0] = sum over the elements of vectors [0, nfirst), out[1] = the same over [nfirst, nvec), out[2] = out[0] + out[1];
// one thread, a fixed order: the ten losses of a step
__global__ void sum_vectors_kernel(VecList v, int nvec, int nfirst, float* __restrict__ out) {
    if (threadIdx.x != 0) return;
    float s0 = 0.f, s1 = 0.f;
    for (int k = 0; k < nvec; ++k)
        for (int i = 0; i < v.len[k]; ++i) {
            if (k < nfirst) s0 += v.p[k][i];
            else s1 += v.p[k][i];
        }
    out[0] = s0;
    out[1] = s1;
    out[2] = s0 + s1;
}

// vec[i] = *p[i] (i < n), vec[n] = their sum: the loss vector of the loop's guard (tools/train_net.py:186 allreduce_dict input)
__global__ void guard_gather_kernel(VecList v, int n, float* __restrict__ vec) {
    if (threadIdx.x != 0) return;
    float s = 0.f;
    for (int i = 0; i < n; ++i) {
        const float x = v.p[i][0];
        vec[i] = x;
        s += x;
    }
    vec[n] = s;
}

struct CounterList {
    long long* p[GLUE_MAXP];
};
__global__ void bump_counters_kernel(CounterList c, int n, long long delta) {
    const int i = threadIdx.x;
    if (i < n) c.p[i][0] += delta;
}

}  // namespace

extern "C" {

// src: n values at element stride `stride` (0 = one broadcast scalar), double when src_f64 != 0 else float; coef: n HOST doubles
// (NULL = all ones); denom [nullable]: one device value of src's type, clamped from below by denom_min.  n <= 16.
int omni_scale_vec(const void* src, int src_f64, int stride, const double* coef, const void* denom, double denom_min, int n, float* out,
                   void* stream) {
    if (src == nullptr || out == nullptr || n <= 0 || n > GLUE_MAXV || stride < 0) return OMNI_ERR_ARG;
    ScaleArgs a;
    for (int i = 0; i < GLUE_MAXV; ++i) a.coef[i] = (coef != nullptr && i < n) ? coef[i] : 1.0;
    if (src_f64)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_vec_kernel<double>), dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)src, stride, a,
                           (const double*)denom, denom_min, n, out);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_vec_kernel<float>), dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)src, stride, a,
                           (const float*)denom, denom_min, n, out);
    return omni_launch_status();
}

// vecs: nvec HOST entries, each a device pointer to lens[k] floats; out: 3 floats (see sum_vectors_kernel).  nvec <= 16.
int omni_sum_vectors(const void* const* vecs, const int* lens, int nvec, int nfirst, float* out, void* stream) {
    if (vecs == nullptr || lens == nullptr || out == nullptr || nvec <= 0 || nvec > GLUE_MAXV || nfirst < 0 || nfirst > nvec) return OMNI_ERR_ARG;
    VecList v;
    for (int k = 0; k < GLUE_MAXV; ++k) {
        v.p[k] = k < nvec ? (const float*)vecs[k] : nullptr;
        v.len[k] = k < nvec ? lens[k] : 0;
        if (k < nvec && (vecs[k] == nullptr || lens[k] < 0 || lens[k] > 4096)) return OMNI_ERR_ARG;
    }
    hipLaunchKernelGGL(sum_vectors_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, v, nvec, nfirst, out);
    return omni_launch_status();
}

// scalars: n HOST entries, each a device pointer to one float; vec: n + 1 floats (values, then their sum) -- omni_guard_pre with
// the stacking of the loss dict folded in.  n <= 16.
int omni_guard_gather(const void* const* scalars, int n, float* vec, void* stream) {
    if (scalars == nullptr || vec == nullptr || n <= 0 || n > GLUE_MAXV) return OMNI_ERR_ARG;
    VecList v;
    for (int k = 0; k < GLUE_MAXV; ++k) {
        v.p[k] = k < n ? (const float*)scalars[k] : nullptr;
        v.len[k] = 1;
        if (k < n && scalars[k] == nullptr) return OMNI_ERR_ARG;
    }
    hipLaunchKernelGGL(guard_gather_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, v, n, vec);
    return omni_launch_status();
}

// *counters[i] += delta for n HOST entries of device int64 pointers (`num_batches_tracked += 1` of every BatchNorm: 64 per launch)
int omni_bump_counters(const void* const* counters, int n, long long delta, void* stream) {
    if (n < 0 || (n > 0 && counters == nullptr)) return OMNI_ERR_ARG;
    for (int base = 0; base < n; base += GLUE_MAXP) {
        CounterList c;
        const int m = n - base < GLUE_MAXP ? n - base : GLUE_MAXP;
        for (int i = 0; i < GLUE_MAXP; ++i) {
            c.p[i] = i < m ? (long long*)counters[base + i] : nullptr;
            if (i < m && c.p[i] == nullptr) return OMNI_ERR_ARG;
        }
        hipLaunchKernelGGL(bump_counters_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, c, m, delta);
    }
    return omni_launch_status();
}

// nbytes zero bytes at p (the gradient bucket's zero_grad) as a kernel node -- see omni_memset_async
int omni_zero(void* p, long long nbytes, void* stream) {
    if (nbytes < 0 || (nbytes > 0 && p == nullptr)) return OMNI_ERR_ARG;
    omni_memset_async(p, 0, (size_t)nbytes, (hipStream_t)stream);
    return omni_launch_status();
}

}  // extern "C"
