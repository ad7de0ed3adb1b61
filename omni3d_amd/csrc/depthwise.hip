// depthwise.hip -- depthwise (groups = channels) k x k convolution, forward / data gradient / weight gradient, NHWC.
//
// Reference: the `nn.Conv2d(c, c, k, padding=k//2, stride=s, groups=c, bias=False)` layers of torchvision's mnasnet1_0 that
// /root/reference/cubercnn/modeling/backbone/mnasnet.py:14-17 lifts (`base[3]` and the middle convolution of every inverted
// residual; k in {3, 5}, s in {1, 2}).  One multiply-add per (pixel, channel, tap): 2 k^2 flop for 8+ bytes of traffic, i.e.
// HBM-bound by two orders of magnitude -- no MFMA here.  Lanes run along the channel axis (float4 = 4 channels per lane, 16 B
// coalesced loads), the k x k window is walked in registers, weights are read as (k, k, C) so a tap is one float4 load.
//
// x (N, H, W, C), w (R, R, C) tap-major, y (N, OH, OW, C) with OH = (H + 2 pad - R) / stride + 1.
#include <device_rt.h>

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void fma4(float4& a, const float4& x, const float4& w) {
    a.x += x.x * w.x; a.y += x.y * w.y; a.z += x.z * w.z; a.w += x.w * w.w;
}
inline int dw_grid(long total) {
    long b = (total + 255) / 256;
    if (b > 8192) b = 8192;
    return (int)(b < 1 ? 1 : b);
}

__global__ void __launch_bounds__(256) dwconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         float* __restrict__ y, int N, int H, int W, int C, int R, int stride,
                                                         int pad, int OH, int OW) {
    const int C4 = C >> 2;
    const long total = (long)N * OH * OW * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int col = (int)(i % C4);
        long q = i / C4;
        const int ow = (int)(q % OW); q /= OW;
        const int oh = (int)(q % OH);
        const int n = (int)(q / OH);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = 0; r < R; ++r) {
            const int ih = oh * stride - pad + r;
            if (ih < 0 || ih >= H) continue;
            for (int s = 0; s < R; ++s) {
                const int iw = ow * stride - pad + s;
                if (iw < 0 || iw >= W) continue;
                fma4(acc, ld4(x + (((long)n * H + ih) * W + iw) * C + 4 * col), ld4(w + ((long)r * R + s) * C + 4 * col));
            }
        }
        st4(y + 4 * i, acc);
    }
}

// dx[n,h,w,c] = sum over taps (r,s) and the output pixels that read (h,w) through them: oh = (h + pad - r) / stride when divisible
__global__ void __launch_bounds__(256) dwconv_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                           float* __restrict__ dx, int N, int H, int W, int C, int R, int stride,
                                                           int pad, int OH, int OW) {
    const int C4 = C >> 2;
    const long total = (long)N * H * W * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int col = (int)(i % C4);
        long q = i / C4;
        const int iw = (int)(q % W); q /= W;
        const int ih = (int)(q % H);
        const int n = (int)(q / H);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = 0; r < R; ++r) {
            const int th = ih + pad - r;
            if (th < 0 || th % stride != 0) continue;
            const int oh = th / stride;
            if (oh >= OH) continue;
            for (int s = 0; s < R; ++s) {
                const int tw = iw + pad - s;
                if (tw < 0 || tw % stride != 0) continue;
                const int ow = tw / stride;
                if (ow >= OW) continue;
                fma4(acc, ld4(dy + (((long)n * OH + oh) * OW + ow) * C + 4 * col), ld4(w + ((long)r * R + s) * C + 4 * col));
            }
        }
        st4(dx + 4 * i, acc);
    }
}

// dw[r,s,c] = sum over output pixels of x[n, oh s - pad + r, ow s - pad + s', c] dy[n, oh, ow, c].  A workgroup owns 64 channel
// quads x 4 pixel slices; each lane keeps the R x R partial sums of its 4 channels in registers over its slice of the output
// pixels, the 4 slices are folded through LDS and one atomic per (workgroup, tap, channel) lands in dw (zeroed by the launcher).
template <int R>
__global__ void __launch_bounds__(256) dwconv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ dw, int N, int H, int W, int C, int stride, int pad,
                                                           int OH, int OW, int slices) {
    __shared__ float4 fold[4][64];
    const int C4 = C >> 2;
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int slice = blockIdx.y * 4 + sub, nslice = slices * 4;
    const bool active = col < C4;
    float4 acc[R * R];
    for (int t = 0; t < R * R; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    const long P = (long)N * OH * OW;
    if (active) {
        for (long p = slice; p < P; p += nslice) {
            const int ow = (int)(p % OW);
            long q = p / OW;
            const int oh = (int)(q % OH);
            const int n = (int)(q / OH);
            const float4 g = ld4(dy + p * C + 4 * col);
            for (int r = 0; r < R; ++r) {
                const int ih = oh * stride - pad + r;
                if (ih < 0 || ih >= H) continue;
                for (int s = 0; s < R; ++s) {
                    const int iw = ow * stride - pad + s;
                    if (iw < 0 || iw >= W) continue;
                    fma4(acc[r * R + s], ld4(x + (((long)n * H + ih) * W + iw) * C + 4 * col), g);
                }
            }
        }
    }
    for (int t = 0; t < R * R; ++t) {
        fold[sub][lane] = acc[t];
        __syncthreads();
        if (sub == 0 && active) {
            float4 v = fold[0][lane];
            for (int k = 1; k < 4; ++k) { const float4 o = fold[k][lane]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            float* d = dw + (long)t * C + 4 * col;
            atomicAdd(d + 0, v.x); atomicAdd(d + 1, v.y); atomicAdd(d + 2, v.z); atomicAdd(d + 3, v.w);
        }
        __syncthreads();
    }
}

inline bool bad(int N, int H, int W, int C, int R, int stride, int pad) {
    return N < 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || (R != 3 && R != 5) || (stride != 1 && stride != 2) || pad < 0 || pad >= R ||
           H + 2 * pad < R || W + 2 * pad < R;
}

}  // namespace

extern "C" {

int omni_dwconv_fwd(const float* x, const float* w, float* y, int N, int H, int W, int C, int R, int stride, int pad, void* stream) {
    if (bad(N, H, W, C, R, stride, pad)) return OMNI_ERR_ARG;
    const int OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - R) / stride + 1;
    const long total = (long)N * OH * OW * (C / 4);
    if (total == 0) return OMNI_OK;
    hipLaunchKernelGGL(dwconv_fwd_kernel, dim3(dw_grid(total)), dim3(256), 0, (hipStream_t)stream, x, w, y, N, H, W, C, R, stride, pad,
                       OH, OW);
    return omni_launch_status();
}

int omni_dwconv_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int R, int stride, int pad, void* stream) {
    if (bad(N, H, W, C, R, stride, pad)) return OMNI_ERR_ARG;
    const int OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - R) / stride + 1;
    const long total = (long)N * H * W * (C / 4);
    if (total == 0) return OMNI_OK;
    hipLaunchKernelGGL(dwconv_dgrad_kernel, dim3(dw_grid(total)), dim3(256), 0, (hipStream_t)stream, dy, w, dx, N, H, W, C, R, stride,
                       pad, OH, OW);
    return omni_launch_status();
}

// dw (R, R, C) is overwritten.
int omni_dwconv_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int R, int stride, int pad, void* stream) {
    if (bad(N, H, W, C, R, stride, pad)) return OMNI_ERR_ARG;
    const int OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - R) / stride + 1;
    hipStream_t st = (hipStream_t)stream;
    omni_memset_async(dw, 0, sizeof(float) * (size_t)R * R * C, st);
    const long P = (long)N * OH * OW;
    if (P == 0) return OMNI_OK;
    const int cb = (C / 4 + 63) / 64;
    long want = 2048 / cb;                       // ~2048 workgroups over the chip, at least 64 pixels per lane
    if (want > P / 256) want = P / 256;
    const int slices = (int)(want < 1 ? 1 : want);
    if (R == 3)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(dwconv_wgrad_kernel<3>), dim3(cb, slices), dim3(256), 0, st, x, dy, dw, N, H, W, C, stride, pad,
                           OH, OW, slices);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(dwconv_wgrad_kernel<5>), dim3(cb, slices), dim3(256), 0, st, x, dy, dw, N, H, W, C, stride, pad,
                           OH, OW, slices);
    return omni_launch_status();
}

}  // extern "C"
