// pool3.hip -- nn.MaxPool2d(kernel_size=3, stride=2, padding=1) forward / backward on NHWC fp32.
//
// Reference call site: the torchvision ResNet stem used by the ResNet-34 backbone
//   /root/reference/cubercnn/modeling/backbone/resnet.py:34,52  (`self.maxpool = base.maxpool`).
// HBM-bound; float4 (4 channels) per lane, lanes walk C.  Padding behaves as -inf; the first maximum in
// window scan order wins (ATen).  Backward is a gather (each input pixel checks the <= 4 windows that contain
// it and re-derives their argmax), so it is deterministic and needs no atomics.
#include <device_rt.h>

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// argmax position (window-local index 0..8, -1 if empty) of channel lane `k` of window (oh, ow)
__device__ __forceinline__ void window_argmax(const float* __restrict__ x, int n, int oh, int ow, int H, int W, int C,
                                              int c4, float4& best, int (&arg)[4]) {
    best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    arg[0] = arg[1] = arg[2] = arg[3] = -1;
    for (int r = 0; r < 3; ++r) {
        const int ih = oh * 2 - 1 + r;
        if (ih < 0 || ih >= H) continue;
        for (int s = 0; s < 3; ++s) {
            const int iw = ow * 2 - 1 + s;
            if (iw < 0 || iw >= W) continue;
            const float4 v = ld4(x + (((long)n * H + ih) * W + iw) * C + 4 * c4);
            const int id = r * 3 + s;
            if (v.x > best.x || arg[0] < 0) { best.x = v.x; arg[0] = id; }
            if (v.y > best.y || arg[1] < 0) { best.y = v.y; arg[1] = id; }
            if (v.z > best.z || arg[2] < 0) { best.z = v.z; arg[2] = id; }
            if (v.w > best.w || arg[3] < 0) { best.w = v.w; arg[3] = id; }
        }
    }
}

__global__ void __launch_bounds__(256) maxpool3_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H,
                                                           int W, int C, int OH, int OW) {
    const int C4 = C >> 2;
    const long total = (long)N * OH * OW * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long q = i / C4;
        const int ow = (int)(q % OW); q /= OW;
        const int oh = (int)(q % OH);
        const int n = (int)(q / OH);
        float4 best;
        int arg[4];
        window_argmax(x, n, oh, ow, H, W, C, c4, best, arg);
        *reinterpret_cast<float4*>(y + 4 * i) = best;
    }
}

__global__ void __launch_bounds__(256) maxpool3_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ dx, int N, int H, int W, int C, int OH, int OW) {
    const int C4 = C >> 2;
    const long total = (long)N * H * W * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long q = i / C4;
        const int iw = (int)(q % W); q /= W;
        const int ih = (int)(q % H);
        const int n = (int)(q / H);
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        // windows containing (ih, iw): oh in {floor((ih+1)/2) - (0|1)} with 2*oh - 1 <= ih <= 2*oh + 1
        for (int oh = (ih + 1) / 2 - 1; oh <= (ih + 1) / 2; ++oh) {
            if (oh < 0 || oh >= OH || ih < 2 * oh - 1 || ih > 2 * oh + 1) continue;
            for (int ow = (iw + 1) / 2 - 1; ow <= (iw + 1) / 2; ++ow) {
                if (ow < 0 || ow >= OW || iw < 2 * ow - 1 || iw > 2 * ow + 1) continue;
                float4 best;
                int arg[4];
                window_argmax(x, n, oh, ow, H, W, C, c4, best, arg);
                const int me = (ih - (2 * oh - 1)) * 3 + (iw - (2 * ow - 1));
                const float4 d = ld4(dy + (((long)n * OH + oh) * OW + ow) * C + 4 * c4);
                if (arg[0] == me) g.x += d.x;
                if (arg[1] == me) g.y += d.y;
                if (arg[2] == me) g.z += d.z;
                if (arg[3] == me) g.w += d.w;
            }
        }
        *reinterpret_cast<float4*>(dx + 4 * i) = g;
    }
}

inline int ew_grid(long total) {
    long g = (total + 255) / 256;
    if (g > 2048) g = 2048;
    return (int)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" {

// y (N,OH,OW,C) = maxpool3x3/s2/p1(x (N,H,W,C)), OH = (H + 2 - 3)/2 + 1.
int omni_maxpool3s2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream) {
    if ((C & 3) || H < 1 || W < 1) return OMNI_ERR_ARG;
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const long total = (long)N * OH * OW * (C / 4);
    if (total == 0) return OMNI_OK;
    hipLaunchKernelGGL(maxpool3_fwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, C, OH, OW);
    return omni_launch_status();
}

int omni_maxpool3s2_bwd(const float* x, const float* dy, float* dx, int N, int H, int W, int C, void* stream) {
    if ((C & 3) || H < 1 || W < 1) return OMNI_ERR_ARG;
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const long total = (long)N * H * W * (C / 4);
    if (total == 0) return OMNI_OK;
    hipLaunchKernelGGL(maxpool3_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, N, H, W, C, OH, OW);
    return omni_launch_status();
}

}  // extern "C"
