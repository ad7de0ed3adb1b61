// resize.hip -- the input pipeline's image resampling on the device: detectron2's ResizeShortestEdge / ResizeTransform
// (T.ResizeShortestEdge built from cfg.INPUT.MIN_SIZE_TRAIN 256..640 / MAX_SIZE_TRAIN, reference
// /root/reference/cubercnn/data/dataset_mapper.py:25-27 `self.augmentations(aug_input)`, configs/Base.yaml:10-13) calls
// PIL `Image.resize((w, h), BILINEAR)` on the uint8 image; RandomFlip is a horizontal mirror.
//
// Pillow's 8-bit resampling (ImagingResample, Resample.c) is a separable convolution with a triangle filter whose
// support grows with the down-scale factor (antialiasing), evaluated in 32-bit fixed point (22 fractional bits, rounded
// through a 0.5 offset) with an intermediate uint8 image between the horizontal and the vertical pass.  The per-output
// coefficient rows are built on the host in double precision exactly as Pillow does (omni3d_amd/kernels/resize.py) and
// handed over as int32; the two kernels below are the two passes, bit-exact with Pillow (tests/test_resize.py pins them
// to PIL itself, which is installed).  HBM-bound: one read of the source, one uint8 intermediate, one write.
#include <device_rt.h>

namespace {

__device__ __forceinline__ unsigned char clip8(int v) {
    v >>= 22;                       // PRECISION_BITS = 32 - 8 - 2
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: src (P, H, W) planes -> dst (P, H, WO); optional mirror of the OUTPUT columns (RandomFlip after resize
// == resize after flip for a symmetric filter; detectron2 applies the flip after the resize)
__global__ void __launch_bounds__(256) resize_h_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                        const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                        int planes, int H, int W, int WO, int flip) {
    const long total = (long)planes * H * WO;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int xx = (int)(i % WO);
        const long row = i / WO;
        const int xmin = bounds[2 * xx], cnt = bounds[2 * xx + 1];
        const unsigned char* s = src + row * W + xmin;
        const int* k = kk + (long)xx * ksize;
        int ss = 1 << 21;
        for (int x = 0; x < cnt; ++x) ss += (int)s[x] * k[x];
        dst[row * WO + (flip ? WO - 1 - xx : xx)] = clip8(ss);
    }
}

// vertical pass: src (P, H, WO) -> dst (P, HO, WO)
__global__ void __launch_bounds__(256) resize_v_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                        const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                        int planes, int H, int HO, int WO) {
    const long total = (long)planes * HO * WO;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % WO);
        const long r = i / WO;
        const int yy = (int)(r % HO), p = (int)(r / HO);
        const int ymin = bounds[2 * yy], cnt = bounds[2 * yy + 1];
        const unsigned char* s = src + ((long)p * H + ymin) * WO + x;
        const int* k = kk + (long)yy * ksize;
        int ss = 1 << 21;
        for (int y = 0; y < cnt; ++y) ss += (int)s[(long)y * WO] * k[y];
        dst[i] = clip8(ss);
    }
}

__global__ void __launch_bounds__(256) hflip_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, long rows, int W) {
    const long total = rows * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / W;
        const int x = (int)(i - r * W);
        dst[i] = src[r * W + (W - 1 - x)];
    }
}

inline unsigned grid_for(long n) {
    long g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

extern "C" {

// src (planes, H, W) uint8 -> dst (planes, HO, WO) uint8; tmp: planes*H*WO bytes of scratch.
// bounds_h (WO, 2) / kk_h (WO, ksize_h) and bounds_v (HO, 2) / kk_v (HO, ksize_v): Pillow's per-output-pixel source window
// [first, count) and fixed-point coefficients.  H == HO skips the vertical pass, W == WO the horizontal one (Pillow does the same).
int omni_resize_bilinear_u8(const unsigned char* src, unsigned char* dst, unsigned char* tmp, int planes, int H, int W, int HO,
                            int WO, const int* bounds_h, const int* kk_h, int ksize_h, const int* bounds_v, const int* kk_v,
                            int ksize_v, int flip, void* stream) {
    if (planes <= 0 || H <= 0 || W <= 0 || HO <= 0 || WO <= 0 || ksize_h <= 0 || ksize_v <= 0) return OMNI_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const bool need_h = W != WO, need_v = H != HO;
    const unsigned char* cur = src;
    if (need_h || flip) {
        unsigned char* out = need_v ? tmp : dst;
        if (need_h)
            hipLaunchKernelGGL(resize_h_kernel, dim3(grid_for((long)planes * H * WO)), dim3(256), 0, st, cur, out, bounds_h, kk_h, ksize_h,
                               planes, H, W, WO, flip);
        else
            hipLaunchKernelGGL(hflip_kernel, dim3(grid_for((long)planes * H * W)), dim3(256), 0, st, cur, out, (long)planes * H, W);
        cur = out;
    }
    if (need_v)
        hipLaunchKernelGGL(resize_v_kernel, dim3(grid_for((long)planes * HO * WO)), dim3(256), 0, st, cur, dst, bounds_v, kk_v, ksize_v,
                           planes, H, HO, WO);
    else if (cur == src)
        (void)hipMemcpyAsync(dst, src, (size_t)planes * H * W, hipMemcpyDeviceToDevice, st);
    return omni_launch_status();
}

}  // extern "C"
