// box_loss.hip -- Fast R-CNN box-head losses (softmax cross-entropy + per-class L1 box regression),
// forward statistics and the gradient wrt the fused prediction tensor.
//
// Reference: FastRCNNOutputs.losses / box_reg_loss
//   /root/reference/cubercnn/modeling/roi_heads/fast_rcnn.py:145-194, 196-260
//   (cross_entropy mean over the sampled ROIs, smooth_l1(beta=0) == L1 on the GT-class deltas of the
//    foreground ROIs, both / max(#ROIs, 1); Box2BoxTransform weights (10,10,5,5));
//   detectron2 _log_classification_stats (fast_rcnn.py:184).
//
// pred (R, ldp): columns [0, K] = class logits (K = background), [K+1, K+1+4K) = deltas (k*4+d).
// cls (R): class in [0, K], or < 0 for padding / ignored rows (contribute nothing).
// One wave per ROI row; the 51-wide softmax is a wave reduction.  Forward: a wave walks several rows and keeps its 7 sums in
// registers, the workgroup combines them in LDS and issues at most 7 fp64 atomics (round 3: 2-7 atomics per ROW on the same
// 7 doubles -- ~6000 serialised atomics for 2048 ROIs -- were 72 us).
#include <device_rt.h>

namespace {

constexpr int BOX_LOSS_FWD_BLOCKS = 64;      // forward: 256 waves, 8 rows each for the 2048 sampled ROIs of a 4-image batch

__device__ __forceinline__ void gt_deltas(const float* pb, const float* gb, float wx, float wy, float ww, float wh,
                                          float (&d)[4]) {
    const float sw = pb[2] - pb[0], sh = pb[3] - pb[1], scx = pb[0] + 0.5f * sw, scy = pb[1] + 0.5f * sh;
    const float tw = gb[2] - gb[0], th = gb[3] - gb[1], tcx = gb[0] + 0.5f * tw, tcy = gb[1] + 0.5f * th;
    d[0] = wx * (tcx - scx) / sw; d[1] = wy * (tcy - scy) / sh; d[2] = ww * logf(tw / sw); d[3] = wh * logf(th / sh);
}

// MODE 0: forward sums.  sums: [ce, reg, n_rows, n_fg, n_accurate, n_fg_accurate, n_false_negative]
// MODE 1: dpred (R, ldp) = g_cls * dCE/n + g_reg * dL1/n   (whole row written, zeros elsewhere)
template <int MODE>
__global__ void __launch_bounds__(256) box_loss_kernel(const float* __restrict__ pred, int ldp, int R, int K,
                                                       const int* __restrict__ cls, const float* __restrict__ prop,
                                                       const float* __restrict__ gt, const int* __restrict__ gt_row,
                                                       float wx, float wy, float ww, float wh, double* __restrict__ sums,
                                                       const float* __restrict__ g_cls, const float* __restrict__ g_reg,
                                                       float* __restrict__ dpred) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double acc[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int r = blockIdx.x * 4 + wave; r < R; r += (int)gridDim.x * 4) {
        const float* row = pred + (long)r * ldp;
        const int c = cls[r];
        if (MODE == 1) {
            float* drow = dpred + (long)r * ldp;
            for (int j = lane; j < ldp; j += 64) drow[j] = 0.f;
        }
        if (c < 0) continue;
        const bool fg = c < K;
        // softmax over K+1 logits (K+1 <= 64 * 2 handled by two slots per lane)
        const float x0 = lane <= K ? row[lane] : -INFINITY;
        const float x1 = (lane + 64) <= K ? row[lane + 64] : -INFINITY;
        const float m = wave_max(fmaxf(x0, x1));
        const float e0 = lane <= K ? expf(x0 - m) : 0.f, e1 = (lane + 64) <= K ? expf(x1 - m) : 0.f;
        const float den = wave_sum(e0 + e1);
        if (MODE == 0) {
            const float xc = row[c];
            const float ce = logf(den) + m - xc;
            // argmax (first maximum), for the logged accuracies
            float bv = x0; int bi = lane;
            if (x1 > bv) { bv = x1; bi = lane + 64; }
            for (int s = 32; s >= 1; s >>= 1) {
                const float ov = __shfl_xor(bv, s, 64);
                const int oi = __shfl_xor(bi, s, 64);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            float reg = 0.f;
            if (fg) {
                float gd[4];
                gt_deltas(prop + 4 * r, gt + 4 * gt_row[r], wx, wy, ww, wh, gd);
                const float* dp = row + (K + 1) + 4 * c;
                reg = fabsf(dp[0] - gd[0]) + fabsf(dp[1] - gd[1]) + fabsf(dp[2] - gd[2]) + fabsf(dp[3] - gd[3]);
            }
            acc[0] += (double)ce;
            acc[2] += 1.0;
            if (fg) {
                acc[1] += (double)reg;
                acc[3] += 1.0;
                if (bi == c) acc[5] += 1.0;
                if (bi == K) acc[6] += 1.0;
            }
            if (bi == c) acc[4] += 1.0;
        } else {
            float* drow = dpred + (long)r * ldp;
            const float n = (float)fmax(sums[2], 1.0);
            const float gc = g_cls[0] / n, gr = g_reg[0] / n;
            if (lane <= K) drow[lane] = (e0 / den - (lane == c ? 1.f : 0.f)) * gc;
            if (lane + 64 <= K) drow[lane + 64] = (e1 / den - ((lane + 64) == c ? 1.f : 0.f)) * gc;
            if (fg && lane < 4) {
                float gd[4];
                gt_deltas(prop + 4 * r, gt + 4 * gt_row[r], wx, wy, ww, wh, gd);
                const float df = row[(K + 1) + 4 * c + lane] - gd[lane];
                drow[(K + 1) + 4 * c + lane] = (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * gr;
            }
        }
    }
    if (MODE == 0) {
        __shared__ double part[4][7];
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 7; ++q) part[wave][q] = acc[q];
        }
        __syncthreads();
        if (threadIdx.x < 7) {
            const double v = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
            if (v != 0.0) atomicAdd(&sums[threadIdx.x], v);
        }
    }
}

// FastRCNNOutputLayers.predict_boxes_for_gt_classes (detectron2; called at roi_heads.py:276-289): Box2BoxTransform.apply_deltas
// of the deltas of each row's (clamped) GT class on its proposal box; no clipping.  Padding rows (cls < 0) copy the proposal.
__global__ void __launch_bounds__(256) box_decode_gt_kernel(const float* __restrict__ pred, int ldp, int R, int K,
                                                            const int* __restrict__ cls, const float* __restrict__ prop,
                                                            float wx, float wy, float ww, float wh, float scale_clamp,
                                                            float* __restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float* pb = prop + 4 * (long)r;
    float* o = out + 4 * (long)r;
    int c = cls[r];
    if (c < 0) { o[0] = pb[0]; o[1] = pb[1]; o[2] = pb[2]; o[3] = pb[3]; return; }
    c = c > K - 1 ? K - 1 : c;                              // gt_classes.clamp_(0, K - 1): background rows use the last class
    const float* d = pred + (long)r * ldp + (K + 1) + 4 * c;
    const float w = pb[2] - pb[0], h = pb[3] - pb[1], cx = pb[0] + 0.5f * w, cy = pb[1] + 0.5f * h;
    const float dx = d[0] / wx, dy = d[1] / wy, dw = fminf(d[2] / ww, scale_clamp), dh = fminf(d[3] / wh, scale_clamp);
    const float pcx = dx * w + cx, pcy = dy * h + cy, pw = expf(dw) * w, ph = expf(dh) * h;
    o[0] = pcx - 0.5f * pw; o[1] = pcy - 0.5f * ph; o[2] = pcx + 0.5f * pw; o[3] = pcy + 0.5f * ph;
}

}  // namespace

extern "C" {

// sums: 7 doubles, zeroed here.  loss_cls = sums[0]/max(sums[2],1); loss_box_reg = sums[1]/max(sums[2],1).
int omni_box_loss_fwd(const float* pred, int ldp, int R, int K, const int* cls, const float* prop, const float* gt,
                      const int* gt_row, float wx, float wy, float ww, float wh, double* sums, void* stream) {
    if (R < 0 || K <= 0 || K + 1 > 128 || ldp < 5 * K + 1) return OMNI_ERR_ARG;
    omni_memset_async(sums, 0, sizeof(double) * 7, (hipStream_t)stream);
    if (R == 0) return OMNI_OK;
    const int blocks = (R + 3) / 4 < BOX_LOSS_FWD_BLOCKS ? (R + 3) / 4 : BOX_LOSS_FWD_BLOCKS;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(box_loss_kernel<0>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, pred, ldp,
                       R, K, cls, prop, gt, gt_row, wx, wy, ww, wh, sums, (const float*)nullptr, (const float*)nullptr,
                       (float*)nullptr);
    return omni_launch_status();
}

// dpred (R, ldp) overwritten; sums from the forward call; g_cls / g_reg device scalars.
int omni_box_loss_bwd(const float* pred, int ldp, int R, int K, const int* cls, const float* prop, const float* gt,
                      const int* gt_row, float wx, float wy, float ww, float wh, const double* sums, const float* g_cls,
                      const float* g_reg, float* dpred, void* stream) {
    if (R < 0 || K <= 0 || K + 1 > 128 || ldp < 5 * K + 1) return OMNI_ERR_ARG;
    if (R == 0) return OMNI_OK;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(box_loss_kernel<1>), dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, pred, ldp,
                       R, K, cls, prop, gt, gt_row, wx, wy, ww, wh, const_cast<double*>(sums), g_cls, g_reg, dpred);
    return omni_launch_status();
}

// out (R, 4): the predicted box of each row's GT class (TRAIN_ON_PRED_BOXES, roi_heads.py:283-289).
int omni_box_decode_gt_class(const float* pred, int ldp, int R, int K, const int* cls, const float* prop, float wx, float wy,
                             float ww, float wh, float scale_clamp, float* out, void* stream) {
    if (R < 0 || K <= 0 || ldp < 5 * K + 1) return OMNI_ERR_ARG;
    if (R == 0) return OMNI_OK;
    hipLaunchKernelGGL(box_decode_gt_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, pred, ldp, R, K, cls, prop,
                       wx, wy, ww, wh, scale_clamp, out);
    return omni_launch_status();
}

}  // extern "C"
