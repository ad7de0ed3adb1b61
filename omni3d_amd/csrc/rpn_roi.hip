// rpn_roi.hip -- anchor matching / labelling, the RPN "IoUness" losses, proposal decoding and the
// ROI-head proposal labelling + sampling, for gfx950.
//
// Reference call sites (paths under /root/reference/cubercnn/modeling):
//   pairwise_iou / pairwise_ioa + Matcher           proposal_generator/rpn.py:62-63,100; roi_heads/roi_heads.py:881-893
//   RPNWithIgnore.label_and_sample_anchors          proposal_generator/rpn.py:41-110
//   subsample_labels (IoU-weighted multinomial)     proposal_generator/rpn.py:275-328
//   _dense_box_regression_loss_with_uncertainty     proposal_generator/rpn.py:206-273 ("IoUness": BCE to the
//       anchor<->GT IoU weighted by that IoU + IoU-weighted L1 on the deltas), losses / normaliser :198-203
//   RPN._decode_proposals + clip + empty filter     detectron2 RPN / find_top_rpn_proposals (Base.yaml:49-54)
//   ROIHeads3D.label_and_sample_proposals           roi_heads/roi_heads.py:862-929, _sample_proposals :826-860
//
// Everything here is small integer / elementwise work over <= 65k anchors and <= ~1k proposals per
// image: the kernels are latency bound; the point of writing them for the GPU is to keep the whole
// training step on the device without host round trips (the reference does ~16 .item()/tolist()
// syncs here).  GT boxes of an image are staged in LDS; per-GT maxima use integer atomics on the
// float bit pattern (IoU >= 0), so results do not depend on execution order.
//
// RPN head tensors: level l holds Y_l (B, H_l, W_l, 16) fp32 NHWC = [3 objectness logits | 12 deltas
// (a*4+d) | 1 pad]; anchor index = a_off[l] + (y*W_l + x)*3 + a  (detectron2 order H, W, A).
#include <device_rt.h>
#include "philox.h"
#pragma clang fp contract(off)

namespace {

constexpr int MAXG = 1024;  // GT boxes per image staged in LDS (round 3: 256 -> 1024; the ROI sampler then holds 56 KB of LDS)
constexpr int MAXL = 8;
constexpr int RPN_A = 3;    // anchors per location
constexpr int RPN_C = 16;   // channels of the fused RPN head output

struct Levels {
    float* y[MAXL];       // (B, hw, 16)
    int hw[MAXL];         // H_l * W_l
    int a_off[MAXL + 1];  // anchor offset of each level
    int nlev;
};

__device__ __forceinline__ float box_area(float4 b) { return (b.z - b.x) * (b.w - b.y); }
__device__ __forceinline__ float box_inter(float4 a, float4 b) {
    const float w = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.f);
    const float h = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.f);
    return w * h;
}
// detectron2 pairwise_iou: inter > 0 ? inter / (a1 + a2 - inter) : 0
__device__ __forceinline__ float iou_d2(float4 a, float4 b) {
    const float inter = box_inter(a, b);
    return inter > 0.f ? inter / (box_area(a) + box_area(b) - inter) : 0.f;
}
// detectron2 pairwise_ioa(boxes1, boxes2): inter / area(boxes2)
__device__ __forceinline__ float ioa_d2(float4 b1, float4 b2) {
    const float inter = box_inter(b1, b2);
    return inter > 0.f ? inter / box_area(b2) : 0.f;
}
__device__ __forceinline__ float4 ldbox(const float* p) { return *reinterpret_cast<const float4*>(p); }

__global__ void pairwise_iou_kernel(const float* __restrict__ b1, int N, const float* __restrict__ b2, int M, int mode,
                                    float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * M) return;
    const float4 a = ldbox(b1 + 4 * (i / M)), b = ldbox(b2 + 4 * (i % M));
    out[i] = mode == 0 ? iou_d2(a, b) : ioa_d2(a, b);
}

// ---- RPN matching, pass 1: per anchor best GT (first maximum), per GT best IoU (atomic max) ----
__global__ void __launch_bounds__(256) rpn_match1_kernel(const float* __restrict__ anchors, int A,
                                                         const float* __restrict__ gt, const int* __restrict__ gt_off,
                                                         float* __restrict__ mval, int* __restrict__ midx,
                                                         int* __restrict__ gt_best_bits) {
    __shared__ float4 sg[MAXG];
    __shared__ int sbest[MAXG];
    const int n = blockIdx.y;
    const int g0 = gt_off[n], G = min(gt_off[n + 1] - g0, MAXG);
    for (int g = threadIdx.x; g < G; g += blockDim.x) { sg[g] = ldbox(gt + 4 * (g0 + g)); sbest[g] = 0; }
    __syncthreads();
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a < A) {
        const float4 ab = ldbox(anchors + 4 * a);
        float best = -1.f;
        int bi = 0;
        for (int g = 0; g < G; ++g) {
            const float v = iou_d2(sg[g], ab);
            if (v > best) { best = v; bi = g; }
            if (v > 0.f) atomicMax(&sbest[g], __float_as_int(v));
        }
        if (G == 0) best = 0.f;
        mval[(long)n * A + a] = best;
        midx[(long)n * A + a] = bi;
    }
    __syncthreads();
    for (int g = threadIdx.x; g < G; g += blockDim.x)
        if (sbest[g] > 0) atomicMax(&gt_best_bits[g0 + g], sbest[g]);
}

// ---- pass 2: Matcher labels (+ low-quality matches), argmax anchor per GT, sampling keys --------
// label = val < lo ? l0 : (val < hi ? l1 : l2); low-quality: every anchor whose IoU equals a GT's
// best IoU becomes 1.  keys = (matched_iou + eps) / E  for the positive / negative candidate sets
// (torch.multinomial without replacement == top-k of w / Exp(1)).
__global__ void __launch_bounds__(256) rpn_match2_kernel(const float* __restrict__ anchors, int A,
                                                         const float* __restrict__ gt, const int* __restrict__ gt_off,
                                                         const float* __restrict__ mval, const int* __restrict__ gt_best_bits,
                                                         float lo, float hi, int l0, int l1, int l2, int allow_low,
                                                         const float* __restrict__ expo, float eps,
                                                         signed char* __restrict__ mlabel, int* __restrict__ gt_best_idx,
                                                         float* __restrict__ key_pos, float* __restrict__ key_neg,
                                                         long long* __restrict__ draw_state, int* __restrict__ ticket) {
    __shared__ float4 sg[MAXG];
    __shared__ int sbest[MAXG];
    __shared__ int sarg[MAXG];
    __shared__ unsigned long long s_ctr;
    const int n = blockIdx.y;
    // expo == nullptr: the Exp(1) variate of (image, anchor) is drawn here (philox.h) instead of read from a pre-filled array
    if (expo == nullptr && threadIdx.x == 0) s_ctr = omni_draw_begin(draw_state);
    const int g0 = gt_off[n], G = min(gt_off[n + 1] - g0, MAXG);
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        sg[g] = ldbox(gt + 4 * (g0 + g));
        sbest[g] = gt_best_bits[g0 + g];
        sarg[g] = 0x7fffffff;
    }
    __syncthreads();
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a < A) {
        const float4 ab = ldbox(anchors + 4 * a);
        const float v = mval[(long)n * A + a];
        int label = v < lo ? l0 : (v < hi ? l1 : l2);
        if (G == 0) label = l0;
        for (int g = 0; g < G; ++g) {
            const float q = iou_d2(sg[g], ab);
            if (__float_as_int(q) == sbest[g]) {   // equals this GT's best quality (0 included when best is 0)
                if (allow_low) label = 1;   // ties included, zero-quality rows too (detectron2 Matcher)
                atomicMin(&sarg[g], a);
            }
        }
        mlabel[(long)n * A + a] = (signed char)label;
        const float e = expo != nullptr ? expo[(long)n * A + a] : omni_exp1((unsigned long long)draw_state[0], s_ctr, (unsigned)n, (unsigned)a);
        key_pos[(long)n * A + a] = (label != -1 && label != 0) ? (v + eps) / e : -INFINITY;
        key_neg[(long)n * A + a] = (label == 0) ? (v + eps) / e : -INFINITY;
    }
    __syncthreads();
    for (int g = threadIdx.x; g < G; g += blockDim.x)
        if (sarg[g] != 0x7fffffff) atomicMin(&gt_best_idx[g0 + g], sarg[g]);
    if (expo == nullptr && threadIdx.x == 0) omni_draw_end(draw_state, ticket, s_ctr, (int)(gridDim.x * gridDim.y));
}

// ---- final labels of one image (rpn.py:79-105).  labels pre-filled with -1. -----------------------
__global__ void __launch_bounds__(256) rpn_finalize_kernel(const float* __restrict__ anchors, int A,
                                                           const int* __restrict__ gt_off, const float* __restrict__ ign,
                                                           const int* __restrict__ ign_off,
                                                           const signed char* __restrict__ mlabel,
                                                           const int* __restrict__ gt_best_idx,
                                                           const float* __restrict__ pos_val, const int* __restrict__ pos_idx,
                                                           const float* __restrict__ neg_val, const int* __restrict__ neg_idx,
                                                           int kpos, int kneg, int batch_per_image, float ignore_thresh,
                                                           signed char* __restrict__ labels, int* __restrict__ counts) {
    __shared__ int s_npos, s_nneg_avail;
    const int n = blockIdx.x, t = threadIdx.x;
    signed char* L = labels + (long)n * A;
    if (t == 0) { s_npos = 0; s_nneg_avail = 0; }
    __syncthreads();
    int c = 0, d = 0;
    for (int j = t; j < kpos; j += blockDim.x) c += (pos_val[(long)n * kpos + j] > -INFINITY) ? 1 : 0;
    for (int j = t; j < kneg; j += blockDim.x) d += (neg_val[(long)n * kneg + j] > -INFINITY) ? 1 : 0;
    atomicAdd(&s_npos, c);
    atomicAdd(&s_nneg_avail, d);
    __syncthreads();
    const int npos = s_npos;
    int nneg = batch_per_image - npos;
    nneg = nneg < s_nneg_avail ? nneg : s_nneg_avail;
    if (nneg < 0) nneg = 0;
    for (int j = t; j < npos; j += blockDim.x) L[pos_idx[(long)n * kpos + j]] = 1;
    for (int j = t; j < nneg; j += blockDim.x) L[neg_idx[(long)n * kneg + j]] = 0;
    __syncthreads();
    // best anchor of every GT is always positive if the matcher labelled it positive (rpn.py:75,83-84)
    const int g0 = gt_off[n], G = gt_off[n + 1] - g0;
    for (int g = t; g < G; g += blockDim.x) {
        const int a = gt_best_idx[g0 + g];
        if (a >= 0 && a < A && mlabel[(long)n * A + a] == 1) L[a] = 1;
    }
    __syncthreads();
    // sampled background inside an ignore region -> -1 (rpn.py:93-105; needs > 1 background anchor)
    const int i0 = ign_off[n], NI = ign_off[n + 1] - i0;
    if (NI > 0 && nneg > 1) {
        for (int j = t; j < nneg; j += blockDim.x) {
            const int a = neg_idx[(long)n * kneg + j];
            const float4 ab = ldbox(anchors + 4 * a);
            float m = 0.f;
            for (int q = 0; q < NI; ++q) m = fmaxf(m, ioa_d2(ldbox(ign + 4 * (i0 + q)), ab));
            if (m >= ignore_thresh) L[a] = -1;
        }
    }
    if (t == 0 && counts) { counts[2 * n] = npos; counts[2 * n + 1] = nneg; }
}

__device__ __forceinline__ void anchor_to_level(const Levels& lv, int a, int& l, int& loc, int& k) {
    l = 0;
    while (l + 1 < lv.nlev && a >= lv.a_off[l + 1]) ++l;
    const int r = a - lv.a_off[l];
    loc = r / RPN_A;
    k = r - loc * RPN_A;
}

// compact objectness logits (B, A) from the level tensors
__global__ void rpn_gather_logits_kernel(Levels lv, int B, int A, float* __restrict__ logits) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * A) return;
    const int n = (int)(i / A), a = (int)(i % A);
    int l, loc, k;
    anchor_to_level(lv, a, l, loc, k);
    logits[i] = lv.y[l][((long)n * lv.hw[l] + loc) * RPN_C + k];
}

// ---- IoUness losses.  MODE 0: forward sums; MODE 1: gradients into dY (every element written: no pre-zeroing) ----
// sums[0] = sum BCE(x, t)*t, sums[1] = sum |dpred - dgt|_1 * t, sums[2] = #pos, sums[3] = #neg (label 0),
// sums[4] = sum sigmoid(x) over pos, sums[5] = sum sigmoid(x) over non-pos.
// Forward: grid-stride over the anchors, per-wave float sums, one fp64 combine per workgroup in LDS and at most 6 atomics per
// workgroup (round 3: one atomic set per WAVE -- 4092 waves hitting the same 6 doubles -- was 84 us of serialised atomics).
constexpr int RPN_LOSS_MAX_BLOCKS = 512;
__device__ __forceinline__ void rpn_loss_flush(const float (&s)[6], double* __restrict__ sums) {
    __shared__ double part[4][6];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const float v = wave_sum(s[q]);
        if (lane == 0) part[wave][q] = (double)v;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const double v = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        if (v != 0.0) atomicAdd(&sums[threadIdx.x], v);
    }
}
// zero the channels of one location that no anchor owns (the padding of the fused head output), by the anchor-0 thread
__device__ __forceinline__ void rpn_zero_pad(float* __restrict__ d, long base, int k) {
    if (k == 0) {
#pragma unroll
        for (int c = RPN_A * 5; c < RPN_C; ++c) d[base + c] = 0.f;
    }
}
template <int MODE>
__global__ void __launch_bounds__(256) rpn_loss_kernel(Levels lv, Levels dlv, int B, int A,
                                                       const float* __restrict__ anchors,
                                                       const signed char* __restrict__ labels,
                                                       const int* __restrict__ midx, const float* __restrict__ gt,
                                                       const int* __restrict__ gt_off, double* __restrict__ sums,
                                                       const float* __restrict__ g_cls, const float* __restrict__ g_loc,
                                                       float inv_norm) {
    float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const long tot = (long)B * A;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / A), a = (int)(i % A);
        int l, loc, k;
        anchor_to_level(lv, a, l, loc, k);
        const long base = ((long)n * lv.hw[l] + loc) * RPN_C;
        const float x = lv.y[l][base + k];
        const int lab = labels[i];
        const float sg = 1.f / (1.f + expf(-x));
        float dcls = 0.f, dloc[4] = {0.f, 0.f, 0.f, 0.f};
        if (lab == 1) {
            const float4 ab = ldbox(anchors + 4 * a);
            const float4 gb = ldbox(gt + 4 * (gt_off[n] + midx[i]));
            const float inter = box_inter(ab, gb);
            const float t = inter / (box_area(ab) + box_area(gb) - inter);   // matched_pairwise_iou (rpn.py:330-353)
            // Box2BoxTransform.get_deltas, weights (1,1,1,1)
            const float sw = ab.z - ab.x, sh = ab.w - ab.y, scx = ab.x + 0.5f * sw, scy = ab.y + 0.5f * sh;
            const float tw = gb.z - gb.x, th = gb.w - gb.y, tcx = gb.x + 0.5f * tw, tcy = gb.y + 0.5f * th;
            const float gd[4] = {(tcx - scx) / sw, (tcy - scy) / sh, logf(tw / sw), logf(th / sh)};
            if (MODE == 0) {
                const float bce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
                float l1 = 0.f;
#pragma unroll
                for (int d = 0; d < 4; ++d) l1 += fabsf(lv.y[l][base + RPN_A + k * 4 + d] - gd[d]);
                s[0] += bce * t; s[1] += l1 * t; s[2] += 1.f; s[4] += sg;
            } else {
                dcls = (sg - t) * t * inv_norm * g_cls[0];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const float df = lv.y[l][base + RPN_A + k * 4 + d] - gd[d];
                    const float sgn = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
                    dloc[d] = sgn * t * inv_norm * g_loc[0];
                }
            }
        } else if (MODE == 0) {
            s[3] += lab == 0 ? 1.f : 0.f;
            s[5] += sg;
        }
        if (MODE == 1) {
            float* d = dlv.y[l];
            d[base + k] = dcls;
#pragma unroll
            for (int q = 0; q < 4; ++q) d[base + RPN_A + k * 4 + q] = dloc[q];
            rpn_zero_pad(d, base, k);
        }
    }
    if (MODE == 0) rpn_loss_flush(s, sums);
}

// MODEL.RPN.OBJECTNESS_UNCERTAINTY 'none' (rpn.py:181-195, detectron2's RPN losses): objectness = BCE-with-logits against the 0 / 1
// anchor labels over every sampled anchor, localisation = L1 on the foreground deltas without the IoU weight.  Same sums layout.
template <int MODE>
__global__ void __launch_bounds__(256) rpn_loss_plain_kernel(Levels lv, Levels dlv, int B, int A, const float* __restrict__ anchors,
                                                             const signed char* __restrict__ labels, const int* __restrict__ midx,
                                                             const float* __restrict__ gt, const int* __restrict__ gt_off,
                                                             double* __restrict__ sums, const float* __restrict__ g_cls,
                                                             const float* __restrict__ g_loc, float inv_norm) {
    float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const long tot = (long)B * A;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / A), a = (int)(i % A);
        int l, loc, k;
        anchor_to_level(lv, a, l, loc, k);
        const long base = ((long)n * lv.hw[l] + loc) * RPN_C;
        const float x = lv.y[l][base + k];
        const int lab = labels[i];
        const float sg = 1.f / (1.f + expf(-x));
        float dcls = 0.f, dloc[4] = {0.f, 0.f, 0.f, 0.f};
        if (lab >= 0) {                                    // valid_mask = gt_labels >= 0
            const float t = lab == 1 ? 1.f : 0.f;
            if (MODE == 0) s[0] += fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
            else dcls = (sg - t) * inv_norm * g_cls[0];
        }
        if (lab == 1) {
            const float4 ab = ldbox(anchors + 4 * a);
            const float4 gb = ldbox(gt + 4 * (gt_off[n] + midx[i]));
            const float sw = ab.z - ab.x, sh = ab.w - ab.y, scx = ab.x + 0.5f * sw, scy = ab.y + 0.5f * sh;
            const float tw = gb.z - gb.x, th = gb.w - gb.y, tcx = gb.x + 0.5f * tw, tcy = gb.y + 0.5f * th;
            const float gd[4] = {(tcx - scx) / sw, (tcy - scy) / sh, logf(tw / sw), logf(th / sh)};
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const float df = lv.y[l][base + RPN_A + k * 4 + d] - gd[d];
                if (MODE == 0) s[1] += fabsf(df);
                else dloc[d] = (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * inv_norm * g_loc[0];
            }
            if (MODE == 0) { s[2] += 1.f; s[4] += sg; }
        } else if (MODE == 0) {
            s[3] += lab == 0 ? 1.f : 0.f;
            s[5] += sg;
        }
        if (MODE == 1) {
            float* d = dlv.y[l];
            d[base + k] = dcls;
#pragma unroll
            for (int q = 0; q < 4; ++q) d[base + RPN_A + k * 4 + q] = dloc[q];
            rpn_zero_pad(d, base, k);
        }
    }
    if (MODE == 0) rpn_loss_flush(s, sums);
}

// ---- decode the selected anchors: Box2BoxTransform.apply_deltas + clip + validity -----------------
// slot j of image n in level l: anchor a_off[l] + idx.  out boxes (B, Ktot, 4), valid (B, Ktot) = finite & w>0 & h>0.
__global__ void rpn_decode_kernel(Levels lv, int B, int Ktot, const int* __restrict__ slot_level,
                                  const int* __restrict__ idx, const float* __restrict__ anchors,
                                  const int* __restrict__ image_hw, float scale_clamp, float min_size,
                                  float* __restrict__ boxes, int* __restrict__ valid) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * Ktot) return;
    const int n = (int)(i / Ktot), j = (int)(i % Ktot);
    const int l = slot_level[j];
    const int r = idx[i];
    float4 ob = make_float4(0.f, 0.f, 0.f, 0.f);
    int ok = 0;
    if (r >= 0) {
        const int loc = r / RPN_A, k = r - loc * RPN_A;
        const float4 ab = ldbox(anchors + 4 * (lv.a_off[l] + r));
        const float* dp = lv.y[l] + ((long)n * lv.hw[l] + loc) * RPN_C + RPN_A + k * 4;
        const float w = ab.z - ab.x, h = ab.w - ab.y, cx = ab.x + 0.5f * w, cy = ab.y + 0.5f * h;
        const float dx = dp[0], dy = dp[1], dw = fminf(dp[2], scale_clamp), dh = fminf(dp[3], scale_clamp);
        const float pcx = dx * w + cx, pcy = dy * h + cy, pw = expf(dw) * w, ph = expf(dh) * h;
        float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph, x2 = pcx + 0.5f * pw, y2 = pcy + 0.5f * ph;
        const bool fin = isfinite(x1) && isfinite(y1) && isfinite(x2) && isfinite(y2);
        const float H = (float)image_hw[2 * n], W = (float)image_hw[2 * n + 1];
        x1 = fminf(fmaxf(x1, 0.f), W); x2 = fminf(fmaxf(x2, 0.f), W);
        y1 = fminf(fmaxf(y1, 0.f), H); y2 = fminf(fmaxf(y2, 0.f), H);
        ob = make_float4(x1, y1, x2, y2);
        ok = (fin && (x2 - x1) > min_size && (y2 - y1) > min_size) ? 1 : 0;
    }
    *reinterpret_cast<float4*>(boxes + 4 * i) = ob;
    valid[i] = ok;
}

// ---- ROI heads: label and sample the proposals of one image (roi_heads.py:862-929) -----------------
// candidates = proposals[0..np) ++ valid GT boxes; Matcher(thr) without low-quality; background inside
// an ignore region -> -1; IoU-weighted sampling (<= nfg_max foreground, fill with background).
// One 1024-thread workgroup per image; candidates <= 2048.
constexpr int ROI_T = 1024;
constexpr int ROI_MAXC = 2048;

__device__ void bitonic_desc(unsigned long long* s, int n, int t, int nthreads) {
    for (int size = 2; size <= n; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = t; i < (n >> 1); i += nthreads) {
                const int lo = 2 * i - (i & (stride - 1)), hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long a = s[lo], b = s[hi];
                if ((a < b) == desc) { s[lo] = b; s[hi] = a; }
            }
            __syncthreads();
        }
}
__device__ __forceinline__ unsigned long long fkey(float f, int idx) {
    unsigned u = __float_as_uint(f);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)idx);
}

__global__ void __launch_bounds__(ROI_T) roi_sample_kernel(const float* __restrict__ prop_boxes, const int* __restrict__ prop_count,
                                                           int pmax, const float* __restrict__ gt, const int* __restrict__ gt_cls,
                                                           const int* __restrict__ gt_off, const float* __restrict__ ign,
                                                           const int* __restrict__ ign_off, const float* __restrict__ expo,
                                                           float iou_thr, float ignore_thresh, float eps, int num_classes,
                                                           int batch_per_image, int nfg_max, int append_gt,
                                                           float* __restrict__ out_boxes, int* __restrict__ out_cls,
                                                           int* __restrict__ out_gt, float* __restrict__ out_iou,
                                                           int* __restrict__ out_counts, long long* __restrict__ draw_state,
                                                           int* __restrict__ ticket, int* __restrict__ out_row, int first,
                                                           float* __restrict__ first_boxes, int* __restrict__ first_cls,
                                                           int* __restrict__ first_row) {
    __shared__ float4 sg[MAXG];
    __shared__ unsigned long long kf[ROI_MAXC], kb[ROI_MAXC];
    __shared__ short s_cls[ROI_MAXC];
    __shared__ short s_m[ROI_MAXC];
    __shared__ int s_nbg, s_nfg;
    __shared__ unsigned long long s_ctr;
    const int n = blockIdx.x, t = threadIdx.x;
    if (expo == nullptr && t == 0) s_ctr = omni_draw_begin(draw_state);
    const int g0 = gt_off[n], G = min(gt_off[n + 1] - g0, MAXG);
    const int np = min(prop_count ? prop_count[n] : pmax, pmax);
    const int nc = min(np + (append_gt ? G : 0), ROI_MAXC);
    for (int g = t; g < G; g += ROI_T) sg[g] = ldbox(gt + 4 * (g0 + g));
    if (t == 0) { s_nbg = 0; s_nfg = 0; }
    __syncthreads();
    const int i0 = ign_off[n], NI = ign_off[n + 1] - i0;
    // pass A: match, count background
    float my_iou[ROI_MAXC / ROI_T];
    int my_lab[ROI_MAXC / ROI_T], my_m[ROI_MAXC / ROI_T];
    float4 my_box[ROI_MAXC / ROI_T];
#pragma unroll
    for (int r = 0; r < ROI_MAXC / ROI_T; ++r) {
        const int i = t + r * ROI_T;
        my_lab[r] = -2; my_iou[r] = 0.f; my_m[r] = 0; my_box[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < nc) {
            const float4 b = i < np ? ldbox(prop_boxes + 4 * ((long)n * pmax + i)) : sg[i - np];
            float best = -1.f;
            int bi = 0;
            for (int g = 0; g < G; ++g) {
                const float v = iou_d2(sg[g], b);
                if (v > best) { best = v; bi = g; }
            }
            if (G == 0) best = 0.f;
            my_box[r] = b; my_iou[r] = best; my_m[r] = bi;
            my_lab[r] = (G > 0 && best >= iou_thr) ? 1 : 0;
            if (my_lab[r] == 0) atomicAdd(&s_nbg, 1);
        }
    }
    __syncthreads();
    const int nbg_matcher = s_nbg;
    __syncthreads();
    if (t == 0) s_nbg = 0;
    __syncthreads();
    // pass B: ignore regions, classes, sampling keys
#pragma unroll
    for (int r = 0; r < ROI_MAXC / ROI_T; ++r) {
        const int i = t + r * ROI_T;
        unsigned long long k_f = 0ull, k_b = 0ull;
        if (i < nc) {
            int lab = my_lab[r];
            if (lab == 0 && NI > 0 && nbg_matcher > 1) {
                float m = 0.f;
                for (int q = 0; q < NI; ++q) m = fmaxf(m, ioa_d2(ldbox(ign + 4 * (i0 + q)), my_box[r]));
                if (m >= ignore_thresh) lab = -1;
            }
            int cls;
            if (G > 0) cls = lab == 1 ? gt_cls[g0 + my_m[r]] : (lab == 0 ? num_classes : -1);
            else cls = num_classes;
            s_cls[i] = (short)cls;
            s_m[i] = (short)my_m[r];
            const float e = expo != nullptr ? expo[(long)n * ROI_MAXC + i] : omni_exp1((unsigned long long)draw_state[0], s_ctr, (unsigned)n, (unsigned)i);
            const float key = (my_iou[r] + eps) / e;
            if (cls != -1 && cls != num_classes) { k_f = fkey(key, i); atomicAdd(&s_nfg, 1); }
            else if (cls == num_classes) { k_b = fkey(key, i); atomicAdd(&s_nbg, 1); }
        }
        if (i < ROI_MAXC) { kf[i] = k_f; kb[i] = k_b; }
    }
    __syncthreads();
    bitonic_desc(kf, ROI_MAXC, t, ROI_T);
    bitonic_desc(kb, ROI_MAXC, t, ROI_T);
    const int nfg = min(s_nfg, nfg_max);
    const int nbgs = min(s_nbg, batch_per_image - nfg);
    const int ns = nfg + nbgs;
    for (int j = t; j < batch_per_image; j += ROI_T) {
        const long o = (long)n * batch_per_image + j;
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        int cls = -2, row = -1;   // padding slot, ignored by every consumer
        float best = 0.f;
        if (j < ns) {
            const unsigned long long k = j < nfg ? kf[j] : kb[j - nfg];
            const int i = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
            b = i < np ? ldbox(prop_boxes + 4 * ((long)n * pmax + i)) : sg[i - np];
            cls = s_cls[i];
            row = G > 0 ? g0 + s_m[i] : -1;
            for (int g = 0; g < G; ++g) best = fmaxf(best, iou_d2(sg[g], b));
        }
        *reinterpret_cast<float4*>(out_boxes + 4 * o) = b;
        out_cls[o] = cls;
        out_gt[o] = row;
        out_iou[o] = best;
        // round 6: what the loss kernels index the ground truth with (background / padding marker -1 -> row 0, `sgt.clamp(min=0)`
        // before) and the contiguous copies of the first `first` slots the cube head works on -- three clamp / slice-copy launches less
        const int row0 = row < 0 ? 0 : row;
        if (out_row != nullptr) out_row[o] = row0;
        if (j < first) {
            const long of = (long)n * first + j;
            *reinterpret_cast<float4*>(first_boxes + 4 * of) = b;
            first_cls[of] = cls;
            first_row[of] = row0;
        }
    }
    if (t == 0) { out_counts[2 * n] = nfg; out_counts[2 * n + 1] = nbgs; }
    if (expo == nullptr && t == 0) omni_draw_end(draw_state, ticket, s_ctr, (int)gridDim.x);
}

Levels make_levels(const void* const* ptrs, const int* hw, int nlev) {
    Levels lv;
    int off = 0;
    for (int l = 0; l < MAXL; ++l) { lv.y[l] = nullptr; lv.hw[l] = 0; lv.a_off[l] = 0; }
    for (int l = 0; l < nlev; ++l) {
        lv.y[l] = (float*)ptrs[l];
        lv.hw[l] = hw[l];
        lv.a_off[l] = off;
        off += hw[l] * RPN_A;
    }
    for (int l = nlev; l <= MAXL; ++l) lv.a_off[l] = off;
    lv.nlev = nlev;
    return lv;
}

}  // namespace

namespace {

// ---- glue of find_top_rpn_proposals after NMS (detectron2; rpn.py call site): two launches instead of eleven element-wise ones ----
// masked[i] = keep[i] ? scores[i] : -inf   (the suppressed / invalid candidates leave the post-NMS ranking)
__global__ void rpn_mask_scores_kernel(const float* __restrict__ scores, const int* __restrict__ keep, long n, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = keep[i] != 0 ? scores[i] : -INFINITY;
}
// proposals of image b: prop[b][p] = boxes[b][top_i[b][p]] where the p-th ranked score is a real one (> -inf), zeros after them;
// count[b] = number of real ones.  One workgroup per image.
__global__ void __launch_bounds__(256) rpn_collect_kernel(const float* __restrict__ boxes, const float* __restrict__ top_v,
                                                          const int* __restrict__ top_i, int N, int P, float* __restrict__ prop,
                                                          int* __restrict__ count) {
    __shared__ int part[4];
    const int b = blockIdx.x, t = threadIdx.x;
    int c = 0;
    for (int p = t; p < P; p += 256) {
        const bool ok = top_v[(long)b * P + p] > -INFINITY;
        int gi = top_i[(long)b * P + p];
        gi = gi < 0 ? 0 : gi;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) v = ldbox(boxes + ((long)b * N + gi) * 4);
        *reinterpret_cast<float4*>(prop + ((long)b * P + p) * 4) = v;
        c += ok ? 1 : 0;
    }
    for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, 64);
    if ((t & 63) == 0) part[t >> 6] = c;
    __syncthreads();
    if (t == 0) count[b] = part[0] + part[1] + part[2] + part[3];
}

}  // namespace

static inline long rpn_loss_blocks(long tot) {
    const long b = (tot + 255) / 256;
    return b < RPN_LOSS_MAX_BLOCKS ? b : RPN_LOSS_MAX_BLOCKS;
}

extern "C" {

// mode 0: IoU (detectron2 pairwise_iou), mode 1: IoA = inter / area(boxes2).  out (N, M).
int omni_pairwise_iou(const float* boxes1, int N, const float* boxes2, int M, int mode, float* out, void* stream) {
    if (N < 0 || M < 0) return OMNI_ERR_ARG;
    const long tot = (long)N * M;
    if (tot == 0) return OMNI_OK;
    hipLaunchKernelGGL(pairwise_iou_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, boxes1,
                       N, boxes2, M, mode, out);
    return omni_launch_status();
}

// RPN anchor matching for a batch.  gt: concatenated VALID GT boxes, gt_off (B+1).  expo: (B, A)
// Exp(1) variates (the multinomial's randomness).  Outputs per (image, anchor): matched IoU/index,
// Matcher label, sampling keys; per GT: best anchor index.  gt_best_bits: (G) int scratch.
// expo == NULL: the variates are drawn inside the kernel from draw_state (2 int64: seed, draw counter) / ticket (1 int32, zero) --
// see philox.h; the draw counter is advanced by the launch.
int omni_rpn_match_draw(const float* anchors, int A, const float* gt, const int* gt_off, int B, int G, float thr_lo,
                        float thr_hi, int l0, int l1, int l2, int allow_low_quality, const float* expo, long long* draw_state,
                        int* ticket, float eps, float* matched_val, int* matched_idx, signed char* match_label, int* gt_best_bits,
                        int* gt_best_idx, float* key_pos, float* key_neg, void* stream) {
    if (A <= 0 || B <= 0 || G < 0 || (expo == nullptr && (draw_state == nullptr || ticket == nullptr))) return OMNI_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (G > 0) {
        omni_memset_async(gt_best_bits, 0, sizeof(int) * G, st);
        omni_memset_async(gt_best_idx, 0x7f, sizeof(int) * G, st);
    }
    dim3 grid((A + 255) / 256, B);
    hipLaunchKernelGGL(rpn_match1_kernel, grid, dim3(256), 0, st, anchors, A, gt, gt_off, matched_val, matched_idx,
                       gt_best_bits);
    hipLaunchKernelGGL(rpn_match2_kernel, grid, dim3(256), 0, st, anchors, A, gt, gt_off, (const float*)matched_val,
                       (const int*)gt_best_bits, thr_lo, thr_hi, l0, l1, l2, allow_low_quality, expo, eps, match_label,
                       gt_best_idx, key_pos, key_neg, draw_state, ticket);
    return omni_launch_status();
}

int omni_rpn_match(const float* anchors, int A, const float* gt, const int* gt_off, int B, int G, float thr_lo,
                   float thr_hi, int l0, int l1, int l2, int allow_low_quality, const float* expo, float eps,
                   float* matched_val, int* matched_idx, signed char* match_label, int* gt_best_bits, int* gt_best_idx,
                   float* key_pos, float* key_neg, void* stream) {
    if (expo == nullptr) return OMNI_ERR_ARG;
    return omni_rpn_match_draw(anchors, A, gt, gt_off, B, G, thr_lo, thr_hi, l0, l1, l2, allow_low_quality, expo, nullptr, nullptr, eps,
                               matched_val, matched_idx, match_label, gt_best_bits, gt_best_idx, key_pos, key_neg, stream);
}

// Final anchor labels {-1,0,1} (B, A) from the sampled candidates (sorted top-k lists of the keys).
int omni_rpn_finalize_labels(const float* anchors, int A, int B, const int* gt_off, const float* ign, const int* ign_off,
                             const signed char* match_label, const int* gt_best_idx, const float* pos_val,
                             const int* pos_idx, const float* neg_val, const int* neg_idx, int kpos, int kneg,
                             int batch_per_image, float ignore_thresh, signed char* labels, int* counts, void* stream) {
    if (A <= 0 || B <= 0) return OMNI_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    omni_memset_async(labels, 0xFF, (size_t)A * B, st);
    hipLaunchKernelGGL(rpn_finalize_kernel, dim3(B), dim3(256), 0, st, anchors, A, gt_off, ign, ign_off, match_label,
                       gt_best_idx, pos_val, pos_idx, neg_val, neg_idx, kpos, kneg, batch_per_image, ignore_thresh, labels,
                       counts);
    return omni_launch_status();
}

int omni_rpn_gather_logits(const void* const* level_ptrs, const int* level_hw, int nlev, int B, float* logits,
                           void* stream) {
    if (nlev <= 0 || nlev > MAXL) return OMNI_ERR_ARG;
    Levels lv = make_levels(level_ptrs, level_hw, nlev);
    const int A = lv.a_off[nlev];
    const long tot = (long)B * A;
    if (tot == 0) return OMNI_OK;
    hipLaunchKernelGGL(rpn_gather_logits_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, lv,
                       B, A, logits);
    return omni_launch_status();
}

// sums: 6 doubles (zeroed here): [cls, loc, #pos, #neg, sum sigmoid pos, sum sigmoid non-pos]
int omni_rpn_loss_fwd(const void* const* level_ptrs, const int* level_hw, int nlev, int B, const float* anchors,
                      const signed char* labels, const int* matched_idx, const float* gt, const int* gt_off, double* sums,
                      void* stream) {
    if (nlev <= 0 || nlev > MAXL) return OMNI_ERR_ARG;
    Levels lv = make_levels(level_ptrs, level_hw, nlev);
    const int A = lv.a_off[nlev];
    hipStream_t st = (hipStream_t)stream;
    omni_memset_async(sums, 0, sizeof(double) * 6, st);
    const long tot = (long)B * A;
    if (tot == 0) return OMNI_OK;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(rpn_loss_kernel<0>), dim3((unsigned)rpn_loss_blocks(tot)), dim3(256), 0, st, lv, lv,
                       B, A, anchors, labels, matched_idx, gt, gt_off, sums, (const float*)nullptr, (const float*)nullptr,
                       0.f);
    return omni_launch_status();
}

// grads of (g_cls * cls_sum + g_loc * loc_sum) * inv_norm wrt the level tensors, written into
// dlevel_ptrs (same shapes; every element is written).  g_cls / g_loc are device scalars.
int omni_rpn_loss_bwd(const void* const* level_ptrs, const void* const* dlevel_ptrs, const int* level_hw, int nlev, int B,
                      const float* anchors, const signed char* labels, const int* matched_idx, const float* gt,
                      const int* gt_off, const float* g_cls, const float* g_loc, float inv_norm, void* stream) {
    if (nlev <= 0 || nlev > MAXL) return OMNI_ERR_ARG;
    Levels lv = make_levels(level_ptrs, level_hw, nlev);
    Levels dlv = make_levels(dlevel_ptrs, level_hw, nlev);
    const int A = lv.a_off[nlev];
    hipStream_t st = (hipStream_t)stream;
    const long tot = (long)B * A;                   // one thread per anchor; together they write every element of the level tensors
    if (tot == 0) return OMNI_OK;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(rpn_loss_kernel<1>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, lv, dlv,
                       B, A, anchors, labels, matched_idx, gt, gt_off, (double*)nullptr, g_cls, g_loc, inv_norm);
    return omni_launch_status();
}

// The same two entry points for MODEL.RPN.OBJECTNESS_UNCERTAINTY 'none' (plain 0 / 1 objectness targets, unweighted L1).
int omni_rpn_loss_plain_fwd(const void* const* level_ptrs, const int* level_hw, int nlev, int B, const float* anchors,
                            const signed char* labels, const int* matched_idx, const float* gt, const int* gt_off, double* sums,
                            void* stream) {
    if (nlev <= 0 || nlev > MAXL) return OMNI_ERR_ARG;
    Levels lv = make_levels(level_ptrs, level_hw, nlev);
    const int A = lv.a_off[nlev];
    hipStream_t st = (hipStream_t)stream;
    omni_memset_async(sums, 0, sizeof(double) * 6, st);
    const long tot = (long)B * A;
    if (tot == 0) return OMNI_OK;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(rpn_loss_plain_kernel<0>), dim3((unsigned)rpn_loss_blocks(tot)), dim3(256), 0, st, lv, lv,
                       B, A, anchors, labels, matched_idx, gt, gt_off, sums, (const float*)nullptr, (const float*)nullptr, 0.f);
    return omni_launch_status();
}

int omni_rpn_loss_plain_bwd(const void* const* level_ptrs, const void* const* dlevel_ptrs, const int* level_hw, int nlev, int B,
                            const float* anchors, const signed char* labels, const int* matched_idx, const float* gt,
                            const int* gt_off, const float* g_cls, const float* g_loc, float inv_norm, void* stream) {
    if (nlev <= 0 || nlev > MAXL) return OMNI_ERR_ARG;
    Levels lv = make_levels(level_ptrs, level_hw, nlev);
    Levels dlv = make_levels(dlevel_ptrs, level_hw, nlev);
    const int A = lv.a_off[nlev];
    hipStream_t st = (hipStream_t)stream;
    const long tot = (long)B * A;                   // one thread per anchor; together they write every element of the level tensors
    if (tot == 0) return OMNI_OK;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(rpn_loss_plain_kernel<1>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, lv, dlv,
                       B, A, anchors, labels, matched_idx, gt, gt_off, (double*)nullptr, g_cls, g_loc, inv_norm);
    return omni_launch_status();
}

// masked (n) = keep ? scores : -inf
int omni_rpn_mask_scores(const float* scores, const int* keep, long long n, float* masked, void* stream) {
    if (n < 0) return OMNI_ERR_ARG;
    if (n == 0) return OMNI_OK;
    hipLaunchKernelGGL(rpn_mask_scores_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, scores, keep, (long)n, masked);
    return omni_launch_status();
}

// boxes (B, N, 4); top_v / top_i (B, P): the post-NMS ranking (sorted scores, -inf / -1 padded) -> prop (B, P, 4), count (B)
int omni_rpn_collect(const float* boxes, const float* top_v, const int* top_i, int B, int N, int P, float* prop, int* count, void* stream) {
    if (B < 0 || N < 0 || P < 0) return OMNI_ERR_ARG;
    if (B == 0) return OMNI_OK;
    hipLaunchKernelGGL(rpn_collect_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, boxes, top_v, top_i, N, P, prop, count);
    return omni_launch_status();
}

// Decode the per-level top-k anchors of every image.  slot_level (Ktot): level of each slot;
// idx (B, Ktot): anchor index inside that level (-1 = empty slot); image_hw (B, 2) ints.
int omni_rpn_decode(const void* const* level_ptrs, const int* level_hw, int nlev, int B, int Ktot, const int* slot_level,
                    const int* idx, const float* anchors, const int* image_hw, float scale_clamp, float min_size,
                    float* boxes, int* valid, void* stream) {
    if (nlev <= 0 || nlev > MAXL) return OMNI_ERR_ARG;
    Levels lv = make_levels(level_ptrs, level_hw, nlev);
    const long tot = (long)B * Ktot;
    if (tot == 0) return OMNI_OK;
    hipLaunchKernelGGL(rpn_decode_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, lv, B,
                       Ktot, slot_level, idx, anchors, image_hw, scale_clamp, min_size, boxes, valid);
    return omni_launch_status();
}

// ROI-head proposal labelling + sampling.  prop_boxes (B, pmax, 4) with prop_count (B) [nullable];
// gt / gt_cls concatenated valid GT with gt_off (B+1); ign / ign_off ignore regions; expo (B, 2048)
// Exp(1) variates.  Outputs (B, batch_per_image): boxes, class (num_classes = background, -2 = padding),
// global GT row (or -1), matched IoU; counts (B, 2) = sampled fg / bg.
// Round 6 form.  expo == NULL: variates drawn in the kernel (draw_state / ticket as in omni_rpn_match_draw).  out_row (B,
// batch_per_image) [nullable]: out_gt with the background / padding marker -1 replaced by row 0 (what the loss kernels index with).
// first > 0: first_boxes (B, first, 4), first_cls / first_row (B, first) = contiguous copies of the first `first` sampled slots of
// every image (the cube head's ROIs, roi_heads.py:341-362: foreground first).
int omni_roi_sample_draw(const float* prop_boxes, const int* prop_count, int B, int pmax, const float* gt, const int* gt_cls,
                         const int* gt_off, const float* ign, const int* ign_off, const float* expo, long long* draw_state, int* ticket,
                         float iou_thr, float ignore_thresh, float eps, int num_classes, int batch_per_image, int nfg_max,
                         int append_gt, float* out_boxes, int* out_cls, int* out_gt, float* out_iou, int* out_counts, int* out_row,
                         int first, float* first_boxes, int* first_cls, int* first_row, void* stream) {
    if (B <= 0 || pmax < 0 || pmax > ROI_MAXC || batch_per_image <= 0 || batch_per_image > ROI_MAXC) return OMNI_ERR_ARG;
    if (expo == nullptr && (draw_state == nullptr || ticket == nullptr)) return OMNI_ERR_ARG;
    if (first < 0 || first > batch_per_image || (first > 0 && (first_boxes == nullptr || first_cls == nullptr || first_row == nullptr)))
        return OMNI_ERR_ARG;
    hipLaunchKernelGGL(roi_sample_kernel, dim3(B), dim3(ROI_T), 0, (hipStream_t)stream, prop_boxes, prop_count, pmax, gt,
                       gt_cls, gt_off, ign, ign_off, expo, iou_thr, ignore_thresh, eps, num_classes, batch_per_image,
                       nfg_max, append_gt, out_boxes, out_cls, out_gt, out_iou, out_counts, draw_state, ticket, out_row, first,
                       first_boxes, first_cls, first_row);
    return omni_launch_status();
}

int omni_roi_sample(const float* prop_boxes, const int* prop_count, int B, int pmax, const float* gt, const int* gt_cls,
                    const int* gt_off, const float* ign, const int* ign_off, const float* expo, float iou_thr,
                    float ignore_thresh, float eps, int num_classes, int batch_per_image, int nfg_max, int append_gt,
                    float* out_boxes, int* out_cls, int* out_gt, float* out_iou, int* out_counts, void* stream) {
    if (expo == nullptr) return OMNI_ERR_ARG;
    return omni_roi_sample_draw(prop_boxes, prop_count, B, pmax, gt, gt_cls, gt_off, ign, ign_off, expo, nullptr, nullptr, iou_thr,
                                ignore_thresh, eps, num_classes, batch_per_image, nfg_max, append_gt, out_boxes, out_cls, out_gt, out_iou,
                                out_counts, nullptr, 0, nullptr, nullptr, nullptr, stream);
}


}  // extern "C"
