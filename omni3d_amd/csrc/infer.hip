// infer.hip -- `fast_rcnn_inference` for the WHOLE batch with static shapes and no host round trip.
//
// Reference: cubercnn/modeling/roi_heads/fast_rcnn.py:57-116 (`fast_rcnn_inference_single_image`, called per image from a
// Python loop :33-54): drop rows with non-finite boxes / scores, clip the per-class boxes to the image, keep the
// (roi, class) pairs with score > SCORE_THRESH_TEST, per-class NMS (detectron2 batched_nms = one NMS over boxes shifted
// by class * (max coordinate + 1)), keep the DETECTIONS_PER_IMAGE best.  Each step there is a boolean-mask gather with a
// device->host sync; here:
//   det_score_kernel      one wave per ROI: softmax over the K+1 logits, Box2BoxTransform.apply_deltas for every class,
//                         clip, validity; writes the dense score matrix (B, P*K) with -inf for everything filtered out;
//   omni_topk_rows        (csrc/select_nms.hip) sorts the candidates of every image: stable descending order of the
//                         row-major (roi, class) list = torch.sort(stable) over the masked list of the reference;
//   det_nms_boxes_kernel  gathers the candidates' boxes, finds the image's maximum coordinate, adds the class offsets;
//   omni_nms_sorted       (csrc/select_nms.hip) one NMS problem per image;
//   det_compact_kernel    the first `topk` survivors of every image in score order -> fixed (B, topk) slots + counts.
#include <device_rt.h>
#include <math.h>

namespace {

constexpr float SCALE_CLAMP = 4.135166556742356f;       // log(1000 / 16), detectron2 Box2BoxTransform

// pred (B*P, ld) = [K+1 logits | 4K deltas | pad]; rois (B*P, 4); count (B) valid proposals per image; image_hw (B, 2)
__global__ void __launch_bounds__(64) det_score_kernel(const float* __restrict__ pred, int ld, const float* __restrict__ rois,
                                                        const int* __restrict__ count, const int* __restrict__ image_hw, int B, int P,
                                                        int K, float wx, float wy, float ww, float wh, float thr,
                                                        float* __restrict__ scores, float* __restrict__ probs, float* __restrict__ boxes) {
    const int row = blockIdx.x, b = row / P, pidx = row - b * P, lane = threadIdx.x;
    const float NEG = -INFINITY;
    float* srow = scores + ((long)b * P + pidx) * K;
    if (pidx >= count[b]) {
        for (int c = lane; c < K; c += 64) { srow[c] = NEG; probs[(long)row * K + c] = 0.f; }
        return;
    }
    const float* pr = pred + (long)row * ld;
    // softmax over K + 1 logits (F.softmax, fast_rcnn.py predict_probs)
    float mx = NEG;
    for (int c = lane; c <= K; c += 64) mx = fmaxf(mx, pr[c]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int c = lane; c <= K; c += 64) sum += expf(pr[c] - mx);
    sum = wave_sum(sum);
    const float x1 = rois[row * 4 + 0], y1 = rois[row * 4 + 1], x2 = rois[row * 4 + 2], y2 = rois[row * 4 + 3];
    const float w = x2 - x1, h = y2 - y1, cx = x1 + 0.5f * w, cy = y1 + 0.5f * h;
    const float H = (float)image_hw[2 * b], W = (float)image_hw[2 * b + 1];
    bool finite = true;
    for (int c0 = 0; c0 < K + 1; c0 += 64) {            // every probability and every class box of the row must be finite (:72-76)
        const int c = c0 + lane;
        if (c <= K) finite = finite && isfinite(expf(pr[c] - mx) / sum);
    }
    for (int c0 = 0; c0 < K; c0 += 64) {
        const int c = c0 + lane;
        if (c < K) {
            const float* d = pr + (K + 1) + 4 * c;
            const float dx = d[0] / wx, dy = d[1] / wy, dw = fminf(d[2] / ww, SCALE_CLAMP), dh = fminf(d[3] / wh, SCALE_CLAMP);
            const float pcx = dx * w + cx, pcy = dy * h + cy, pw = expf(dw) * w, ph = expf(dh) * h;
            float bx[4] = {pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph};
            finite = finite && isfinite(bx[0]) && isfinite(bx[1]) && isfinite(bx[2]) && isfinite(bx[3]);
            float* o = boxes + ((long)row * K + c) * 4;
            o[0] = fminf(fmaxf(bx[0], 0.f), W); o[1] = fminf(fmaxf(bx[1], 0.f), H);      // Boxes.clip (:81)
            o[2] = fminf(fmaxf(bx[2], 0.f), W); o[3] = fminf(fmaxf(bx[3], 0.f), H);
        }
    }
    const bool row_ok = __all(finite);
    for (int c = lane; c < K; c += 64) {
        const float pv = expf(pr[c] - mx) / sum;
        probs[(long)row * K + c] = pv;
        srow[c] = (row_ok && pv > thr) ? pv : NEG;                                       // :88-92
    }
}

// one 1024-thread workgroup per image: gather the sorted candidates' boxes, class offset = class * (max coordinate + 1)
__global__ void __launch_bounds__(1024) det_nms_boxes_kernel(const float* __restrict__ boxes, const float* __restrict__ vals,
                                                              const int* __restrict__ idx, int PK, int K, int cap,
                                                              float* __restrict__ nms_boxes, int* __restrict__ valid) {
    __shared__ float red[16];
    const int b = blockIdx.x, t = threadIdx.x;
    float mx = -INFINITY;
    for (int j = t; j < cap; j += 1024) {
        const int i = idx[(long)b * cap + j];
        const bool ok = i >= 0 && vals[(long)b * cap + j] > -INFINITY;
        valid[(long)b * cap + j] = ok ? 1 : 0;
        if (ok) {
            const float* s = boxes + ((long)b * PK + i) * 4;
            mx = fmaxf(mx, fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])));
        }
    }
    mx = wave_max(mx);
    if ((t & 63) == 0) red[t >> 6] = mx;
    __syncthreads();
    mx = red[0];
    for (int k = 1; k < 16; ++k) mx = fmaxf(mx, red[k]);
    const float stride = mx + 1.f;                                                       // torchvision batched_nms
    for (int j = t; j < cap; j += 1024) {
        const int i = idx[(long)b * cap + j];
        float* o = nms_boxes + ((long)b * cap + j) * 4;
        if (valid[(long)b * cap + j]) {
            const float* s = boxes + ((long)b * PK + i) * 4;
            const float off = (float)(i % K) * stride;
            o[0] = s[0] + off; o[1] = s[1] + off; o[2] = s[2] + off; o[3] = s[3] + off;
        } else {
            o[0] = o[1] = o[2] = o[3] = 0.f;
        }
    }
}

// one wave per image: the first `topk` kept candidates in score order -> fixed slots
__global__ void __launch_bounds__(64) det_compact_kernel(const int* __restrict__ keep, const int* __restrict__ valid,
                                                          const float* __restrict__ vals, const int* __restrict__ idx,
                                                          const float* __restrict__ boxes, int PK, int K, int cap, int topk,
                                                          float* __restrict__ out_box, float* __restrict__ out_score,
                                                          int* __restrict__ out_cls, int* __restrict__ out_roi, int* __restrict__ out_count) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int n = 0;
    for (int base = 0; base < cap && n < topk; base += 64) {
        const int j = base + lane;
        const bool k = j < cap && keep[(long)b * cap + j] != 0 && valid[(long)b * cap + j] != 0;
        const unsigned long long m = __ballot(k);
        const int slot = n + __popcll(m & ((1ull << lane) - 1ull));
        if (k && slot < topk) {
            const int i = idx[(long)b * cap + j];
            const float* s = boxes + ((long)b * PK + i) * 4;
            float* o = out_box + ((long)b * topk + slot) * 4;
            o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = s[3];
            out_score[(long)b * topk + slot] = vals[(long)b * cap + j];
            out_cls[(long)b * topk + slot] = i % K;
            out_roi[(long)b * topk + slot] = i / K;
        }
        n += __popcll(m);
    }
    n = n < topk ? n : topk;
    for (int s = n + lane; s < topk; s += 64) {
        float* o = out_box + ((long)b * topk + s) * 4;
        o[0] = o[1] = 0.f; o[2] = o[3] = 1.f;                 // harmless dummy box for the fixed-shape cube head
        out_score[(long)b * topk + s] = 0.f;
        out_cls[(long)b * topk + s] = 0;
        out_roi[(long)b * topk + s] = 0;
    }
    if (lane == 0) out_count[b] = n;
}

}  // namespace

extern "C" {

int omni_det_scores(const float* pred, int ld, const float* rois, const int* count, const int* image_hw, int B, int P, int K,
                    float wx, float wy, float ww, float wh, float score_thresh, float* scores, float* probs, float* boxes,
                    void* stream) {
    if (B < 0 || P <= 0 || K <= 0 || ld < 5 * K + 1) return OMNI_ERR_ARG;
    if (B == 0) return OMNI_OK;
    hipLaunchKernelGGL(det_score_kernel, dim3((unsigned)(B * P)), dim3(64), 0, (hipStream_t)stream, pred, ld, rois, count, image_hw, B, P, K,
                       wx, wy, ww, wh, score_thresh, scores, probs, boxes);
    return omni_launch_status();
}

int omni_det_nms_boxes(const float* boxes, const float* vals, const int* idx, int B, int PK, int K, int cap, float* nms_boxes,
                       int* valid, void* stream) {
    if (B < 0 || PK <= 0 || K <= 0 || cap <= 0) return OMNI_ERR_ARG;
    if (B == 0) return OMNI_OK;
    hipLaunchKernelGGL(det_nms_boxes_kernel, dim3((unsigned)B), dim3(1024), 0, (hipStream_t)stream, boxes, vals, idx, PK, K, cap, nms_boxes, valid);
    return omni_launch_status();
}

int omni_det_compact(const int* keep, const int* valid, const float* vals, const int* idx, const float* boxes, int B, int PK, int K,
                     int cap, int topk, float* out_box, float* out_score, int* out_cls, int* out_roi, int* out_count, void* stream) {
    if (B < 0 || PK <= 0 || K <= 0 || cap <= 0 || topk <= 0) return OMNI_ERR_ARG;
    if (B == 0) return OMNI_OK;
    hipLaunchKernelGGL(det_compact_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, keep, valid, vals, idx, boxes, PK, K, cap, topk,
                       out_box, out_score, out_cls, out_roi, out_count);
    return omni_launch_status();
}

}  // extern "C"
