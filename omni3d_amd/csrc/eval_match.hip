// eval_match.hip -- the greedy detection <-> ground-truth matching of the Omni3D evaluator for ALL (image, category)
// groups, area (depth) ranges and IoU thresholds in one launch.
//
// Reference: Omni3Deval.evaluateImg, /root/reference/cubercnn/evaluation/omni3d_evaluation.py:1433-1551 (3D mode, eval_prox
// off), called from a Python triple loop over categories x area ranges x images (:1346-1351).  Integer / index work: results
// are identical to the reference's, including its tie rules:
//   * ground truths are ordered "ignored last" by a STABLE sort (:1463);
//   * a detection (descending score order) takes the unmatched ground truth with the largest IoU >= min(t, 1 - 1e-10),
//     the LATER one among equal IoUs (`<` at :1512), searching the non-ignored ones first and the ignored ones only if none
//     of those qualifies (the `break` at :1508);
//   * an unmatched detection outside the depth range is ignored (:1528-1532).
// One 64-lane wave per (group, range, threshold): the detections are walked sequentially, the ground truths of a group
// are spread over the lanes and reduced with shuffles.  Comparisons are made in double like numpy's.
#include <device_rt.h>

namespace {

constexpr int EVAL_MAXG = 1024;

struct EvalP {
    const float* ious;        // ragged: group g holds a (D_g, G_g) row-major matrix at iou_off[g]
    const long long* iou_off; // (ngroups)
    const int* dt_off;        // (ngroups + 1)
    const int* gt_off;        // (ngroups + 1)
    const int* gt_ignore;     // (sumG)
    const float* gt_range;    // (sumG)
    const float* dt_range;    // (sumD)
    const float* areas;       // (A, 2)
    const double* thrs;       // (T)
    int ngroups, A, T, sumD, sumG;
    int* dt_match;            // (A, T, sumD) original gt index within the group or -1
    int* gt_match;            // (A, T, sumG) dt index within the group or -1, by ORIGINAL gt position
    unsigned char* dt_ignore; // (A, T, sumD)
    int* gt_order;            // (A, sumG) stable ignore-last permutation (sorted position -> original index)
    unsigned char* gt_ig;     // (A, sumG) `_ignore` per original gt
};

__global__ void __launch_bounds__(64) eval_match_kernel(EvalP p) {
    __shared__ int s_order[EVAL_MAXG];
    __shared__ int s_gtm[EVAL_MAXG];
    __shared__ unsigned char s_ig[EVAL_MAXG];   // by sorted position
    __shared__ int s_m;
    const int lane = threadIdx.x;
    int b = blockIdx.x;
    const int ti = b % p.T; b /= p.T;
    const int ai = b % p.A;
    const int gi = b / p.A;
    const int d0 = p.dt_off[gi], D = p.dt_off[gi + 1] - d0;
    const int g0 = p.gt_off[gi], G = p.gt_off[gi + 1] - g0;
    const float lo = p.areas[2 * ai], hi = p.areas[2 * ai + 1];
    // ---- `_ignore` flags and the stable ignore-last order ----
    int n0 = 0;   // number of non-ignored ground truths
    for (int base = 0; base < G; base += 64) {
        const int g = base + lane;
        bool ig = false;
        if (g < G) ig = p.gt_ignore[g0 + g] != 0 || p.gt_range[g0 + g] < lo || p.gt_range[g0 + g] > hi;
        n0 += __popcll(__ballot(g < G && !ig));
    }
    int c0 = 0, c1 = 0;
    for (int base = 0; base < G; base += 64) {
        const int g = base + lane;
        bool in = g < G, ig = false;
        if (in) ig = p.gt_ignore[g0 + g] != 0 || p.gt_range[g0 + g] < lo || p.gt_range[g0 + g] > hi;
        const unsigned long long m0 = __ballot(in && !ig), m1 = __ballot(in && ig);
        const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        if (in) {
            const int pos = ig ? n0 + c1 + __popcll(m1 & below) : c0 + __popcll(m0 & below);
            s_order[pos] = g;
            s_ig[pos] = ig ? 1 : 0;
            s_gtm[pos] = -1;
            if (ti == 0) p.gt_ig[(long)ai * p.sumG + g0 + g] = ig ? 1 : 0;
        }
        c0 += __popcll(m0);
        c1 += __popcll(m1);
    }
    __syncthreads();
    if (ti == 0)
        for (int q = lane; q < G; q += 64) p.gt_order[(long)ai * p.sumG + g0 + q] = s_order[q];
    const long ob = ((long)ai * p.T + ti);
    const double t = p.thrs[ti];
    const double thr = t < 1.0 - 1e-10 ? t : 1.0 - 1e-10;
    const float* io = p.ious + p.iou_off[gi];
    // ---- detections in score order ----
    for (int d = 0; d < D; ++d) {
        int m = -1;
        if (G > 0) {
            for (int seg = 0; seg < 2 && m < 0; ++seg) {             // non-ignored first, ignored only if nothing matched
                const int qb = seg == 0 ? 0 : n0, qe = seg == 0 ? n0 : G;
                double best = -1.0;
                int bpos = -1;
                for (int q = qb + lane; q < qe; q += 64) {
                    if (s_gtm[q] >= 0) continue;
                    const double v = (double)io[(long)d * G + s_order[q]];
                    if (v >= thr && v >= best) { best = v; bpos = q; }     // ascending q per lane: later wins ties
                }
#pragma unroll
                for (int sh = 32; sh >= 1; sh >>= 1) {
                    const double ov = __shfl_xor(best, sh, 64);
                    const int op = __shfl_xor(bpos, sh, 64);
                    if (ov > best || (ov == best && op > bpos)) { best = ov; bpos = op; }
                }
                m = bpos;
            }
        }
        if (lane == 0) {
            unsigned char ig;
            if (m >= 0) {
                s_gtm[m] = d;
                ig = s_ig[m];
                p.dt_match[ob * p.sumD + d0 + d] = s_order[m];
            } else {
                const float r = p.dt_range[d0 + d];
                ig = (r < lo || r > hi) ? 1 : 0;
                p.dt_match[ob * p.sumD + d0 + d] = -1;
            }
            p.dt_ignore[ob * p.sumD + d0 + d] = ig;
        }
        __syncthreads();
    }
    for (int q = lane; q < G; q += 64) p.gt_match[ob * p.sumG + g0 + s_order[q]] = s_gtm[q];
}

}  // namespace

extern "C" {

int omni_eval_match(const float* ious, const long long* iou_off, const int* dt_off, const int* gt_off, const int* gt_ignore,
                    const float* gt_range, const float* dt_range, const float* areas, const double* thrs, int ngroups, int A,
                    int T, int sumD, int sumG, int max_gt, int* dt_match, int* gt_match, unsigned char* dt_ignore, int* gt_order,
                    unsigned char* gt_ig, void* stream) {
    if (ngroups < 0 || A <= 0 || T <= 0 || sumD < 0 || sumG < 0 || max_gt > EVAL_MAXG) return OMNI_ERR_ARG;
    if (ngroups == 0) return OMNI_OK;
    EvalP p{ious, iou_off, dt_off, gt_off, gt_ignore, gt_range, dt_range, areas, thrs, ngroups, A, T, sumD, sumG,
            dt_match, gt_match, dt_ignore, gt_order, gt_ig};
    hipLaunchKernelGGL(eval_match_kernel, dim3((unsigned)((long)ngroups * A * T)), dim3(64), 0, (hipStream_t)stream, p);
    return omni_launch_status();
}

}  // extern "C"
