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


// =================================================================================================================
// Omni3Deval.accumulate (omni3d_evaluation.py:1172-1313): precision / recall / score tables [T, R, K, A, M].
//   For every (category k, range a, maxDets m): the detections of all evaluated images, each image's list cut to m, are
//   merged in descending score order (stable), tp / fp cumulated, precision made monotone from the right, and sampled at the
//   R recall thresholds with searchsorted(side='left').  The merge order (a stable sort by (category, -score) of all
//   detections) is prepared once by the caller; one 64-lane wave per (k, a, m, t) then makes two passes over the category's
//   list: a forward count and a BACKWARD sweep that carries the running maximum of the precision (= the monotone envelope)
//   and, at the c-th true positive, fills the thresholds in ((c-1)/npig, c/npig].  Doubles and the same expressions as numpy.
// =================================================================================================================
struct AccP {
    const int* order;            // (N) detection index (into the group-concatenated arrays) by sorted position
    const int* cat_off;          // (K + 1) ranges of `order` per category
    const int* rank;             // (sumD) rank of a detection inside its (image, category) list (descending score)
    const double* score;         // (sumD)
    const int* dt_match;         // (A, T, sumD) >= 0 matched
    const unsigned char* dt_ig;  // (A, T, sumD)
    const int* npig;             // (K, A) number of non-ignored ground truths
    const int* has_e;            // (K) any evaluated image holds a gt or dt of the category
    const double* rec_thrs;      // (R) ascending
    const int* max_dets;         // (M)
    int K, A, M, T, R, sumD;
    double* precision;           // (T, R, K, A, M)
    double* recall;              // (T, K, A, M)
    double* scores;              // (T, R, K, A, M)
};

__device__ __forceinline__ double shfl_down_d(double v, int d) { return __shfl_down(v, (unsigned)d, 64); }

__global__ void __launch_bounds__(64) eval_accumulate_kernel(AccP p) {
    int id = blockIdx.x;
    const int t = id % p.T; id /= p.T;
    const int m = id % p.M; id /= p.M;
    const int a = id % p.A;
    const int k = id / p.A;
    const int lane = threadIdx.x;
    if (!p.has_e[k]) return;                                   // E empty: the -1 initialisation stays (:1243-1244)
    const int npig = p.npig[k * p.A + a];
    if (npig == 0) return;                                     // :1259-1260
    const int s0 = p.cat_off[k], s1 = p.cat_off[k + 1], maxdet = p.max_dets[m];
    const int* dtm = p.dt_match + ((long)a * p.T + t) * p.sumD;
    const unsigned char* dtg = p.dt_ig + ((long)a * p.T + t) * p.sumD;
    auto pidx = [&](int r) { return ((((long)t * p.R + r) * p.K + k) * p.A + a) * p.M + m; };
    for (int r = lane; r < p.R; r += 64) { p.precision[pidx(r)] = 0.0; p.scores[pidx(r)] = 0.0; }
    // ---- pass 1: totals over the included detections (rank < maxDet)
    int tp_tot = 0, fp_tot = 0, nd = 0, first = 0x7fffffff;
    for (int s = s0 + lane; s < s1; s += 64) {
        const int d = p.order[s];
        if (p.rank[d] < maxdet) {
            ++nd;
            first = min(first, s);
            const bool ig = dtg[d] != 0, mt = dtm[d] >= 0;
            tp_tot += (mt && !ig);
            fp_tot += (!mt && !ig);
        }
    }
    tp_tot = wave_sum_i(tp_tot);
    fp_tot = wave_sum_i(fp_tot);
    nd = wave_sum_i(nd);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) first = min(first, __shfl_xor(first, d, 64));
    if (lane == 0) p.recall[(((long)t * p.K + k) * p.A + a) * p.M + m] = nd ? (double)tp_tot / npig : 0.0;     // :1278-1282
    if (nd == 0) return;
    // ---- pass 2: backward sweep
    const double eps = 2.220446049250313e-16;                 // np.spacing(1)
    int tp_after = 0, fp_after = 0;
    double carry_max = -1.0;
    const int nchunk = (s1 - s0 + 63) / 64;
    for (int c = nchunk - 1; c >= 0; --c) {
        const int s = s0 + c * 64 + lane;
        bool incl = false, is_tp = false, is_fp = false;
        double sc = 0.0;
        if (s < s1) {
            const int d = p.order[s];
            if (p.rank[d] < maxdet) {
                incl = true;
                const bool ig = dtg[d] != 0, mt = dtm[d] >= 0;
                is_tp = mt && !ig;
                is_fp = !mt && !ig;
                sc = p.score[d];
            }
        }
        const unsigned long long tpm = __ballot(is_tp), fpm = __ballot(is_fp);
        const unsigned long long higher = lane == 63 ? 0ull : (~0ull << (lane + 1));
        const int tp_i = tp_tot - tp_after - __popcll(tpm & higher);      // cumulative counts up to and including this element
        const int fp_i = fp_tot - fp_after - __popcll(fpm & higher);
        double env = incl ? (double)tp_i / ((double)fp_i + (double)tp_i + eps) : -1.0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {                    // inclusive suffix maximum over the lanes >= this one
            const double o = shfl_down_d(env, d);
            if (lane + d < 64) env = fmax(env, o);
        }
        env = fmax(env, carry_max);
        if (is_tp) {
            const double rc = (double)tp_i / npig, rc_prev = (double)(tp_i - 1) / npig;
            int lo = 0, hi = p.R;                               // jlo = #thr <= rc_prev
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (p.rec_thrs[mid] <= rc_prev) lo = mid + 1; else hi = mid; }
            const int jlo = lo;
            lo = 0; hi = p.R;                                   // jhi = #thr <= rc
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (p.rec_thrs[mid] <= rc) lo = mid + 1; else hi = mid; }
            for (int j = jlo; j < lo; ++j) { p.precision[pidx(j)] = env; p.scores[pidx(j)] = sc; }
        }
        tp_after += __popcll(tpm);
        fp_after += __popcll(fpm);
        carry_max = __shfl(env, 0, 64);                        // lane 0 holds the maximum over this chunk and everything after it
    }
    // thresholds <= 0 (recThrs[0] = 0): searchsorted gives index 0 = the first included element
    if (lane == 0) {
        const double sc0 = p.score[p.order[first]];
        for (int j = 0; j < p.R && p.rec_thrs[j] <= 0.0; ++j) { p.precision[pidx(j)] = carry_max; p.scores[pidx(j)] = sc0; }
    }
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


// Omni3Deval.accumulate on the device: precision / scores (T,R,K,A,M) and recall (T,K,A,M), pre-filled with -1 by the caller.
int omni_eval_accumulate(const int* order, const int* cat_off, const int* rank, const double* score, const int* dt_match,
                         const unsigned char* dt_ignore, const int* npig, const int* has_e, const double* rec_thrs,
                         const int* max_dets, int K, int A, int M, int T, int R, int sumD, double* precision, double* recall,
                         double* scores, void* stream) {
    if (K < 0 || A <= 0 || M <= 0 || T <= 0 || R <= 0 || sumD < 0) return OMNI_ERR_ARG;
    if (K == 0) return OMNI_OK;
    AccP p{order, cat_off, rank, score, dt_match, dt_ignore, npig, has_e, rec_thrs, max_dets, K, A, M, T, R, sumD, precision, recall, scores};
    hipLaunchKernelGGL(eval_accumulate_kernel, dim3((unsigned)((long)K * A * M * T)), dim3(64), 0, (hipStream_t)stream, p);
    return omni_launch_status();
}

}  // extern "C"
