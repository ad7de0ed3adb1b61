// select_nms.hip -- index-exact selection kernels: row-wise sorted top-k and greedy NMS.
//
// Replaces (bit-exact on the selected indices):
//   logits.sort(descending)[:k] per level            detectron2 find_top_rpn_proposals (RPN of
//                                                     /root/reference/configs/Base.yaml:49-54)
//   torch.multinomial(w, k) == topk(w / Exp(1), k)   /root/reference/cubercnn/modeling/proposal_generator/rpn.py:318,322
//   torchvision.ops.nms via detectron2 batched_nms   rpn [upstream]; cubercnn/modeling/roi_heads/fast_rcnn.py:105
//
// Top-k: one 1024-thread workgroup per row.  Keys become order-preserving 64-bit integers
// (float bits, then ~index so that equal scores resolve to the LOWER index = a stable descending
// sort); an 8-pass MSB-first radix select finds the k-th largest key exactly, survivors are
// compacted into LDS and bitonic-sorted there.  k <= 8192.  Rows of up to 65536 elements (every row of the training step:
// 49152 anchors of the finest level, 65472 sampling keys per image) are read from memory ONCE: each thread keeps the score bits
// of its <= 64 elements in registers and the radix passes and the compaction run on those (round 3; the passes used to re-read
// the row, up to 8 x 64 dependent L2 round trips per thread -- 132 / 118 / 48 us for the three launches of a step).
//
// NMS: boxes arrive sorted by descending score.  A 64x64-bit suppression matrix tile per
// workgroup (strict IoU > thr, areas (x2-x1)*(y2-y1), no +1), then one wave per problem walks the
// rows in 64-box chunks: intra-chunk resolution from the diagonal word, then the kept rows OR their
// mask rows into the running `removed` words (one 64-bit word per lane).
#include <device_rt.h>
#include <cstdlib>

namespace {

constexpr int TOPK_THREADS = 1024;
constexpr int TOPK_MAXK = 8192;      // 64 KB of LDS keys; the batched-inference candidate sort uses the full size

__device__ __forceinline__ unsigned score_bits(float f) {
    unsigned u = __float_as_uint(f);
    if (f != f) u = 0u;  // NaN sorts last
    else if (f == 0.f) u = 0x80000000u;  // -0.0 and +0.0 compare equal (torch.sort): one key, the index decides
    else u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return u;
}
__device__ __forceinline__ unsigned long long key_of_bits(unsigned u, int idx) {
    return ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)idx);
}
__device__ __forceinline__ unsigned long long make_key(float f, int idx) { return key_of_bits(score_bits(f), idx); }
// key_of_bits(u, idx) >= thr, on the two 32-bit halves (cached rows keep ONE register per element: a 64-bit key per element that
// lives across the passes doubles them)
// sel[slot] = key_of_bits(u, idx) as two 32-bit stores: a 64-bit store makes the register allocator keep every cached score word
// in the odd half of an aligned pair
__device__ __forceinline__ void store_key(unsigned long long* sel, int slot, unsigned u, int idx) {
    unsigned* w = reinterpret_cast<unsigned*>(sel + slot);
    w[0] = 0xFFFFFFFFu - (unsigned)idx;
    w[1] = u;
}
__device__ __forceinline__ bool key_ge(unsigned u, int idx, unsigned thi, unsigned tlo) {
    return u > thi || (u == thi && (0xFFFFFFFFu - (unsigned)idx) >= tlo);
}
__device__ __forceinline__ float key_value(unsigned long long k) {
    unsigned u = (unsigned)(k >> 32);
    u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    return __uint_as_float(u);
}
__device__ __forceinline__ int key_index(unsigned long long k) { return (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull)); }

// Column segments of a row: segment s covers elements [off[s], off[s] + n[s]); one workgroup per (row, segment).
constexpr int TOPK_MAXSEG = 8;
struct SegP {
    int S;
    int off[TOPK_MAXSEG], n[TOPK_MAXSEG];
};

// ---- radix selection machinery (one 1024-thread workgroup) ----------------------------------------------------------------
// Scores cluster (the sign/exponent byte of RPN logits takes 2-3 values), so plain LDS atomics would serialise tens of thousands
// of adds on one bin: each wave first folds its lanes that share the leader's digit into ONE add (4 rounds), the rest fall back
// to per-lane atomics.  Must be called by whole waves (ballots / shuffles).
__device__ __forceinline__ void count_digit(unsigned* hist, bool todo, unsigned digit) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int round = 0; round < 4; ++round) {
        const unsigned long long pending = __ballot(todo);
        if (pending == 0ull) break;
        const int leader = __ffsll((long long)pending) - 1;
        const unsigned d = (unsigned)__shfl((int)digit, leader, 64);
        const unsigned long long same = __ballot(todo && digit == d);
        if (lane == leader) atomicAdd(&hist[d], (unsigned)__popcll(same));
        if (todo && digit == d) todo = false;
    }
    if (todo) atomicAdd(&hist[digit], 1u);
}

struct RadixShared {
    unsigned hist[256];
    unsigned wsum[4];
    unsigned long long prefix;
    int need, done;
};

// After a pass's histogram is complete (and a barrier): the bin that holds the `need`-th largest candidate, found by a suffix scan
// of the 256 bins over 256 threads (round 3: thread 0 used to walk the bins one dependent LDS read at a time, ~3 us per pass).
// Extends the prefix by that digit, lowers `need` by the candidates in the higher bins, sets `done` when every remaining candidate
// is needed.  Contains one barrier; the caller adds one after it.
__device__ __forceinline__ void radix_pick_bin(RadixShared& R, int shift) {
    const int t = threadIdx.x;
    const int need = R.need;                       // (read by everyone before the barrier below, written after it)
    const unsigned long long prefix = R.prefix;
    unsigned v = 0u, incl = 0u;
    if (t < 256) {
        v = R.hist[255 - t];                        // thread t looks at digit 255 - t: inclusive scan = candidates with digit >= d
        incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned u = __shfl_up(incl, o, 64);
            if ((t & 63) >= o) incl += u;
        }
        if ((t & 63) == 63) R.wsum[t >> 6] = incl;
    }
    __syncthreads();
    if (t < 256) {
        for (int w = 0; w < (t >> 6); ++w) incl += R.wsum[w];
        const unsigned above = incl - v;            // candidates in the bins above this one
        // the first bin (from the top) whose cumulative count reaches `need`; digit 0 takes whatever is left
        if (((int)incl >= need || t == 255) && (int)above < need) {
            R.need = need - (int)above;
            R.prefix = prefix | ((unsigned long long)(255 - t) << shift);
            // all remaining candidates share the prefix and every one of them is needed: the lower bytes cannot separate
            // anything any more (distinct scores settle after the 4 score bytes; the index bytes only matter when equal scores
            // straddle the k-th place)
            R.done = ((int)v == need - (int)above) ? 1 : 0;
        }
    }
}

// k-th largest (kth >= 1, at most the number of valid entries) of the 64-bit keys (eu[c] << 32 | el[c]) spread over the threads'
// register lists; -> threshold: exactly kth keys are >= it (keys are unique).  Whole workgroup; barriers inside.
template <int CNT>
__device__ __forceinline__ unsigned long long radix_kth(RadixShared& R, const unsigned (&eu)[CNT], const unsigned (&el)[CNT],
                                                        const bool (&ev)[CNT], int kth) {
    const int t = threadIdx.x;
    if (t == 0) { R.prefix = 0ull; R.need = kth; }
    __syncthreads();
    for (int pass = 7; pass >= 0; --pass) {
        if (t < 256) R.hist[t] = 0u;
        __syncthreads();
        const unsigned long long prefix = R.prefix;
        const unsigned phi = (unsigned)(prefix >> 32), plo = (unsigned)prefix;
        const bool hi = pass >= 4;
        const int sh = hi ? pass * 8 - 32 : pass * 8;
#pragma unroll
        for (int c = 0; c < CNT; ++c) {
            const unsigned w = hi ? eu[c] : el[c], pw = hi ? phi : plo;
            bool todo = ev[c] && (hi || eu[c] == phi);
            if (sh < 24) todo = todo && ((w >> (sh + 8)) == (pw >> (sh + 8)));
            count_digit(R.hist, todo, (w >> sh) & 255u);
        }
        __syncthreads();
        radix_pick_bin(R, pass * 8);
        __syncthreads();
        if (R.done) break;
    }
    const unsigned long long thr = R.prefix;
    __syncthreads();                                 // (the next selection resets the shared state)
    return thr;
}

// keys: (rows, n) with row pitch `pitch` and element stride `estride` (floats).
// NPT > 0: every segment has at most NPT * 1024 elements and a thread caches its NPT score words in registers; 0: streamed.
//
// Cached rows with k <= TOPK_PREFILTER_K take a two-level selection (round 3): every thread picks its own 4 largest elements in
// registers; the k-th largest of those 4096 keys (a radix selection over 4 register entries per thread) is a LOWER bound of the
// row's k-th largest key, and because the per-thread maxima are nearly the row's top 4096, only a few more than k elements pass
// it (measured on RPN logits: ~2300 for k = 2000 of 49152).  Those are compacted into LDS and the exact k-th largest is selected
// among them, again 4 entries per thread.  The radix passes therefore touch 4 entries per thread instead of 48-64, and the
// selection is the same set -- the keys are unique 64-bit (score, index) words throughout.
constexpr int TOPK_NPT = 64;
constexpr int TOPK_PREFILTER_K = 3072, TOPK_PREFILTER_CAND = 4096;
// PREF (needs NPT > 0): the two-level selection; a segment it does not settle (k >= n, or more than TOPK_PREFILTER_CAND
// survivors of the bound: adversarial ties) takes the streamed full selection, so that the cached words are dead after level 2
// (both full selections compiled into one kernel keep them live everywhere and spill).
template <int NPT, bool PREF>
__global__ void __launch_bounds__(TOPK_THREADS) topk_rows_kernel(const float* __restrict__ keys, SegP seg, long pitch,
                                                                 int estride, int k, float* __restrict__ out_val,
                                                                 int* __restrict__ out_idx) {
    __shared__ RadixShared R;
    __shared__ unsigned long long sel[TOPK_MAXK];
    __shared__ int s_count;
    const int t = threadIdx.x;
    const int rowid = (int)blockIdx.x / seg.S, sid = (int)blockIdx.x - rowid * seg.S;
    const int n = seg.n[sid];
    const float* row = keys + (long)rowid * pitch + (long)seg.off[sid] * estride;
    const int kk = k < n ? k : n;
    int kpad = 1;
    while (kpad < kk) kpad <<= 1;
    if (kpad < 2) kpad = 2;

    // element j of this thread is i = j * 1024 + t (the loops below have the same, workgroup-uniform, trip count for every thread:
    // the wave-level aggregation uses ballots / shuffles)
    unsigned ku[NPT > 0 ? NPT : 1];
    if (NPT > 0) {
        // all loads issued back to back through a buffer resource sized to the row (elements past n come back as zeros and are
        // never looked at); a guarded global load is a branch + a full wait per element, 64 serial round trips
        const omni_rsrc_t rr = omni_make_rsrc(row, n > 0 ? (unsigned)(((long)(n - 1) * estride + 1) * 4) : 0u);
        int tt = t;
        OMNI_OPAQUE_V(tt);                     // (a private copy of t per loop: see the note in the pass loop)
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            const int i = j * TOPK_THREADS + tt;
            ku[j] = score_bits(__uint_as_float(omni_bufld1(rr, i < n ? i * estride * 4 : OMNI_OOB)));
        }
    }

    bool selected = false;                      // sel[0 .. kk) already holds the answer set (unsorted), sel[kk .. kpad) zeros
    if (PREF && kk < n && kk <= TOPK_PREFILTER_K) {
        // ---- level 1: per-thread top 4 (ties: the earlier element = lower index = larger key stays ahead) ----
        unsigned bu[4] = {0u, 0u, 0u, 0u};
        int bj[4] = {-1, -1, -1, -1};
        {
            int tt = t, nn = n;
            OMNI_OPAQUE_V(tt);
            OMNI_OPAQUE_S(nn);
#pragma unroll
            for (int j = 0; j < NPT; ++j) {
                if (j * TOPK_THREADS >= nn) break;
                if (j * TOPK_THREADS + tt < nn) {
                    unsigned cu = ku[j];
                    OMNI_OPAQUE_V(cu);          // a copy: otherwise (ku[j], j) become an aligned 64-bit pair for the swaps below
                    int cj = j;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (cj >= 0 && (bj[q] < 0 || cu > bu[q])) {
                            const unsigned su = bu[q]; const int sj = bj[q];
                            bu[q] = cu; bj[q] = cj; cu = su; cj = sj;
                        }
                    }
                }
            }
        }
        unsigned el[4];
        bool ev[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { ev[q] = bj[q] >= 0; el[q] = 0xFFFFFFFFu - (unsigned)(bj[q] * TOPK_THREADS + t); }
        const unsigned long long bound = radix_kth<4>(R, bu, el, ev, kk);      // at least kk elements of the row are >= bound
        // ---- level 2: compact everything >= bound, select among the survivors ----
        if (t == 0) s_count = 0;
        __syncthreads();
        {
            int tt = t, nn = n;
            OMNI_OPAQUE_V(tt);
            OMNI_OPAQUE_S(nn);
            const unsigned bhi = (unsigned)(bound >> 32), blo = (unsigned)bound;
#pragma unroll
            for (int j = 0; j < NPT; ++j) {
                if (j * TOPK_THREADS >= nn) break;
                const int i = j * TOPK_THREADS + tt;
                if (i < nn && key_ge(ku[j], i, bhi, blo)) {
                    const int slot = atomicAdd(&s_count, 1);
                    if (slot < TOPK_PREFILTER_CAND) store_key(sel, slot, ku[j], i);
                }
            }
        }
        __syncthreads();
        const int ncand = s_count;
        if (ncand <= TOPK_PREFILTER_CAND) {              // (otherwise: the full selection below)
            unsigned cu[4], cl[4];
            bool cv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = q * TOPK_THREADS + t;
                cv[q] = idx < ncand;
                const unsigned long long key = cv[q] ? sel[idx] : 0ull;
                cu[q] = (unsigned)(key >> 32);
                cl[q] = (unsigned)key;
            }
            __syncthreads();                             // every candidate is in registers: sel can be rewritten
            if (ncand > kk) {
                const unsigned long long thr = radix_kth<4>(R, cu, cl, cv, kk);
                if (t == 0) s_count = 0;
                __syncthreads();
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned long long key = ((unsigned long long)cu[q] << 32) | cl[q];
                    if (cv[q] && key >= thr) sel[atomicAdd(&s_count, 1)] = key;
                }
            }
            for (int i = kk + t; i < kpad; i += TOPK_THREADS) sel[i] = 0ull;
            selected = true;
        }
    }

    if (!selected) {
        unsigned long long thr = 0ull;  // keys >= thr are selected
        if (kk < n) {
            if (t == 0) { R.prefix = 0ull; R.need = kk; }
            __syncthreads();
            for (int pass = 7; pass >= 0; --pass) {
                if (t < 256) R.hist[t] = 0u;
                __syncthreads();
                const unsigned long long prefix = R.prefix;
                const int shift = pass * 8;
                if (NPT > 0 && !PREF) {
                    // 32-bit arithmetic on the cached score word / the index word (a 64-bit key per element would double the
                    // registers), and the element index and the trip count are re-derived from opaque copies in every pass:
                    // otherwise the 64 index words, range predicates and loop-exit conditions are hoisted out of the pass loop
                    // as loop invariants and spill.
                    int tt = t, nn = n;
                    OMNI_OPAQUE_V(tt);
                    OMNI_OPAQUE_S(nn);
                    const unsigned phi = (unsigned)(prefix >> 32), plo = (unsigned)prefix;
                    const bool hi = pass >= 4;
                    const int sh = hi ? shift - 32 : shift;                    // shift inside the word this pass looks at
#pragma unroll
                    for (int j = 0; j < NPT; ++j) {
                        if (j * TOPK_THREADS >= nn) break;
                        const int i = j * TOPK_THREADS + tt;
                        const unsigned w = hi ? ku[j] : (0xFFFFFFFFu - (unsigned)i), pw = hi ? phi : plo;
                        // candidates: all bytes above this pass's byte equal the prefix
                        bool todo = i < nn && (hi || ku[j] == phi);
                        if (sh < 24) todo = todo && ((w >> (sh + 8)) == (pw >> (sh + 8)));
                        count_digit(R.hist, todo, (w >> sh) & 255u);
                    }
                } else {
                    // uniform trip count: the wave-level aggregation uses ballots / shuffles
                    for (int base = 0; base < n; base += TOPK_THREADS) {
                        const int i = base + t;
                        bool todo = false;
                        unsigned digit = 0u;
                        if (i < n) {
                            const unsigned long long key = make_key(row[(long)i * estride], i);
                            todo = (pass == 7) || ((key >> (shift + 8)) == (prefix >> (shift + 8)));
                            digit = (unsigned)(key >> shift) & 255u;
                        }
                        count_digit(R.hist, todo, digit);
                    }
                }
                __syncthreads();
                radix_pick_bin(R, shift);
                __syncthreads();
                if (R.done) break;
            }
            thr = R.prefix;  // keys are unique, so exactly kk keys are >= thr
        }
        if (t == 0) s_count = 0;
        for (int i = t; i < kpad; i += TOPK_THREADS) sel[i] = 0ull;
        __syncthreads();
        if (NPT > 0 && !PREF) {
            int tt = t;
            OMNI_OPAQUE_V(tt);
            const unsigned thi = (unsigned)(thr >> 32), tlo = (unsigned)thr;
#pragma unroll
            for (int j = 0; j < NPT; ++j) {
                if (j * TOPK_THREADS >= n) break;
                const int i = j * TOPK_THREADS + tt;
                if (i < n && key_ge(ku[j], i, thi, tlo)) {
                    const int slot = atomicAdd(&s_count, 1);
                    if (slot < TOPK_MAXK) store_key(sel, slot, ku[j], i);
                }
            }
        } else {
            for (int i = t; i < n; i += TOPK_THREADS) {
                const unsigned long long key = make_key(row[(long)i * estride], i);
                if (key >= thr) {
                    const int slot = atomicAdd(&s_count, 1);
                    if (slot < TOPK_MAXK) sel[slot] = key;
                }
            }
        }
    }
    __syncthreads();
    // bitonic sort, descending, kpad a power of two
    for (int size = 2; size <= kpad; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = t; i < (kpad >> 1); i += TOPK_THREADS) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long a = sel[lo], b = sel[hi];
                if ((a < b) == desc) { sel[lo] = b; sel[hi] = a; }
            }
            __syncthreads();
        }
    }
    for (int i = t; i < k; i += TOPK_THREADS) {
        const long o = (long)blockIdx.x * k + i;
        if (i < kk) { out_val[o] = key_value(sel[i]); out_idx[o] = key_index(sel[i]); }
        else { out_val[o] = -INFINITY; out_idx[o] = -1; }
    }
}

// ---- NMS ---------------------------------------------------------------------------------------
// problem q: boxes[q*nmax .. q*nmax + count[q]) sorted by descending score; valid[] == 0 marks boxes
// that take no part (empty / non-finite).  mask: (Q, nmax, words) 64-bit, words = ceil(nmax/64).
// One 4-wave workgroup = 4 row blocks x one column block (the 64 column boxes are staged in LDS once for the four); round 3:
// one wave per workgroup before, 20 000 tiny workgroups for the 20 problems of a training step.
__global__ void __launch_bounds__(256) nms_mask_kernel(const float* __restrict__ boxes, const int* __restrict__ counts,
                                                       int nmax, int words, float thr,
                                                       unsigned long long* __restrict__ mask) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.z, rb = blockIdx.y * 4 + wave, cb = blockIdx.x;
    const int n = counts ? counts[q] : nmax;
    if ((int)blockIdx.y * 4 > cb || cb * 64 >= n) return;            // (uniform) nothing above the diagonal / past the count here
    __shared__ float cbx[64 * 4];
    const float* B = boxes + (long)q * nmax * 4;
    if (wave == 0) {
        const int j = cb * 64 + lane;
        if (j < n) {
            cbx[lane * 4 + 0] = B[j * 4 + 0]; cbx[lane * 4 + 1] = B[j * 4 + 1];
            cbx[lane * 4 + 2] = B[j * 4 + 2]; cbx[lane * 4 + 3] = B[j * 4 + 3];
        }
    }
    __syncthreads();
    if (cb < rb || rb * 64 >= n) return;
    const int i = rb * 64 + lane;
    if (i >= n) return;
    const float x1 = B[i * 4 + 0], y1 = B[i * 4 + 1], x2 = B[i * 4 + 2], y2 = B[i * 4 + 3];
    const float ai = (x2 - x1) * (y2 - y1);
    unsigned long long bits = 0ull;
    const int jmax = (n - cb * 64) < 64 ? (n - cb * 64) : 64;
    for (int jj = (rb == cb ? lane + 1 : 0); jj < jmax; ++jj) {
        const float u1 = cbx[jj * 4 + 0], v1 = cbx[jj * 4 + 1], u2 = cbx[jj * 4 + 2], v2 = cbx[jj * 4 + 3];
        const float w = fmaxf(fminf(x2, u2) - fmaxf(x1, u1), 0.f);
        const float h = fmaxf(fminf(y2, v2) - fmaxf(y1, v1), 0.f);
        const float inter = w * h;
        const float aj = (u2 - u1) * (v2 - v1);
        const float iou = inter / (ai + aj - inter);
        if (iou > thr) bits |= (1ull << jj);
    }
    mask[((long)q * nmax + i) * words + cb] = bits;
}

template <int NMS_SCAN_WAVES>
__global__ void __launch_bounds__(64 * NMS_SCAN_WAVES) nms_scan_kernel(const unsigned long long* __restrict__ mask,
                                                                       const int* __restrict__ counts, const int* __restrict__ valid,
                                                                       int nmax, int words, int* __restrict__ keep) {
    // One workgroup of NMS_SCAN_WAVES waves per problem.  Per 64-box chunk: wave 0 resolves the chunk greedily (64 shuffle steps),
    // then every wave ORs the mask rows of the kept boxes among ITS 64 / NMS_SCAN_WAVES rows of the chunk into a running per-wave
    // `removed` set that lives in registers (lane l holds words l and 64 + l: <= 128 words = 8192 boxes).  The only word anybody
    // needs next is word c + 1, so each wave publishes that one word and wave 0 ORs them -- two barriers per chunk.
    // Round 3: (1) the mask rows of a chunk are fetched BEFORE its resolution decides which are needed (all 64, selected by the
    // `alive` bits afterwards): the loads of chunk c + 1 fly under the work of chunk c instead of sitting between two barriers;
    // (2) 2 barriers per chunk instead of 3 and no LDS pass over the partial sets.  Measured for the 20 x 2000-box problems of a
    // training step: 138 us before, 119 us with (1), 181 us with (1) + (2) on 4 waves (16 rows = 32 loads per lane and chunk queue up
    // behind each other), 16 waves: see profiles/README.md.  Only entries the mask kernel wrote are ever selected (row < n, word >
    // chunk, word * 64 < n), so the scratch needs no zeroing.
    constexpr int RPW = 64 / NMS_SCAN_WAVES;            // rows of a chunk per wave
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = counts ? counts[q] : nmax;
    __shared__ unsigned long long s_alive;
    __shared__ unsigned long long s_next[NMS_SCAN_WAVES];
    for (int i = n + tid; i < nmax; i += 64 * NMS_SCAN_WAVES) keep[(long)q * nmax + i] = 0;      // boxes beyond the count are never kept
    const int nchunks = (n + 63) / 64;
    unsigned long long rw0[RPW], rw1[RPW], diag = 0ull, acc0 = 0ull, acc1 = 0ull, removed_c = 0ull;
    int okv = 0;
    auto fetch = [&](int c) {
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int row = c * 64 + wave * RPW + rr;
            const long rowbase = ((long)q * nmax + row) * words;
            const int w0 = lane, w1 = 64 + lane;
            rw0[rr] = (row < n && w0 > c && w0 < words && w0 * 64 < n) ? mask[rowbase + w0] : 0ull;
            rw1[rr] = (words > 64 && row < n && w1 > c && w1 < words && w1 * 64 < n) ? mask[rowbase + w1] : 0ull;
        }
        if (wave == 0) {
            const int i = c * 64 + lane;
            diag = i < n ? mask[((long)q * nmax + i) * words + c] : 0ull;
            okv = (i < n && (valid == nullptr || valid[(long)q * nmax + i] != 0)) ? 1 : 0;
        }
    };
    if (nchunks > 0) fetch(0);
    for (int c = 0; c < nchunks; ++c) {
        if (wave == 0) {
            const int i = c * 64 + lane;
            // start state of this chunk: not removed by earlier keeps, and a valid box
            unsigned long long alive = __ballot(okv != 0) & ~removed_c;
            for (int r = 0; r < 64; ++r) {
                const unsigned long long d = __shfl(diag, r, 64);
                if ((alive >> r) & 1ull) alive &= ~d;
            }
            if (i < n) keep[(long)q * nmax + i] = (int)((alive >> lane) & 1ull);
            if (lane == 0) s_alive = alive;
        }
        __syncthreads();
        const unsigned long long alive = s_alive;
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            if ((alive >> (wave * RPW + rr)) & 1ull) { acc0 |= rw0[rr]; acc1 |= rw1[rr]; }
        }
        // word c + 1 of this wave's removed set sits in lane (c + 1) % 64 of acc0 (c + 1 < 64) or acc1
        const int nw = c + 1;
        const unsigned long long mine = __shfl(nw < 64 ? acc0 : acc1, nw & 63, 64);
        if (lane == 0) s_next[wave] = mine;
        if (c + 1 < nchunks) fetch(c + 1);
        __syncthreads();
        if (wave == 0) {
            removed_c = 0ull;
#pragma unroll
            for (int k = 0; k < NMS_SCAN_WAVES; ++k) removed_c |= s_next[k];
        }
    }
}

inline void launch_topk(int blocks, int nmax, const float* keys, const SegP& seg, long pitch, int estride, int k, float* out_val,
                        int* out_idx, hipStream_t st) {
    if (nmax <= TOPK_NPT * TOPK_THREADS && k <= TOPK_PREFILTER_K)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(topk_rows_kernel<TOPK_NPT, true>), dim3(blocks), dim3(TOPK_THREADS), 0, st, keys, seg, pitch, estride,
                           k, out_val, out_idx);
    else if (nmax <= TOPK_NPT * TOPK_THREADS)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(topk_rows_kernel<TOPK_NPT, false>), dim3(blocks), dim3(TOPK_THREADS), 0, st, keys, seg, pitch, estride,
                           k, out_val, out_idx);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(topk_rows_kernel<0, false>), dim3(blocks), dim3(TOPK_THREADS), 0, st, keys, seg, pitch, estride, k,
                           out_val, out_idx);
}

}  // namespace

extern "C" {

// Sorted (descending, ties -> lower index) top-k of each row.  keys element (r, i) lives at
// keys[r*pitch + i*estride].  out_val/out_idx are (rows, k); slots beyond min(k, n) hold -inf / -1.
int omni_topk_rows(const float* keys, int rows, int n, long long pitch, int estride, int k, float* out_val, int* out_idx,
                   void* stream) {
    if (rows < 0 || n < 0 || k <= 0 || k > TOPK_MAXK || estride <= 0) return OMNI_ERR_ARG;
    if (rows == 0) return OMNI_OK;
    SegP seg;
    seg.S = 1;
    for (int i = 0; i < TOPK_MAXSEG; ++i) { seg.off[i] = 0; seg.n[i] = n; }
    launch_topk(rows, n, keys, seg, (long)pitch, estride, k, out_val, out_idx, (hipStream_t)stream);
    return omni_launch_status();
}

// Same, for `nseg` (<= 8) column segments of every row in ONE launch (the per-FPN-level pre-NMS top-k of
// find_top_rpn_proposals): segment s covers elements [seg_off[s], seg_off[s] + seg_n[s]) (host int arrays).
// out_val / out_idx: (rows, nseg, k); indices are relative to the segment start.
int omni_topk_segments(const float* keys, int rows, long long pitch, int estride, int nseg, const int* seg_off,
                       const int* seg_n, int k, float* out_val, int* out_idx, void* stream) {
    if (rows < 0 || nseg <= 0 || nseg > TOPK_MAXSEG || k <= 0 || k > TOPK_MAXK || estride <= 0) return OMNI_ERR_ARG;
    if (rows == 0) return OMNI_OK;
    SegP seg;
    seg.S = nseg;
    for (int i = 0; i < TOPK_MAXSEG; ++i) {
        seg.off[i] = i < nseg ? seg_off[i] : 0;
        seg.n[i] = i < nseg ? seg_n[i] : 0;
        if (seg.off[i] < 0 || seg.n[i] < 0) return OMNI_ERR_ARG;
    }
    int nmax = 0;
    for (int i = 0; i < nseg; ++i) nmax = seg.n[i] > nmax ? seg.n[i] : nmax;
    launch_topk(rows * nseg, nmax, keys, seg, (long)pitch, estride, k, out_val, out_idx, (hipStream_t)stream);
    return omni_launch_status();
}

// Greedy NMS for Q independent problems of up to nmax score-sorted boxes each ((Q, nmax, 4) XYXY).
// counts [nullable] (Q) active boxes per problem; valid [nullable] (Q, nmax) 0 = box takes no part.
// mask_ws: Q * nmax * ceil(nmax/64) 64-bit words of scratch (need not be zeroed: only entries the mask kernel writes are read).
// keep (Q, nmax) int32 out (0/1), every element written.
int omni_nms_sorted(const float* boxes, const int* counts, const int* valid, int Q, int nmax, float iou_thr,
                    unsigned long long* mask_ws, int* keep, void* stream) {
    if (Q < 0 || nmax < 0 || nmax > 8192) return OMNI_ERR_ARG;
    if (Q == 0 || nmax == 0) return OMNI_OK;
    const int words = (nmax + 63) / 64;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(words, (words + 3) / 4, Q), dim3(256), 0, st, boxes, counts, nmax, words, iou_thr,
                       mask_ws);
    static const int scan_waves = [] { const char* e = getenv("OMNI_NMS_SCAN_WAVES"); return e ? atoi(e) : 16; }();   // A/B knob
    if (scan_waves == 4)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(nms_scan_kernel<4>), dim3(Q), dim3(256), 0, st, (const unsigned long long*)mask_ws, counts, valid,
                           nmax, words, keep);
    else if (scan_waves == 8)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(nms_scan_kernel<8>), dim3(Q), dim3(512), 0, st, (const unsigned long long*)mask_ws, counts, valid,
                           nmax, words, keep);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(nms_scan_kernel<16>), dim3(Q), dim3(1024), 0, st, (const unsigned long long*)mask_ws, counts, valid,
                           nmax, words, keep);
    return omni_launch_status();
}

}  // extern "C"
