// conv_gemm.hip -- implicit-GEMM convolution / linear layers on exact-fp32 matrix cores (gfx950).
//
// Replaces the cuDNN / cuBLAS calls the reference reaches through torch.nn.Conv2d / nn.Linear:
//   DLA-34 bottom-up convs      /root/reference/cubercnn/modeling/backbone/dla.py:43-51,159-161,241-245,291-295
//   FPN lateral/output convs    detectron2 FPN, built at dla.py:500-506
//   RPN head convs              detectron2 StandardRPNHead (configs/Base.yaml:49)
//   box-head / cube-head FCs    detectron2 FastRCNNConvFCHead; cube_head.py:70,108-144,156-163
// forward, data-gradient and weight-gradient.
//
// MI355X design
//   * activations are NHWC fp32 (torch channels_last), weights KRSC, so the reduction index
//     (r,s,c) is contiguous for both GEMM operands of the forward pass; a Linear layer is the
//     1x1 case with H=W=1.
//   * v_mfma_f32_32x32x2_f32 (exact fp32 = fmaf chain, 157 TFLOP/s peak): the parity bar is fp32
//     1e-4 against the reference's fp32 CPU path, so no reduced-precision operands.
//   * 256-thread workgroup = 4 waves; block tile BM x BN x 16, each wave owns WM x WN 32x32
//     accumulators (<= 64 VGPRs); operands staged through double-buffered LDS with the next
//     tile's global loads (16 B / lane, coalesced along C) in flight during the MFMA phase;
//     one barrier per k-step.
//   * each lane fetches FOUR consecutive k values with one ds_read_b128 and feeds them to four
//     MFMAs: lane half h supplies k = 8*kc + 4*h + t to MFMA t, identically for A and B, which
//     keeps both fragments a single 16-byte LDS read (row pitch 20 floats = conflict-free).
//   * blockIdx -> tile mapping walks M fastest inside an XCD-sized chunk so the 8 private L2s
//     each see a contiguous band of output rows (weights are tiny and shared).
//   * weight gradient: reduction over output pixels split across grid.z, partial tiles are
//     accumulated with fp32 atomics into a zero-initialised dW.
#include <device_rt.h>
#include "split_reduce.h"

namespace {

constexpr int OMNI_MAX_SRC = 6;
struct ConvP {
    const float* x;   // fwd: input NHWC (pitch ldx) | dgrad: dY NHWC (pitch ldx) | wgrad: input NHWC
    const float* w;   // fwd/dgrad: weights KRSC    | wgrad: dY (pitch ldw)
    const float* bias;  // fwd only, nullable
    float* out;       // fwd: output (pitch ldo) | dgrad: dX (pitch ldo) | wgrad: dW KRSC
    int N, H, W, C;   // input tensor
    int OH, OW, K;    // output tensor
    int R, S, stride, pad;
    int ldx, ldo, ldw;
    int relu;
    int accumulate;   // dgrad: out += result
    int splits;       // fwd/dgrad: reduction split over gridDim.y (atomic epilogue into a zeroed output)
    long xb, wb, ob;  // fwd/wgrad: element strides of x / w / out between the gridDim.z problems of a batched GEMM
    float* stats;     // fwd, nullable: per-m-tile column sums [tiles_m][2][K] of the output (BatchNorm statistics, see omni_conv2d_fwd_stats)
    // deterministic mode (round 4, split_reduce.h): ctr != nullptr asks for run-to-run identical sums -- a split reduction meets in
    // the workspace `ws` and the last-arriving workgroup adds the partial tiles in split order; accumulating epilogues use plain
    // read-modify-write (every output element has exactly one owner per launch) instead of atomics
    float* ws;
    unsigned* ctr;
    // multi-source input (round 5, the DLA Root): nsrc > 0 asks the 1 x 1 / stride 1 forward kernel (PF == 1) to read the input
    // channels [coff[s], coff[s + 1]) from the dense NHWC tensor xs[s] (pitch = its own channel count) -- torch.cat(xs, 1) is never
    // formed.  Every coff[s] is a multiple of the slab depth, so a slab has one source.
    const float* xs[OMNI_MAX_SRC];
    int coff[OMNI_MAX_SRC + 1];
    int nsrc;
};


__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
// 16 bytes through a buffer resource: an offset outside the resource (OMNI_OOB) returns zeros WITHOUT a branch
typedef unsigned omni_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 bufld4(omni_rsrc_t r, int voff) {
#ifdef OMNI_HIPEMU
    float4 v = zero4();
    if ((unsigned)voff + 16u <= r.bytes) memcpy(&v, r.base + (unsigned)voff, 16);
    return v;
#else
    const omni_u4 u = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
    return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
#endif
}


// XCD-aware tile order: consecutive workgroup ids round-robin over the 8 XCDs, so give each XCD
// a contiguous chunk of the (m-fastest) tile sequence.
__device__ __forceinline__ void tile_coords(int tiles_m, int tiles_n, int& tm, int& tn) {
    const int nwg = tiles_m * tiles_n;
    int id = blockIdx.x;
    const int q = nwg / 8, r = nwg % 8;
    const int xcd = id % 8, k = id / 8;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    id = (nwg >= 8) ? swz : id;
    if (tiles_n <= 4) {
        // few column tiles (K <= 512): walk them fastest, so the big row operand (activations) of an m-tile is fetched
        // once and re-hit in this XCD's L2 by the next column tile (PMC: FETCH_SIZE halves, profiles/README.md)
        tn = id % tiles_n;
        tm = id / tiles_n;
    } else {
        tm = id % tiles_m;
        tn = id / tiles_m;
    }
}

// workgroup id -> position in a work sequence such that each XCD (ids round-robin over 8 of them) walks one contiguous chunk
__device__ __forceinline__ int xcd_chunked(int id, int total) {
    if (total < 8) return id;
    const int q = total / 8, r = total % 8, xcd = id % 8, k = id / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// ---- fragment fetch + MFMA over one BK=16 slab -------------------------------------------------
// A_KCONTIG: As[m][BKP] else As[k][LDA] (m contiguous).  B likewise.
template <int WM, int WN, bool A_KCONTIG, bool B_KCONTIG, int LDA, int LDB, int BKX = 16>
__device__ __forceinline__ void mma_slab(const float* __restrict__ As, const float* __restrict__ Bs, int a_row0,
                                         int b_row0, int lane, f32x16 (&acc)[WM][WN]) {
    const int l31 = lane & 31, h = lane >> 5;
    constexpr int BKP = BKX + 4;
#pragma unroll
    for (int kc = 0; kc < BKX / 8; ++kc) {
        float a[WM][4], b[WN][4];
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            if (A_KCONTIG) {
                const float4 v = *reinterpret_cast<const float4*>(As + (a_row0 + 32 * i + l31) * BKP + 8 * kc + 4 * h);
                a[i][0] = v.x; a[i][1] = v.y; a[i][2] = v.z; a[i][3] = v.w;
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) a[i][t] = As[(8 * kc + 4 * h + t) * LDA + a_row0 + 32 * i + l31];
            }
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            if (B_KCONTIG) {
                const float4 v = *reinterpret_cast<const float4*>(Bs + (b_row0 + 32 * j + l31) * BKP + 8 * kc + 4 * h);
                b[j][0] = v.x; b[j][1] = v.y; b[j][2] = v.z; b[j][3] = v.w;
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) b[j][t] = Bs[(8 * kc + 4 * h + t) * LDB + b_row0 + 32 * j + l31];
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = mfma_32x32x2(a[i][t], b[j][t], acc[i][j]);
    }
}

template <int WM, int WN>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[WM][WN]) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

// =================================================================================================
// forward:  out[m, n] = sum_{r,s,c} x[pix(m; r, s), c] * w[n, r, s, c] (+ bias[n]) (ReLU)
//   GEMM M = N*OH*OW, N = K, reduction Kd = R*S*C.  A and B k-contiguous in LDS.
// =================================================================================================
// PF = register-staged prefetch depth: the global loads of slab kt + PF are issued while slab kt is in the MFMAs.  PF = 1 is the
// classic double buffer (one MFMA phase per load in flight).  The small-map problems of this network (Winograd point GEMMs of DLA
// levels 3-5, reduction depth 128-512 = 4-16 slabs, 64x64 tiles so that the launch has >= 2 workgroups per CU) are bound by that
// dependency chain -- a 64x64 tile's MFMA phase is 0.43 us per wave against ~2 us of load latency -- so they run PF = 3: three
// slabs (48 VGPRs) in flight per thread, LDS still double buffered.
template <int BM, int BN, int WAVES_M, int WAVES_N, int BKX = 16, int PF = 1>
__global__ void __launch_bounds__(256) conv_fwd_kernel(ConvP p) {
    constexpr int WM = BM / (32 * WAVES_M), WN = BN / (32 * WAVES_N);
    constexpr int BKP = BKX + 4, KQ = BKX / 4;        // float4 columns per slab row
    constexpr int RPP = 256 / KQ;                      // tile rows covered per pass of the 256 threads
    constexpr int AI = (BM + RPP - 1) / RPP, BI = (BN + RPP - 1) / RPP;
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * BKP];
    p.x += (long)blockIdx.z * p.xb;       // batched GEMM (Winograd): problem blockIdx.z
    p.w += (long)blockIdx.z * p.wb;
    p.out += (long)blockIdx.z * p.ob;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int M = p.N * p.OH * p.OW, Kd = p.R * p.S * p.C;
    int tile_m, tile_n;
    tile_coords((M + BM - 1) / BM, (p.K + BN - 1) / BN, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kq = tid % KQ, lrow = tid / KQ;

    // Operands come through buffer resources (round 5, the design of gemm_nt_pf_kernel): padding taps, rows past M / K and slabs
    // past the reduction become an out-of-range OFFSET, not a branch around the load -- every load_slab() issues the same loads, so
    // the prefetch ring (PF > 1) keeps its slabs in flight instead of draining vmcnt at every guarded load.  Tensors < 2 GiB.
    const omni_rsrc_t rx_ = omni_make_rsrc(p.x, (unsigned)((((long)p.N * p.H * p.W - 1) * p.ldx + p.C) * 4));
    const omni_rsrc_t rw_ = omni_make_rsrc(p.w, (unsigned)((long)p.K * Kd * 4));
    int a_base[AI];                                    // element offset of (img, ih0, iw0, 0) (may be negative: padding)
    int a_ih[AI], a_iw[AI];
    bool a_ok[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int m = m0 + lrow + RPP * i;
        a_ok[i] = (lrow + RPP * i < BM) && m < M;
        const int mm = a_ok[i] ? m : 0;
        const int img = mm / (p.OH * p.OW), rem = mm - img * (p.OH * p.OW);
        const int oh = rem / p.OW, ow = rem - oh * p.OW;
        a_ih[i] = oh * p.stride - p.pad;
        a_iw[i] = ow * p.stride - p.pad;
        a_base[i] = ((img * p.H + a_ih[i]) * p.W + a_iw[i]) * p.ldx;
    }
    // reduction cursor of this thread's float4 column: kd = (r*S + s)*C + c, advanced by BKX per slab
    const int nk_total = (Kd + BKX - 1) / BKX;
    const int sps = (nk_total + (int)gridDim.y - 1) / (int)gridDim.y;       // slabs per split
    const int kt_begin = (int)blockIdx.y * sps;
    const int nk = min(sps, nk_total - kt_begin);
    // Slab order.  Default (r, s, c): kd runs through the KRSC weight row.  When C is a multiple of the slab depth
    // the order is (c-chunk, r, s) instead: the R*S taps of one channel chunk are consecutive slabs, so the
    // shifted re-reads of the same input pixels hit L1/L2 while they are hot (measured: FETCH_SIZE of the
    // 3x3 256->256 @128x128 conv drops, see profiles/README.md).  Any order gives the same sum up to rounding.
    const bool tap_inner = (p.C % BKX) == 0 && p.R * p.S > 1;
    const int RS = p.R * p.S;
    int kd, c_cur, r_cur, s_cur;
    if (tap_inner) {
        const int chunk = kt_begin / RS, tap0 = kt_begin - chunk * RS;
        c_cur = chunk * BKX + kq * 4;
        r_cur = tap0 / p.S;
        s_cur = tap0 - r_cur * p.S;
        kd = tap0 * p.C + c_cur;
    } else {
        kd = kt_begin * BKX + kq * 4;
        const int tap0 = kd / p.C;
        c_cur = kd - tap0 * p.C;
        r_cur = tap0 / p.S;
        s_cur = tap0 - r_cur * p.S;
    }
    float4 ra[PF][AI], rb[PF][BI];
    int issued = 0;                                    // slabs requested so far: the ring asks past this split's last slab
    auto load_slab = [&](const int st) {
        const bool kok = kd < Kd && issued < nk;
        ++issued;
        if constexpr (PF == 1) {
            if (p.nsrc > 0) {                          // (uniform) 1 x 1 convolution over the channel concatenation of p.xs[]
                const int c0 = (kt_begin + issued - 1) * BKX;                  // first channel of this slab: picks its source
                const float* src = p.xs[0];
                int lo = 0, hi = p.coff[1];
#pragma unroll
                for (int q = 1; q < OMNI_MAX_SRC; ++q)
                    if (q < p.nsrc && c0 >= p.coff[q]) { src = p.xs[q]; lo = p.coff[q]; hi = p.coff[q + 1]; }
                const int cs = hi - lo, cl = kd - lo;
#pragma unroll
                for (int i = 0; i < AI; ++i)
                    ra[st][i] = (a_ok[i] && kok) ? ldg4(src + (long)(m0 + lrow + RPP * i) * cs + cl) : zero4();
#pragma unroll
                for (int j = 0; j < BI; ++j) {
                    const int n = n0 + lrow + RPP * j;
                    rb[st][j] = bufld4(rw_, (lrow + RPP * j < BN && n < p.K && kok) ? (n * Kd + kd) * 4 : OMNI_OOB);
                }
                kd += BKX;
                return;
            }
        }
        const int tap_off = (r_cur * p.W + s_cur) * p.ldx + c_cur;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int ih = a_ih[i] + r_cur, iw = a_iw[i] + s_cur;
            const bool ok = a_ok[i] && kok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            ra[st][i] = bufld4(rx_, ok ? (a_base[i] + tap_off) * 4 : OMNI_OOB);
        }
#pragma unroll
        for (int j = 0; j < BI; ++j) {
            const int n = n0 + lrow + RPP * j;
            rb[st][j] = bufld4(rw_, (lrow + RPP * j < BN && n < p.K && kok) ? (n * Kd + kd) * 4 : OMNI_OOB);
        }
        if (tap_inner) {
            kd += p.C;
            if (++s_cur == p.S) {
                s_cur = 0;
                if (++r_cur == p.R) { r_cur = 0; c_cur += BKX; kd = c_cur; }
            }
        } else {
            kd += BKX;
            c_cur += BKX;
            while (c_cur >= p.C) {
                c_cur -= p.C;
                if (++s_cur == p.S) { s_cur = 0; ++r_cur; }
            }
        }
    };
    auto store_slab = [&](int buf, const int st) {
        float* As = smem + buf * (BM + BN) * BKP;
        float* Bs = As + BM * BKP;
#pragma unroll
        for (int i = 0; i < AI; ++i)
            if (lrow + RPP * i < BM) *reinterpret_cast<float4*>(As + (lrow + RPP * i) * BKP + kq * 4) = ra[st][i];
#pragma unroll
        for (int j = 0; j < BI; ++j)
            if (lrow + RPP * j < BN) *reinterpret_cast<float4*>(Bs + (lrow + RPP * j) * BKP + kq * 4) = rb[st][j];
    };

    f32x16 acc[WM][WN];
    zero_acc<WM, WN>(acc);
    const bool det = gridDim.y > 1 && p.ctr != nullptr;
    if (nk <= 0 && !det) return;          // (a deterministic split with no slabs still has to arrive, with zeros)
    if (nk <= 0) {
    } else if constexpr (PF == 1) {
        load_slab(0);
        store_slab(0, 0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) load_slab(0);
            const float* As = smem + buf * (BM + BN) * BKP;
            mma_slab<WM, WN, true, true, 0, 0, BKX>(As, As + BM * BKP, wm * WM * 32, wn * WN * 32, lane, acc);
            if (kt + 1 < nk) store_slab(buf ^ 1, 0);
            __syncthreads();
        }
    } else {
        // stage s holds slab kt when kt % PF == s; the cursor of load_slab() runs PF slabs ahead of the MFMAs
#pragma unroll
        for (int s = 0; s < PF; ++s) load_slab(s);                             // slabs 0 .. PF - 1 (zeros past the split's end)
        store_slab(0, 0);
        load_slab(0);
        omni_barrier_lds();
        for (int kt0 = 0; kt0 < nk; kt0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int kt = kt0 + u;
                if (kt < nk) {                                                 // (uniform; only the last round of a ragged nk)
                    const int buf = kt & 1;
                    store_slab(buf ^ 1, (u + 1) % PF);                         // slab kt + 1: its loads were issued PF phases ago
                    load_slab((u + 1) % PF);                                   // slab kt + 1 + PF, unconditionally
                    const float* As = smem + buf * (BM + BN) * BKP;
                    mma_slab<WM, WN, true, true, 0, 0, BKX>(As, As + BM * BKP, wm * WM * 32, wn * WN * 32, lane, acc);
                    omni_barrier_lds();       // not __syncthreads(): that would drain the loads in flight (vmcnt(0))
                }
            }
        }
    }
    const int l31 = lane & 31, h = lane >> 5;
    if (det) {
        __syncthreads();                  // (the slab loop's last LDS reads are done before split_reduce reuses a shared word)
        if (!omni_split_reduce<WM, WN>(p.ws, p.ctr, (long)blockIdx.z * gridDim.x + blockIdx.x, (int)blockIdx.y, (int)gridDim.y, acc)) return;
    }
    const bool split = gridDim.y > 1 && !det;      // atomic epilogue; a deterministic launch continues as if it had not been split
    if (p.stats != nullptr && !split) {
        // BatchNorm batch statistics from the accumulators (the conv -> BN pairs of the bottom-up, dla.py:46-66): per-channel
        // sum and sum of squares over this tile's rows; [tile_m][2][K] partials, summed over tiles in fp64 by bn_finalize.
        // Saves the statistics pass (one full read of the activation) of every BN layer behind a non-split convolution.
        __syncthreads();
        float* red = smem;                                   // [WAVES_M][2][BN]
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            float sv = 0.f, sq = 0.f;
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + (wm * WM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float v = m < M ? acc[i][j][r] : 0.f;
                    sv += v;
                    sq += v * v;
                }
            sv += __shfl_xor(sv, 32, 64);
            sq += __shfl_xor(sq, 32, 64);
            if (h == 0) {
                red[(wm * 2 + 0) * BN + (wn * WN + j) * 32 + l31] = sv;
                red[(wm * 2 + 1) * BN + (wn * WN + j) * 32 + l31] = sq;
            }
        }
        __syncthreads();
        for (int idx = tid; idx < 2 * BN; idx += 256) {
            const int which = idx / BN, nl = idx - which * BN;
            float t = 0.f;
#pragma unroll
            for (int wq = 0; wq < WAVES_M; ++wq) t += red[(wq * 2 + which) * BN + nl];
            if (n0 + nl < p.K) p.stats[((long)tile_m * 2 + which) * p.K + n0 + nl] = t;
        }
    }
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int n = n0 + (wn * WN + j) * 32 + l31;
            const float bv = (p.bias != nullptr && n < p.K && (blockIdx.y == 0 || det)) ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * WM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < M && n < p.K) {
                    float v = acc[i][j][r] + bv;
                    if (split) {
                        atomicAdd(p.out + (long)m * p.ldo + n, v);   // ReLU (if any) is applied by a follow-up pass
                    } else {
                        if (p.relu) v = fmaxf(v, 0.f);
                        p.out[(long)m * p.ldo + n] = v;
                    }
                }
            }
        }
}

__global__ void __launch_bounds__(256) relu_inplace_kernel(float* __restrict__ y, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<float4*>(y)[i];
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        reinterpret_cast<float4*>(y)[i] = v;
    }
}

// =================================================================================================
// data gradient:  dx[m_in, c] = sum_{r,s,k} dy[pix_out(m_in; r, s), k] * w[k, r, s, c]
//   GEMM N = C, reduction over (tap, k).  A k-contiguous, B n-contiguous in LDS.
//
//   Strided convs are decomposed into stride^2 PARITY CLASSES (blockIdx.z): input pixels with
//   (ih % stride, iw % stride) = (ph, pw) only ever meet the taps r = r0 + stride*jr, r0 = (ph + pad) % stride
//   (same for s), so each class is a dense stride-1 correlation of the class' pixel sub-grid with its
//   Rc x Sc tap subset: oh = ihc + oh_off - jr.  No MFMA work is spent on the structurally-zero taps
//   (a 3x3/s2 dgrad does 9/4 taps per pixel instead of 9).  stride 1 is the single class (0, 0).
// =================================================================================================
// PF >= 2 (late round 6): the operand path of gemm_nt_pf_kernel -- PF slabs of buffer loads in flight per thread, issued
// unconditionally (padding taps, rows past M, slabs past this split's range get the out-of-range offset and come back as zeros), LDS
// handed over behind an lgkmcnt-only barrier; the slab loop runs to a multiple of PF with zero slabs.  Same slab order and MFMA chain:
// bit-identical to PF = 1 (the classic double buffer with guarded loads, kept for tensors a 32-bit byte offset does not reach).
template <int BM, int BN, int WAVES_M, int WAVES_N, int BKX = 32, int PF = 1>
__global__ void __launch_bounds__(256) conv_dgrad_kernel(ConvP p) {
    constexpr int WM = BM / (32 * WAVES_M), WN = BN / (32 * WAVES_N);
    constexpr int BKP = BKX + 4, KQ = BKX / 4, RPP = 256 / KQ;
    constexpr int AI = (BM + RPP - 1) / RPP;
    constexpr int BF4 = BN / 4, BROWS = 256 / BF4, BI = (BKX + BROWS - 1) / BROWS;
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM * BKP + BKX * BN)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int st = p.stride;
    const int ph = (int)blockIdx.z / st, pw = (int)blockIdx.z - ph * st;
    const int Hc = (p.H - ph + st - 1) / st, Wc = (p.W - pw + st - 1) / st;      // class sub-grid
    const int r0 = (ph + p.pad) % st, s0 = (pw + p.pad) % st;
    const int Rc = r0 < p.R ? (p.R - r0 + st - 1) / st : 0, Sc = s0 < p.S ? (p.S - s0 + st - 1) / st : 0;
    const int RSc = Rc * Sc;
    const int oh_off = (ph + p.pad - r0) / st, ow_off = (pw + p.pad - s0) / st;
    const int M = p.N * Hc * Wc, Kd = RSc * p.K, RSC = p.R * p.S * p.C;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (p.C + BN - 1) / BN;
    if ((int)blockIdx.x >= tiles_m * tiles_n) return;      // grid.x is sized for the largest class (0, 0)
    int tile_m, tile_n;
    tile_coords(tiles_m, tiles_n, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kq = tid % KQ, lrow = tid / KQ;
    const int bn4 = tid % BF4, brow = tid / BF4;

    long a_base[AI];
    int a_oh[AI], a_ow[AI];
    bool a_ok[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int m = m0 + lrow + RPP * i;
        a_ok[i] = (lrow + RPP * i < BM) && m < M;
        const int mm = a_ok[i] ? m : 0;
        const int img = mm / (Hc * Wc), rem = mm - img * (Hc * Wc);
        const int ihc = rem / Wc, iwc = rem - ihc * Wc;
        a_oh[i] = ihc + oh_off;
        a_ow[i] = iwc + ow_off;
        a_base[i] = ((long)(img * p.OH + a_oh[i]) * p.OW + a_ow[i]) * p.ldx;
    }
    const int nk_total = (Kd + BKX - 1) / BKX;
    const int sps = (nk_total + (int)gridDim.y - 1) / (int)gridDim.y;
    const int kt_begin = (int)blockIdx.y * sps;
    const int nk = min(sps, nk_total - kt_begin);
    // Slab order: (k-chunk, jr, js) when K is a multiple of the slab depth (same L2-reuse argument as the forward
    // kernel: the taps of one chunk of dy channels are consecutive slabs), else (jr, js, k).
    const bool tap_inner = (p.K % BKX) == 0 && RSc > 1;
    // A cursor (this thread's float4 column)
    int kd, k_cur, jr_cur, js_cur;
    // B cursors (this thread's BI weight rows): w[k][tap][c]
    int kdb[BI], kb[BI], jrb[BI], jsb[BI];
    if (nk > 0) {
        if (tap_inner) {
            const int chunk = kt_begin / RSc, tap0 = kt_begin - chunk * RSc;
            k_cur = chunk * BKX + kq * 4;
            jr_cur = tap0 / Sc;
            js_cur = tap0 - jr_cur * Sc;
            kd = 0;   // always in range in this mode (nk bounds the loop)
#pragma unroll
            for (int j = 0; j < BI; ++j) { kb[j] = chunk * BKX + brow + BROWS * j; jrb[j] = jr_cur; jsb[j] = js_cur; kdb[j] = 0; }
        } else {
            kd = kt_begin * BKX + kq * 4;
            const int tap0 = kd / p.K;
            k_cur = kd - tap0 * p.K;
            jr_cur = tap0 / Sc;
            js_cur = tap0 - jr_cur * Sc;
#pragma unroll
            for (int j = 0; j < BI; ++j) {
                kdb[j] = kt_begin * BKX + brow + BROWS * j;
                const int t = kdb[j] / p.K;
                kb[j] = kdb[j] - t * p.K;
                jrb[j] = t / Sc;
                jsb[j] = t - jrb[j] * Sc;
            }
        }
    }
    const int cb = n0 + bn4 * 4;
    float4 ra[PF][AI], rb[PF][BI];
    const omni_rsrc_t rx_ = omni_make_rsrc(p.x, PF > 1 ? (unsigned)((long)p.N * p.OH * p.OW * p.ldx * 4) : 0u);
    const omni_rsrc_t rw_ = omni_make_rsrc(p.w, PF > 1 ? (unsigned)((long)p.K * RSC * 4) : 0u);
    int ld_cnt = 0;                                 // slabs fetched so far (PF > 1: a slab past this split's range loads zeros)
    auto load_slab = [&](const int sg) {
        const bool in_range = PF == 1 || ld_cnt < nk;       // (a slab past this split's range: zeros for both operands)
        const bool kok = kd < Kd && in_range;
        ++ld_cnt;
        const long tap_off = -((long)jr_cur * p.OW + js_cur) * p.ldx + k_cur;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int oh = a_oh[i] - jr_cur, ow = a_ow[i] - js_cur;
            const bool ok = a_ok[i] && kok && (unsigned)oh < (unsigned)p.OH && (unsigned)ow < (unsigned)p.OW;
            if (PF > 1) ra[sg][i] = bufld4(rx_, ok ? (int)((a_base[i] + tap_off) * 4) : OMNI_OOB);
            else ra[sg][i] = ok ? ldg4(p.x + a_base[i] + tap_off) : zero4();
        }
#pragma unroll
        for (int j = 0; j < BI; ++j) {
            const bool ok = (brow + BROWS * j < BKX) && kdb[j] < Kd && cb < p.C && in_range;
            const int tap = (r0 + st * jrb[j]) * p.S + s0 + st * jsb[j];
            if (PF > 1) rb[sg][j] = bufld4(rw_, ok ? (int)(((long)kb[j] * RSC + (long)tap * p.C + cb) * 4) : OMNI_OOB);
            else rb[sg][j] = ok ? ldg4(p.w + (long)kb[j] * RSC + (long)tap * p.C + cb) : zero4();
            if (tap_inner) {
                if (++jsb[j] == Sc) {
                    jsb[j] = 0;
                    if (++jrb[j] == Rc) { jrb[j] = 0; kb[j] += BKX; }
                }
            } else {
                kdb[j] += BKX;
                kb[j] += BKX;
                while (kb[j] >= p.K) {
                    kb[j] -= p.K;
                    if (++jsb[j] == Sc) { jsb[j] = 0; ++jrb[j]; }
                }
            }
        }
        if (tap_inner) {
            if (++js_cur == Sc) {
                js_cur = 0;
                if (++jr_cur == Rc) { jr_cur = 0; k_cur += BKX; }
            }
        } else {
            kd += BKX;
            k_cur += BKX;
            while (k_cur >= p.K) {
                k_cur -= p.K;
                if (++js_cur == Sc) { js_cur = 0; ++jr_cur; }
            }
        }
    };
    auto store_slab = [&](int buf, const int sg) {
        float* As = smem + buf * (BM * BKP + BKX * BN);
        float* Bs = As + BM * BKP;
#pragma unroll
        for (int i = 0; i < AI; ++i)
            if (lrow + RPP * i < BM) *reinterpret_cast<float4*>(As + (lrow + RPP * i) * BKP + kq * 4) = ra[sg][i];
#pragma unroll
        for (int j = 0; j < BI; ++j)
            if (brow + BROWS * j < BKX) *reinterpret_cast<float4*>(Bs + (brow + BROWS * j) * BN + bn4 * 4) = rb[sg][j];
    };

    f32x16 acc[WM][WN];
    zero_acc<WM, WN>(acc);
    const bool det = gridDim.y > 1 && p.ctr != nullptr;
    const bool split = gridDim.y > 1 && !det;
    if (nk <= 0) {
        // no taps for this split / class (e.g. the odd pixels of a 1x1/s2 conv): the gradient is zero there.
        // A split launch pre-zeroes dx and an accumulating one adds nothing; otherwise fall through and store zeros
        // (a deterministic split arrives with its zeros: the last arrival writes the tile).
        if (split || (p.accumulate && !det)) return;
    } else if (PF > 1) {
#pragma unroll
        for (int sg = 0; sg < PF; ++sg) load_slab(sg);      // slabs 0 .. PF-1
        store_slab(0, 0);
        load_slab(0);                                       // slab PF
        omni_barrier_lds();
        for (int kt0 = 0; kt0 < nk; kt0 += PF) {            // (the last round may run into zero slabs)
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int buf = (kt0 + u) & 1;
                store_slab(buf ^ 1, (u + 1) % PF);
                load_slab((u + 1) % PF);
                const float* As = smem + buf * (BM * BKP + BKX * BN);
                mma_slab<WM, WN, true, false, 0, BN, BKX>(As, As + BM * BKP, wm * WM * 32, wn * WN * 32, lane, acc);
                omni_barrier_lds();
            }
        }
    } else {
        load_slab(0);
        store_slab(0, 0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) load_slab(0);
            const float* As = smem + buf * (BM * BKP + BKX * BN);
            mma_slab<WM, WN, true, false, 0, BN, BKX>(As, As + BM * BKP, wm * WM * 32, wn * WN * 32, lane, acc);
            if (kt + 1 < nk) store_slab(buf ^ 1, 0);
            __syncthreads();
        }
    }
    const int l31 = lane & 31, h = lane >> 5;
    if (det) {
        __syncthreads();
        if (!omni_split_reduce<WM, WN>(p.ws, p.ctr, (long)blockIdx.z * gridDim.x + blockIdx.x, (int)blockIdx.y, (int)gridDim.y, acc)) return;
    }
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (wm * WM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m >= M) continue;
            long row = m;
            if (st > 1) {
                const int img = m / (Hc * Wc), rem = m - img * (Hc * Wc);
                const int ihc = rem / Wc, iwc = rem - ihc * Wc;
                row = (long)(img * p.H + ihc * st + ph) * p.W + iwc * st + pw;
            }
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int n = n0 + (wn * WN + j) * 32 + l31;
                if (n < p.C) {
                    float* o = p.out + row * p.ldo + n;
                    if (split) atomicAdd(o, acc[i][j][r]);
                    else *o = p.accumulate ? (*o + acc[i][j][r]) : acc[i][j][r];
                }
            }
        }
}

// =================================================================================================
// weight gradient:  dw[k, r, s, c] = sum_{pix} dy[pix, k] * x[pix_in(pix; r, s), c]
//   GEMM M = K, N = R*S*C, reduction over P = N*OH*OW output pixels, split over grid.y.
//   A (dy) and B (x) both have the reduction index as the slow dimension: LDS tiles [pix][m|n].
//   Each thread owns one (tap, c) float4 column of B and walks its BI pixel rows with incremental
//   (oh, ow, offset) cursors -- no integer division inside the slab loop.
// =================================================================================================
// The body takes its place in the launch as arguments (bx / by / bz of gx / gy workgroups) so that ONE launch can carry several
// problems (conv_wgrad_multi_kernel below: a workgroup looks its problem up and runs this body with that problem's coordinates).
template <int BM, int BN, int WAVES_M, int WAVES_N, int BK>
__device__ __forceinline__ void conv_wgrad_body(ConvP& p, int pix_per_split, int bx, int by, int bz, int gx, int gy, float* smem) {
    constexpr int WM = BM / (32 * WAVES_M), WN = BN / (32 * WAVES_N);
    constexpr int AF4 = BM / 4, AROWS = 256 / AF4, AI = BK / AROWS;
    constexpr int BF4 = BN / 4, BROWS = 256 / BF4, BI = BK / BROWS;
    static_assert(AROWS * AI == BK && BROWS * BI == BK && WAVES_M * WAVES_N == 4, "tile mapping");
    p.x += (long)bz * p.xb;       // batched GEMM (Winograd): problem bz
    p.w += (long)bz * p.wb;
    p.out += (long)bz * p.ob;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int P = p.N * p.OH * p.OW, Nn = p.R * p.S * p.C;
    const int tiles_m = (p.K + BM - 1) / BM;
    // p.relu (unused by a weight gradient) = XCD-contiguous tile order: the tiles_m workgroups that share an x panel run on ONE
    // XCD instead of one on each (fc1: 8 tiles per 2048 x 64 panel, exactly one per XCD in the plain order)
    const int bid = p.relu ? xcd_chunked(bx, gx) : bx;
    const int m0 = (bid % tiles_m) * BM, n0 = (bid / tiles_m) * BN;
    const int p_begin = by * pix_per_split;
    const int p_end = (p_begin + pix_per_split < P) ? p_begin + pix_per_split : P;
    const int am4 = tid % AF4, arow = tid / AF4;
    const int bn4 = tid % BF4, brow = tid / BF4;
    // this thread's B column group is fixed: (tap, c)
    const int nn = n0 + bn4 * 4;
    const bool n_ok = nn < Nn;
    const int tap = n_ok ? nn / p.C : 0;
    const int bc = nn - tap * p.C;
    const int br = tap / p.S - p.pad, bs = tap - (tap / p.S) * p.S - p.pad;   // tap offset minus padding
    const int am = m0 + am4 * 4;
    const bool m_ok = am < p.K;
    // multi-source input (1 x 1 filter over the channel concatenation of p.xs[], see ConvP): this thread's four channels live in ONE
    // of the sources (every width is a multiple of 4), whatever the tile straddles
    const float* xsrc = p.x;
    int xld = p.ldx, bcl = bc;
    if (p.nsrc > 0) {
        int lo = 0, hi = p.coff[1];
        xsrc = p.xs[0];
#pragma unroll
        for (int q = 1; q < OMNI_MAX_SRC; ++q)
            if (q < p.nsrc && nn >= p.coff[q]) { xsrc = p.xs[q]; lo = p.coff[q]; hi = p.coff[q + 1]; }
        xld = hi - lo;
        bcl = nn - lo;
    }

    // pixel cursors (img, oh, ow) of this thread's BI rows of the NEXT slab to load; one slab advances them by
    // BK pixels = d_img images + d_oh rows + d_ow columns (precomputed, carries resolved with two compares)
    int b_pix[BI], b_img[BI], b_oh[BI], b_ow[BI];
#pragma unroll
    for (int j = 0; j < BI; ++j) {
        const int pix = p_begin + brow + BROWS * j;
        const int pp = pix < P ? pix : 0;
        const int img = pp / (p.OH * p.OW), rem = pp - img * (p.OH * p.OW);
        b_pix[j] = pix;
        b_img[j] = img;
        b_oh[j] = rem / p.OW;
        b_ow[j] = rem - b_oh[j] * p.OW;
    }
    const int d_img = BK / (p.OH * p.OW), d_rem = BK - d_img * (p.OH * p.OW);
    const int d_oh = d_rem / p.OW, d_ow = d_rem - d_oh * p.OW;
    const float* a_ptr = p.w + (long)(p_begin + arow) * p.ldw + am;
    int a_pix = p_begin + arow;

    float4 ra[AI], rb[BI];
    auto load_slab = [&]() {
#pragma unroll
        for (int i = 0; i < AI; ++i)
            ra[i] = (m_ok && a_pix + AROWS * i < p_end) ? ldg4(a_ptr + (long)(AROWS * i) * p.ldw) : zero4();
        a_ptr += (long)BK * p.ldw;
        a_pix += BK;
#pragma unroll
        for (int j = 0; j < BI; ++j) {
            const int ih = b_oh[j] * p.stride + br, iw = b_ow[j] * p.stride + bs;
            const bool ok = n_ok && b_pix[j] < p_end && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            rb[j] = ok ? ldg4(xsrc + ((long)(b_img[j] * p.H + ih) * p.W + iw) * xld + bcl) : zero4();
            b_pix[j] += BK;
            b_ow[j] += d_ow;
            if (b_ow[j] >= p.OW) { b_ow[j] -= p.OW; ++b_oh[j]; }
            b_oh[j] += d_oh;
            if (b_oh[j] >= p.OH) { b_oh[j] -= p.OH; ++b_img[j]; }
            b_img[j] += d_img;
        }
    };
    auto store_slab = [&](int buf) {
        float* As = smem + buf * BK * (BM + BN);
        float* Bs = As + BK * BM;
#pragma unroll
        for (int i = 0; i < AI; ++i) *reinterpret_cast<float4*>(As + (arow + AROWS * i) * BM + am4 * 4) = ra[i];
#pragma unroll
        for (int j = 0; j < BI; ++j) *reinterpret_cast<float4*>(Bs + (brow + BROWS * j) * BN + bn4 * 4) = rb[j];
    };

    f32x16 acc[WM][WN];
    zero_acc<WM, WN>(acc);
    const int nk = (p_end - p_begin + BK - 1) / BK;
    const bool det = p.ctr != nullptr && gy > 1;
    if (nk <= 0 && !det) return;
    if (nk > 0) {
        load_slab();
        store_slab(0);
        __syncthreads();
    }
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_slab();
        const float* As = smem + buf * BK * (BM + BN);
        mma_slab<WM, WN, false, false, BM, BN, BK>(As, As + BK * BM, wm * WM * 32, wn * WN * 32, lane, acc);
        if (kt + 1 < nk) store_slab(buf ^ 1);
        __syncthreads();
    }
    const int l31 = lane & 31, h = lane >> 5;
    if (det && !omni_split_reduce<WM, WN>(p.ws, p.ctr, (long)bz * gx + bx, by, gy, acc))
        return;
    const bool single = gy == 1 && !p.accumulate;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int n = n0 + (wn * WN + j) * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * WM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < p.K && n < Nn) {
                    float* o = p.out + (long)m * Nn + n;
                    // (an ordered split arrives here with the complete sum: ONE add per element and launch, whose result does
                    // not depend on any order -- the atomic is only the cheapest way to issue a read-modify-write)
                    if (single || (det && !p.accumulate)) *o = acc[i][j][r];
                    else atomicAdd(o, acc[i][j][r]);
                }
            }
        }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int BK = 32>
__global__ void __launch_bounds__(256) conv_wgrad_kernel(ConvP p, int pix_per_split) {
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (BM + BN)];
    conv_wgrad_body<BM, BN, WAVES_M, WAVES_N, BK>(p, pix_per_split, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x,
                                                  (int)gridDim.y, smem);
}

// Late round 6: the same weight gradient with the operand path of gemm_nt_pf_kernel / gemm_tn_pf_kernel -- PF slabs of buffer loads in
// flight per thread, issued unconditionally (pixels past the split, padding taps and columns past K / R*S*C get the out-of-range
// offset and come back as zeros, so the wait-count insertion keeps vmcnt(N) instead of draining at every guarded load), LDS handed
// over behind an lgkmcnt-only barrier.  conv_wgrad_body above waits for every slab's loads before its MFMAs can start (PMC: MFMA-busy
// 0.25 on the direct weight gradients).  Same slab order, same MFMA chain per output element: bit-identical; the reduction is padded
// to a multiple of PF slabs with zero slabs.  Single-source problems whose tensors stay below 2 GiB (the launcher checks).
template <int BM, int BN, int WAVES_M, int WAVES_N, int BK, int PF>
__device__ __forceinline__ void conv_wgrad_body_pf(ConvP& p, int pix_per_split, int bx, int by, int gx, int gy, float* smem) {
    constexpr int WM = BM / (32 * WAVES_M), WN = BN / (32 * WAVES_N);
    constexpr int AF4 = BM / 4, AROWS = 256 / AF4, AI = BK / AROWS;
    constexpr int BF4 = BN / 4, BROWS = 256 / BF4, BI = BK / BROWS;
    static_assert(AROWS * AI == BK && BROWS * BI == BK && WAVES_M * WAVES_N == 4, "tile mapping");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int P = p.N * p.OH * p.OW, Nn = p.R * p.S * p.C;
    const int tiles_m = (p.K + BM - 1) / BM;
    const int bid = p.relu ? xcd_chunked(bx, gx) : bx;
    const int m0 = (bid % tiles_m) * BM, n0 = (bid / tiles_m) * BN;
    const int p_begin = by * pix_per_split;
    const int p_end = (p_begin + pix_per_split < P) ? p_begin + pix_per_split : P;
    const int am4 = tid % AF4, arow = tid / AF4;
    const int bn4 = tid % BF4, brow = tid / BF4;
    const int nn = n0 + bn4 * 4;
    const bool n_ok = nn < Nn;
    const int tap = n_ok ? nn / p.C : 0;
    const int bc = nn - tap * p.C;
    const int br = tap / p.S - p.pad, bs = tap - (tap / p.S) * p.S - p.pad;
    const int am = m0 + am4 * 4;
    const bool m_ok = am < p.K;
    const omni_rsrc_t ra_ = omni_make_rsrc(p.w, (unsigned)((long)P * p.ldw * 4));
    const omni_rsrc_t rb_ = omni_make_rsrc(p.x, (unsigned)((long)p.N * p.H * p.W * p.ldx * 4));
    int b_pix[BI], b_img[BI], b_oh[BI], b_ow[BI];
#pragma unroll
    for (int j = 0; j < BI; ++j) {
        const int pix = p_begin + brow + BROWS * j;
        const int pp = pix < P ? pix : 0;
        const int img = pp / (p.OH * p.OW), rem = pp - img * (p.OH * p.OW);
        b_pix[j] = pix;
        b_img[j] = img;
        b_oh[j] = rem / p.OW;
        b_ow[j] = rem - b_oh[j] * p.OW;
    }
    const int d_img = BK / (p.OH * p.OW), d_rem = BK - d_img * (p.OH * p.OW);
    const int d_oh = d_rem / p.OW, d_ow = d_rem - d_oh * p.OW;
    int a_off = ((p_begin + arow) * p.ldw + am) * 4;          // byte offset of this thread's first dy row of the NEXT slab
    int a_pix = p_begin + arow;
    const int a_row = AROWS * p.ldw * 4, a_slab = BK * p.ldw * 4;

    float4 ra[PF][AI], rb[PF][BI];
    auto load_slab = [&](const int st) {
#pragma unroll
        for (int i = 0; i < AI; ++i) ra[st][i] = bufld4(ra_, (m_ok && a_pix + AROWS * i < p_end) ? a_off + a_row * i : OMNI_OOB);
        a_off += a_slab;
        a_pix += BK;
#pragma unroll
        for (int j = 0; j < BI; ++j) {
            const int ih = b_oh[j] * p.stride + br, iw = b_ow[j] * p.stride + bs;
            const bool ok = n_ok && b_pix[j] < p_end && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            rb[st][j] = bufld4(rb_, ok ? (((b_img[j] * p.H + ih) * p.W + iw) * p.ldx + bc) * 4 : OMNI_OOB);
            b_pix[j] += BK;
            b_ow[j] += d_ow;
            if (b_ow[j] >= p.OW) { b_ow[j] -= p.OW; ++b_oh[j]; }
            b_oh[j] += d_oh;
            if (b_oh[j] >= p.OH) { b_oh[j] -= p.OH; ++b_img[j]; }
            b_img[j] += d_img;
        }
    };
    auto store_slab = [&](int buf, const int st) {
        float* As = smem + buf * BK * (BM + BN);
        float* Bs = As + BK * BM;
#pragma unroll
        for (int i = 0; i < AI; ++i) *reinterpret_cast<float4*>(As + (arow + AROWS * i) * BM + am4 * 4) = ra[st][i];
#pragma unroll
        for (int j = 0; j < BI; ++j) *reinterpret_cast<float4*>(Bs + (brow + BROWS * j) * BN + bn4 * 4) = rb[st][j];
    };

    f32x16 acc[WM][WN];
    zero_acc<WM, WN>(acc);
    const int nk = (p_end - p_begin + BK - 1) / BK;
    const bool det = p.ctr != nullptr && gy > 1;
    if (nk <= 0 && !det) return;
    if (nk > 0) {
#pragma unroll
        for (int s = 0; s < PF; ++s) load_slab(s);          // slabs 0 .. PF-1 (slabs past the split: zeros)
        store_slab(0, 0);
        load_slab(0);                                       // slab PF
        omni_barrier_lds();
        for (int kt0 = 0; kt0 < nk; kt0 += PF) {            // (the last round may run into zero slabs: they add nothing)
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int buf = (kt0 + u) & 1;
                store_slab(buf ^ 1, (u + 1) % PF);
                load_slab((u + 1) % PF);
                const float* As = smem + buf * BK * (BM + BN);
                mma_slab<WM, WN, false, false, BM, BN, BK>(As, As + BK * BM, wm * WM * 32, wn * WN * 32, lane, acc);
                omni_barrier_lds();
            }
        }
    }
    const int l31 = lane & 31, h = lane >> 5;
    if (det && !omni_split_reduce<WM, WN>(p.ws, p.ctr, (long)bx, by, gy, acc))
        return;
    const bool single = gy == 1 && !p.accumulate;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int n = n0 + (wn * WN + j) * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * WM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < p.K && n < Nn) {
                    float* o = p.out + (long)m * Nn + n;
                    if (single || (det && !p.accumulate)) *o = acc[i][j][r];
                    else atomicAdd(o, acc[i][j][r]);
                }
            }
        }
}

// xcd_splits: every tile of ONE pixel split reads the same dy rows (all (tap, c) tiles) and the same x pixels (all K tiles), but in the
// launch order (tile fastest, workgroup ids round-robin over the 8 XCDs) the tiles of a split land on eight different L2s and each
// fetches that range for itself (PMC, 3x3/s2 64->128: 145 MB fetched for 25 MB of operands).  With the flag the first 8 * (splits / 8)
// splits are dealt out so that XCD q runs splits q, q + 8, ... with all their tiles; the remaining splits keep the plain order.
template <int BM, int BN, int WAVES_M, int WAVES_N, int BK, int PF>
__global__ void __launch_bounds__(256) conv_wgrad_pf_kernel(ConvP p, int pix_per_split, int xcd_splits) {
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (BM + BN)];
    int bx = (int)blockIdx.x, by = (int)blockIdx.y;
    const int gx = (int)gridDim.x, gy = (int)gridDim.y;
    if (xcd_splits) {
        const int lin = by * gx + bx, gy8 = gy & ~7;
        if (lin < gx * gy8) {
            const int q = lin & 7, k = lin >> 3;
            by = q + 8 * (k / gx);
            bx = k - (k / gx) * gx;
        }
    }
    conv_wgrad_body_pf<BM, BN, WAVES_M, WAVES_N, BK, PF>(p, pix_per_split, bx, by, gx, gy, smem);
}

// Round 6: the direct weight gradients of one backward stage in ONE launch (the Winograd-domain ones have had
// gemm_tn_multi_kernel since round 4: 0.23 -> 0.66 of peak inside the step).  The 23 direct launches of a step -- 1 x 1 roots,
// projections and laterals, stride-2 3 x 3 layers: 1-2.4 GFLOP each -- ran at 0.23-0.30 of peak inside the step: every launch ramps up
// and drains on its own while the weight-gradient stream shares the chip with the critical path.  Here problem q owns the workgroups
// [first[q], first[q + 1]) of a 1-D grid, local index l -> (tile l % tiles[q], split l / tiles[q]); arithmetic, split structure,
// workspace slots and counters of every problem are exactly those of its own launch of conv_wgrad_kernel (bit-identical results).
constexpr int WGRAD_MULTI_MAX = 12;      // 12 x sizeof(ConvP) + the index arrays stay below the 4 KB of kernel arguments
struct ConvWMulti {
    ConvP p[WGRAD_MULTI_MAX];
    int pps[WGRAD_MULTI_MAX], tiles[WGRAD_MULTI_MAX], splits[WGRAD_MULTI_MAX], first[WGRAD_MULTI_MAX + 1];
    int n;
};
template <int BM, int BN, int WAVES_M, int WAVES_N, int BK = 32>
__global__ void __launch_bounds__(256) conv_wgrad_multi_kernel(ConvWMulti m) {
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (BM + BN)];
    int q = 0;
#pragma unroll
    for (int i = 1; i < WGRAD_MULTI_MAX; ++i)
        if (i < m.n && (int)blockIdx.x >= m.first[i]) q = i;
    const int l = (int)blockIdx.x - m.first[q];
    ConvP p = m.p[q];
    conv_wgrad_body<BM, BN, WAVES_M, WAVES_N, BK>(p, m.pps[q], l % m.tiles[q], l / m.tiles[q], 0, m.tiles[q], m.splits[q], smem);
}


// =================================================================================================
// Persistent batched GEMM  out[b] (M x N) = A[b] (M x K) * B[b] (N x K)^T   (the 16 Winograd-point GEMMs, K = channels)
//   The reduction is short (K = 256 -> 8 slabs), so with one workgroup per tile the pipeline fill (first global load ->
//   LDS -> first MFMA) and the drain are a sizeable share of a tile's life.  Here 512 resident workgroups (2 per CU) walk
//   the item list (b, tile_m, tile_n; n fastest, one contiguous chunk of items per XCD) and keep the register-staged
//   prefetch running ACROSS items: the first slab of the next item is loaded while the last slab of the current one is
//   in the MFMAs, and the accumulators are stored while the next item's LDS tile is already filled.
// =================================================================================================
struct GemmP {
    const float* A;
    const float* B;
    float* out;
    int batch, M, N, K;
    int tiles_m, tiles_n;
};

__global__ void __launch_bounds__(256) gemm_nt_persistent_kernel(GemmP p) {
    constexpr int BM = 128, BN = 128, BKX = 32, BKP = BKX + 4, KQ = BKX / 4, RPP = 256 / KQ, AI = BM / RPP, BI = BN / RPP;
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * BKP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int kq = tid % KQ, lrow = tid / KQ;
    const int per_batch = p.tiles_m * p.tiles_n;
    const int items = p.batch * per_batch;
    const int per_xcd = (items + 7) / 8;
    const int xcd = (int)blockIdx.x & 7, local = (int)blockIdx.x >> 3, step = (int)gridDim.x >> 3;
    const int end = min((xcd + 1) * per_xcd, items);
    int cur = xcd * per_xcd + local;
    if (cur >= end) return;
    const int nk = p.K / BKX;

    // load cursor (the item / slab the NEXT load_slab() fetches)
    const float* a_ptr[AI];
    const float* b_ptr[BI];
    bool a_ok[AI], b_ok[BI];
    auto point_at = [&](int item) {
        const int b = item / per_batch, r = item - b * per_batch;
        const int m0 = (r / p.tiles_n) * BM, n0 = (r % p.tiles_n) * BN;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int m = m0 + lrow + RPP * i;
            a_ok[i] = m < p.M;
            a_ptr[i] = p.A + ((long)b * p.M + (a_ok[i] ? m : 0)) * p.K + kq * 4;
        }
#pragma unroll
        for (int j = 0; j < BI; ++j) {
            const int n = n0 + lrow + RPP * j;
            b_ok[j] = n < p.N;
            b_ptr[j] = p.B + ((long)b * p.N + (b_ok[j] ? n : 0)) * p.K + kq * 4;
        }
    };
    float4 ra[AI], rb[BI];
    auto load_slab = [&]() {
#pragma unroll
        for (int i = 0; i < AI; ++i) { ra[i] = a_ok[i] ? ldg4(a_ptr[i]) : zero4(); a_ptr[i] += BKX; }
#pragma unroll
        for (int j = 0; j < BI; ++j) { rb[j] = b_ok[j] ? ldg4(b_ptr[j]) : zero4(); b_ptr[j] += BKX; }
    };
    auto store_slab = [&](int buf) {
        float* As = smem + buf * (BM + BN) * BKP;
        float* Bs = As + BM * BKP;
#pragma unroll
        for (int i = 0; i < AI; ++i) *reinterpret_cast<float4*>(As + (lrow + RPP * i) * BKP + kq * 4) = ra[i];
#pragma unroll
        for (int j = 0; j < BI; ++j) *reinterpret_cast<float4*>(Bs + (lrow + RPP * j) * BKP + kq * 4) = rb[j];
    };

    f32x16 acc[2][2];
    zero_acc<2, 2>(acc);
    point_at(cur);
    load_slab();
    store_slab(0);
    __syncthreads();
    int buf = 0;
    const int l31 = lane & 31, h = lane >> 5;
    for (;;) {
        const bool more_items = cur + step < end;
        for (int kt = 0; kt < nk; ++kt) {
            const bool last = kt + 1 == nk;
            const bool has_next = !last || more_items;
            if (last && more_items) point_at(cur + step);
            if (has_next) load_slab();
            const float* As = smem + buf * (BM + BN) * BKP;
            mma_slab<2, 2, true, true, 0, 0, BKX>(As, As + BM * BKP, wm * 64, wn * 64, lane, acc);
            if (has_next) store_slab(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
        {   // epilogue of item `cur`
            const int b = cur / per_batch, r = cur - b * per_batch;
            const int m0 = (r / p.tiles_n) * BM, n0 = (r % p.tiles_n) * BN;
            float* o = p.out + (long)b * p.M * p.N;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int n = n0 + (wn * 2 + j) * 32 + l31;
#pragma unroll
                    for (int rr = 0; rr < 16; ++rr) {
                        const int m = m0 + (wm * 2 + i) * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
                        if (m < p.M && n < p.N) o[(long)m * p.N + n] = acc[i][j][rr];
                        acc[i][j][rr] = 0.f;
                    }
                }
        }
        if (!more_items) break;
        cur += step;
    }
}


// =================================================================================================
// Deep-prefetch tile GEMM (round 3)   out[b] (M x N) = A[b] (M x K) * B[b] (N x K)^T,  K % (32 * PF) == 0
//   The small-map Winograd point GEMMs (DLA levels 3-5: 36 x [256..1024 x 128..512]) have a reduction of only 4-16 slabs and
//   need 64x64 tiles to put >= 2 workgroups on every CU.  With the classic double buffer a wave's MFMA phase per slab is
//   0.43 us against ~2 us of load latency: the launch is a chain of exposed latencies (round 2: 0.34 of peak, MFMA-busy 0.26).
//   This kernel keeps PF slabs of global loads in flight per thread:
//     * loads are `buffer_load_dwordx4` through a buffer resource sized to the problem, so rows past M / N and slabs past K
//       return zeros WITHOUT a branch -- every iteration issues the same number of loads unconditionally, which is what lets
//       the compiler's wait-count insertion use vmcnt(PF-1 slabs) instead of vmcnt(0) (a guarded load makes it drain everything);
//     * the LDS hand-over uses s_barrier with an lgkmcnt-only wait (omni_barrier_lds): __syncthreads() would drain vmcnt too;
//     * LDS stays double buffered (2 x 18 KB), stage registers: PF x 16 VGPRs.
// =================================================================================================
struct GemmTP {
    const float* A;
    const float* B;
    float* out;
    int M, N, K;
    long ab, bb, ob;       // element strides between the gridDim.z problems
    float* ws;             // deterministic split reduction (split_reduce.h); ctr == nullptr: fp32 atomics
    unsigned* ctr;
    int batch = 0;         // number of problems (read by the persistent form only: the others take it from the grid)
    int nt_out = 0;        // 1: `nt` stores for out (outputs far larger than the 32 MB of L2 that the consumer streams from HBM anyway:
                           // tools/probes/probe_gemm_nt.hip -- the p2 point GEMM 176 -> 170 us; smaller outputs stay L2-resident for the consumer)
};

template <int PF>
__global__ void __launch_bounds__(256) gemm_nt_pf_kernel(GemmTP p) {
    constexpr int BM = 64, BN = 64, BKX = 32, BKP = BKX + 4, KQ = BKX / 4, RPP = 256 / KQ, AI = BM / RPP, BI = BN / RPP;
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * BKP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int kq = tid % KQ, lrow = tid / KQ;
    // 1-D grid over (problem, tile) items, problem-major, one contiguous chunk of the sequence per XCD (workgroup ids round-robin
    // over the 8 XCDs): all tiles of a problem -- which re-read the same A / B panels -- run on ONE XCD and share its L2.  With a
    // (tiles, 1, batch) grid the 16 tiles of a 256 x 256 point problem were spread over all eight L2s: 66 MB fetched for 28 MB of
    // operands (profiles/r03_pmc_families.csv).
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN, per_problem = tiles_m * tiles_n;
    const int item = xcd_chunked((int)blockIdx.x, (int)gridDim.x);
    const int prob = item / per_problem, tix = item - prob * per_problem;
    const omni_rsrc_t ra_ = omni_make_rsrc(p.A + (long)prob * p.ab, (unsigned)p.M * (unsigned)p.K * 4u);
    const omni_rsrc_t rb_ = omni_make_rsrc(p.B + (long)prob * p.bb, (unsigned)p.N * (unsigned)p.K * 4u);
    float* out = p.out + (long)prob * p.ob;
    const int tile_m = tix / tiles_n, tile_n = tix - tile_m * tiles_n;      // n fastest: the A panel of a row of tiles stays hot
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    // byte offsets of this thread's float4 column in its AI / BI rows; a row past the end lands past the resource -> zeros
    int a_off[AI], b_off[BI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int m = m0 + lrow + RPP * i;
        a_off[i] = m < p.M ? (m * p.K + kq * 4) * 4 : OMNI_OOB;
    }
#pragma unroll
    for (int j = 0; j < BI; ++j) {
        const int n = n0 + lrow + RPP * j;
        b_off[j] = n < p.N ? (n * p.K + kq * 4) * 4 : OMNI_OOB;
    }
    const int nk = p.K / BKX;                       // multiple of PF (launcher)
    const int k_end = p.K * 4;                      // bytes per row: a slab past it must not run into the next row
    int koff = 0;                                   // byte offset of the slab the NEXT load_slab() fetches
    float4 ra[PF][AI], rb[PF][BI];
    auto load_slab = [&](const int st) {
        const bool kok = koff < k_end;
#pragma unroll
        for (int i = 0; i < AI; ++i) ra[st][i] = bufld4(ra_, kok ? a_off[i] + koff : OMNI_OOB);
#pragma unroll
        for (int j = 0; j < BI; ++j) rb[st][j] = bufld4(rb_, kok ? b_off[j] + koff : OMNI_OOB);
        koff += BKX * 4;
    };
    auto store_slab = [&](int buf, const int st) {
        float* As = smem + buf * (BM + BN) * BKP;
        float* Bs = As + BM * BKP;
#pragma unroll
        for (int i = 0; i < AI; ++i) *reinterpret_cast<float4*>(As + (lrow + RPP * i) * BKP + kq * 4) = ra[st][i];
#pragma unroll
        for (int j = 0; j < BI; ++j) *reinterpret_cast<float4*>(Bs + (lrow + RPP * j) * BKP + kq * 4) = rb[st][j];
    };
    f32x16 acc[1][1];
    zero_acc<1, 1>(acc);
#pragma unroll
    for (int s = 0; s < PF; ++s) load_slab(s);       // slabs 0 .. PF-1
    store_slab(0, 0);
    load_slab(0);                                    // slab PF into the stage just drained
    omni_barrier_lds();
    for (int kt0 = 0; kt0 < nk; kt0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int buf = (kt0 + u) & 1;
            store_slab(buf ^ 1, (u + 1) % PF);       // slab kt + 1 (zeros past the end: written, never read)
            load_slab((u + 1) % PF);                 // slab kt + 1 + PF, unconditionally (see header)
            const float* As = smem + buf * (BM + BN) * BKP;
            mma_slab<1, 1, true, true, 0, 0, BKX>(As, As + BM * BKP, wm * 32, wn * 32, lane, acc);
            omni_barrier_lds();
        }
    }
    const int l31 = lane & 31, h = lane >> 5;
    const int n = n0 + wn * 32 + l31;
    if (p.nt_out) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m < p.M && n < p.N) __builtin_nontemporal_store(acc[0][0][r], out + (long)m * p.N + n);
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < p.M && n < p.N) out[(long)m * p.N + n] = acc[0][0][r];
    }
}

// Persistent form of gemm_nt_pf_kernel for launches of MANY tiles (the 128x128-map point GEMMs: 9216 tiles = 9 rounds of the
// 1024 resident workgroups).  PMC on the one-tile-per-workgroup form showed 3.48 of 4 waves per SIMD resident on average and the MFMA
// pipe idle 28 % of the time: every tile pays a workgroup launch and a prologue whose first slab comes from HBM with nothing to
// overlap it.  Here gridDim.x workgroups (a multiple of 8) each walk their XCD's chunk of the (problem, tile) sequence with a
// stride of gridDim.x / 8, and the slab stream runs THROUGH the tile boundaries: the load cursor (item, koff) is PF slabs ahead
// of the MFMAs, so the first slabs of the next tile are in flight while the last ones of this tile multiply; a tile ends with its
// 16 stores and a zeroed accumulator, nothing else.  Same tiles, same slab order, same MFMA chain per output element as
// gemm_nt_pf_kernel: results are bit-identical.  One resource over ALL problems (offsets carry prob * stride; the launcher checks
// batch * stride * 4 < 2 GiB), so no descriptor changes inside the loop.
template <int PF>
__global__ void __launch_bounds__(256) gemm_nt_pfp_kernel(GemmTP p) {
    constexpr int BM = 64, BN = 64, BKX = 32, BKP = BKX + 4, KQ = BKX / 4, RPP = 256 / KQ, AI = BM / RPP, BI = BN / RPP;
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * BKP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int kq = tid % KQ, lrow = tid / KQ;
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN, per_problem = tiles_m * tiles_n;
    const int total = per_problem * p.batch;
    // this XCD's contiguous chunk of the item sequence (xcd_chunked's split), walked by its gridDim.x / 8 workgroups in rounds
    const int x8 = (int)blockIdx.x & 7, stride = (int)gridDim.x >> 3;
    const int q8 = total / 8, r8 = total % 8;
    const int first = x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8;
    const int last = first + q8 + (x8 < r8 ? 1 : 0);
    int cp_item = first + ((int)blockIdx.x >> 3);
    if (cp_item >= last) return;
    const omni_rsrc_t ra_ = omni_make_rsrc(p.A, (unsigned)((long)p.batch * p.ab * 4));
    const omni_rsrc_t rb_ = omni_make_rsrc(p.B, (unsigned)((long)p.batch * p.bb * 4));
    const int k_end = p.K * 4;                      // bytes per row
    int a_off[AI], b_off[BI];
    int ld_item = cp_item, koff = 0;                // load cursor: the slab the NEXT load_slab() fetches
    auto set_load_item = [&]() {
        const bool ok = ld_item < last;
        const int prob = ld_item / per_problem, tix = ld_item - prob * per_problem;
        const int tile_m = tix / tiles_n, tile_n = tix - tile_m * tiles_n;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int m = tile_m * BM + lrow + RPP * i;
            a_off[i] = (ok && m < p.M) ? (int)(((long)prob * p.ab + (long)m * p.K + kq * 4) * 4) : OMNI_OOB;
        }
#pragma unroll
        for (int j = 0; j < BI; ++j) {
            const int n = tile_n * BN + lrow + RPP * j;
            b_off[j] = (ok && n < p.N) ? (int)(((long)prob * p.bb + (long)n * p.K + kq * 4) * 4) : OMNI_OOB;
        }
    };
    set_load_item();
    float4 ra[PF][AI], rb[PF][BI];
    auto load_slab = [&](const int st) {
#pragma unroll
        for (int i = 0; i < AI; ++i) ra[st][i] = bufld4(ra_, a_off[i] + koff);      // OMNI_OOB + koff stays out of range (koff < 2^30)
#pragma unroll
        for (int j = 0; j < BI; ++j) rb[st][j] = bufld4(rb_, b_off[j] + koff);
        koff += BKX * 4;
    };
    // the reduction depth is a multiple of PF slabs, so a tile's last slab is always fetched at the same place of the unrolled
    // body (u == PF - 2) and after the PF loads of the prologue: the only two places that look for the end of the row
    auto next_tile_if_row_done = [&]() {
        if (koff == k_end) {                        // wave-uniform: on to the first slab of this workgroup's next tile
            koff = 0;
            ld_item += stride;
            set_load_item();
        }
    };
    auto store_slab = [&](int buf, const int st) {
        float* As = smem + buf * (BM + BN) * BKP;
        float* Bs = As + BM * BKP;
#pragma unroll
        for (int i = 0; i < AI; ++i) *reinterpret_cast<float4*>(As + (lrow + RPP * i) * BKP + kq * 4) = ra[st][i];
#pragma unroll
        for (int j = 0; j < BI; ++j) *reinterpret_cast<float4*>(Bs + (lrow + RPP * j) * BKP + kq * 4) = rb[st][j];
    };
    f32x16 acc[1][1];
    zero_acc<1, 1>(acc);
#pragma unroll
    for (int s = 0; s < PF; ++s) load_slab(s);
    next_tile_if_row_done();
    store_slab(0, 0);
    load_slab(0);
    omni_barrier_lds();
    const int nk = p.K / BKX;                       // multiple of PF (launcher): stage and buffer parity carry over the tiles
    const int l31 = lane & 31, h = lane >> 5;
    for (; cp_item < last; cp_item += stride) {
        for (int kt0 = 0; kt0 < nk; kt0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int buf = u & 1;              // nk and PF even
                store_slab(buf ^ 1, (u + 1) % PF);
                load_slab((u + 1) % PF);
                if (u == PF - 2) next_tile_if_row_done();
                const float* As = smem + buf * (BM + BN) * BKP;
                mma_slab<1, 1, true, true, 0, 0, BKX>(As, As + BM * BKP, wm * 32, wn * 32, lane, acc);
                omni_barrier_lds();
            }
        }
        const int prob = cp_item / per_problem, tix = cp_item - prob * per_problem;
        const int tile_m = tix / tiles_n, tile_n = tix - tile_m * tiles_n;
        float* out = p.out + (long)prob * p.ob;
        const int n = tile_n * BN + wn * 32 + l31;
        const int mb = tile_m * BM + wm * 32 + 4 * h;
        if (tile_m * BM + BM <= p.M && tile_n * BN + BN <= p.N) {
            if (p.nt_out) {
#pragma unroll
                for (int r = 0; r < 16; ++r) __builtin_nontemporal_store(acc[0][0][r], out + (long)(mb + (r & 3) + 8 * (r >> 2)) * p.N + n);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) out[(long)(mb + (r & 3) + 8 * (r >> 2)) * p.N + n] = acc[0][0][r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                if (m < p.M && n < p.N) out[(long)m * p.N + n] = acc[0][0][r];
            }
        }
        zero_acc<1, 1>(acc);
    }
}

// TN twin of gemm_nt_pf_kernel: out[b] (Kc x C) (+)= A[b]^T B[b] with A (M x Kc), B (M x C), the reduction running over the M
// rows (Winograd-domain weight gradient dU = dM^T V of the small maps).  gridDim.y splits the rows (`rows_per_split`, a multiple
// of 32 * PF); a split launch adds its partial tile atomically into a zeroed output, a single split stores it.
// `item` = this workgroup's position in the (problem, split, tile) sequence of ONE batched GEMM, tile fastest
template <int PF>
__device__ __forceinline__ void gemm_tn_pf_body(const GemmTP& p, const int rows_per_split, const int splits, const int item) {
    constexpr int BM = 64, BN = 64, BK = 32, F4 = BM / 4, ROWS = 256 / F4, LI = BK / ROWS;      // 16 float4 per row, 16 rows per pass
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (BM + BN)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int c4 = tid % F4, lrow = tid / F4;
    const int tiles_m = (p.N + BM - 1) / BM, tiles = tiles_m * ((p.K + BN - 1) / BN);
    const int prob = item / (tiles * splits), rem = item - prob * (tiles * splits);
    const int split = rem / tiles, tix = rem - split * tiles;
    const omni_rsrc_t ra_ = omni_make_rsrc(p.A + (long)prob * p.ab, (unsigned)p.M * (unsigned)p.N * 4u);   // p.N = Kc (A's width)
    const omni_rsrc_t rb_ = omni_make_rsrc(p.B + (long)prob * p.bb, (unsigned)p.M * (unsigned)p.K * 4u);   // p.K = C  (B's width)
    float* out = p.out + (long)prob * p.ob;
    const int m0 = (tix % tiles_m) * BM, n0 = (tix / tiles_m) * BN;
    const int r_begin = split * rows_per_split;
    const int r_end = min(r_begin + rows_per_split, p.M);
    const int nk = (r_end - r_begin + BK - 1) / BK;
    // this thread's float4 column in A / B (fixed) and its first row; out-of-range columns / rows read zeros through the resource
    const int am = m0 + c4 * 4, bn = n0 + c4 * 4;
    const bool a_ok = am < p.N, b_ok = bn < p.K;
    int row = r_begin + lrow;                       // row of the NEXT load_slab()'s first pass
    float4 ra[PF][LI], rb[PF][LI];
    auto load_slab = [&](const int st) {
#pragma unroll
        for (int i = 0; i < LI; ++i) {
            const int r = row + ROWS * i;
            const bool ok = r < r_end;
            ra[st][i] = bufld4(ra_, (ok && a_ok) ? (r * p.N + am) * 4 : OMNI_OOB);
            rb[st][i] = bufld4(rb_, (ok && b_ok) ? (r * p.K + bn) * 4 : OMNI_OOB);
        }
        row += BK;
    };
    auto store_slab = [&](int buf, const int st) {
        float* As = smem + buf * BK * (BM + BN);
        float* Bs = As + BK * BM;
#pragma unroll
        for (int i = 0; i < LI; ++i) {
            *reinterpret_cast<float4*>(As + (lrow + ROWS * i) * BM + c4 * 4) = ra[st][i];
            *reinterpret_cast<float4*>(Bs + (lrow + ROWS * i) * BN + c4 * 4) = rb[st][i];
        }
    };
    f32x16 acc[1][1];
    zero_acc<1, 1>(acc);
#pragma unroll
    for (int s = 0; s < PF; ++s) load_slab(s);
    store_slab(0, 0);
    load_slab(0);
    omni_barrier_lds();
    for (int kt0 = 0; kt0 < nk; kt0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int buf = (kt0 + u) & 1;
            store_slab(buf ^ 1, (u + 1) % PF);
            load_slab((u + 1) % PF);
            const float* As = smem + buf * BK * (BM + BN);
            mma_slab<1, 1, false, false, BM, BN, BK>(As, As + BK * BM, wm * 32, wn * 32, lane, acc);
            omni_barrier_lds();
        }
    }
    const int l31 = lane & 31, h = lane >> 5;
    const int n = n0 + wn * 32 + l31;
    const bool det = splits > 1 && p.ctr != nullptr;
    if (det && !omni_split_reduce<1, 1>(p.ws, p.ctr, (long)prob * tiles + tix, split, splits, acc)) return;
    const bool single = splits == 1 || det;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < p.N && n < p.K) {
            float* o = out + (long)m * p.K + n;
            if (single) *o = acc[0][0][r];
            else atomicAdd(o, acc[0][0][r]);
        }
    }
}

template <int PF>
__global__ void __launch_bounds__(256) gemm_tn_pf_kernel(GemmTP p, int rows_per_split, int splits) {
    // 1-D grid over (problem, split, tile) items, tile fastest, one contiguous chunk per XCD (see gemm_nt_pf_kernel)
    gemm_tn_pf_body<PF>(p, rows_per_split, splits, xcd_chunked((int)blockIdx.x, (int)gridDim.x));
}

// Several batched TN GEMMs of DIFFERENT shapes in one launch: the Winograd-domain weight gradients of all the layers whose data
// gradients one backward stage has produced (round 4).  They are independent of each other, and each alone is a latency-bound
// launch on 256 CUs -- a level-4 layer is 36 x 16 tiles of 8 slabs, 2.25 workgroups per CU, one wave of work whose duration is a
// workgroup's serial time (prologue, 8 slabs, ordered split sum) -- so a stage's worth of them back to back left the chip mostly waiting.
// In one launch the workgroups of the next problem fill the CUs the moment those of the previous one retire.
//   * ids [first[j], first[j] + roundup8(items[j])) belong to problem j; first[j] is a multiple of 8, so id % 8 -- the XCD the
//     hardware hands the workgroup to -- is also the XCD index INSIDE the problem: each problem's items are cut into eight
//     contiguous chunks, one per XCD (its tiles share the operand panels in that L2), and every XCD gets an eighth of every
//     problem, whatever their relative cost (up to seven padding ids per problem exit at once);
//   * per-problem arithmetic, split structure, workspace slots and counters are those of a launch of gemm_tn_pf_kernel on that
//     problem alone: the results are bit-identical to separate launches.
constexpr int TN_MULTI_MAX = 16;
struct GemmTnMulti {
    GemmTP p[TN_MULTI_MAX];
    int rps[TN_MULTI_MAX], splits[TN_MULTI_MAX], items[TN_MULTI_MAX], first[TN_MULTI_MAX + 1];
    int n;
};
template <int PF>
__global__ void __launch_bounds__(256) gemm_tn_multi_kernel(GemmTnMulti t) {
    const int id = (int)blockIdx.x;
    int j = 0;
    while (j + 1 < t.n && id >= t.first[j + 1]) ++j;
    const int local = id - t.first[j];
    if (local >= t.items[j]) return;
    gemm_tn_pf_body<PF>(t.p[j], t.rps[j], t.splits[j], xcd_chunked(local, t.items[j]));
}

inline bool bad_geom(const ConvP& p) {
    return p.N < 0 || p.H <= 0 || p.W <= 0 || p.C <= 0 || p.K <= 0 || p.R <= 0 || p.S <= 0 || p.stride <= 0 ||
           p.pad < 0 || (p.C & 3) || p.OH != (p.H + 2 * p.pad - p.R) / p.stride + 1 ||
           p.OW != (p.W + 2 * p.pad - p.S) / p.stride + 1;
}

}  // namespace

extern "C" {

// out[N,OH,OW,K] = conv(x[N,H,W,C], w[K,R,S,C]) + bias, optional ReLU.  Pitches in floats.
//
// Tile choice: 128x128 when that already fills the 256 CUs, otherwise 64x64 (4x the workgroups); when even
// that leaves CUs idle and the reduction is deep (DLA level 4/5, FC heads with few rows) the reduction is
// split over gridDim.y with an atomic epilogue into a zeroed output.  Slab depth 32 (one barrier per 64
// MFMAs per wave) measured +23 % over 16 on the 3x3 256->256 @128x128 shape (96 -> 119 TFLOP/s).
// `tile` / `splits` select the algorithm explicitly (0 = the launcher's own choice, which is what omni_conv2d_fwd uses):
//   tile 1 = 128x128, 2 = 64x64, 3 = 128x64, 4 = 256x32 (BM x BN output-pixel x output-channel tile), 5 = 64x64 with three
//   slabs of register prefetch; splits >= 1 = number of
//   reduction splits (atomic epilogue into a zeroed output when > 1; needs ldo == K).  Used by tools/bench_kernels.py for A/B
//   measurements and by tests that want a given tile on a small problem.
constexpr bool WGRAD_XCD_ORDER_FC = true;     // fc-class weight gradients (one split): XCD-contiguous tile order (fc1: 615 -> 562 us, profiles/r03_fc_wgrad_xcd_order.log)
constexpr bool FWD64_DEEP_PREFETCH = true;    // batched GEMMs on 64x64 tiles: gemm_nt_pf_kernel (profiles/r03_sweep_batched_gemm.log: 17-27 % faster on every small-map shape)

// Deterministic-mode plumbing shared by the launchers below (split_reduce.h).  `ctr` != nullptr asks for run-to-run identical
// results; `plan` != nullptr makes the launcher report what it WOULD launch -- plan[0] = tile, [1] = reduction splits, [2] = arrival
// counters (unsigned, zeroed once by the caller; 0 when the launch is not split), [3] = workspace floats -- and return without
// launching, so the caller can size the workspace with the launcher's own decision code.
struct DetArgs {
    float* ws;
    long long ws_floats;
    unsigned* ctr;
    int n_ctr;
    long long* plan;
};
static inline bool det_fits(const DetArgs& d, long tiles, long splits, long tile_elems) {
    return d.ctr != nullptr && d.ws != nullptr && d.n_ctr >= omni_split_counters(tiles, splits) &&
           d.ws_floats >= omni_split_ws_floats(tiles, splits, tile_elems);
}
static inline void det_plan(const DetArgs& d, long tile, long tiles, long splits, long tile_elems) {
    d.plan[0] = tile; d.plan[1] = splits;
    d.plan[2] = splits > 1 ? omni_split_counters(tiles, splits) : 0;
    d.plan[3] = splits > 1 ? omni_split_ws_floats(tiles, splits, tile_elems) : 0;
}

struct MultiSrc {
    const void* const* xs;    // nsrc dense NHWC tensors (N, H, W, cs[s])
    const int* cs;
    int nsrc;
};

static int conv2d_fwd_impl(const float* x, const float* w, const float* bias, float* out, int N, int H, int W, int C, int K,
                           int R, int S, int stride, int pad, int ldx, int ldo, int relu, int tile, int splits_req, float* stats,
                           int stats_rows, int* nblk_out, void* stream, const DetArgs& det = DetArgs{nullptr, 0, nullptr, 0, nullptr},
                           const MultiSrc* ms = nullptr) {
    ConvP p{x, w, bias, out, N, H, W, C, (H + 2 * pad - R) / stride + 1, (W + 2 * pad - S) / stride + 1, K,
            R, S, stride, pad, ldx, ldo, 0, relu, 0, 1};
    p.stats = nullptr;
    p.ws = nullptr;
    p.ctr = nullptr;
    p.nsrc = 0;
    if (nblk_out) *nblk_out = 0;
    if (ms != nullptr) {      // the input is the channel concatenation of ms->xs[]: 1 x 1 / stride 1, every width a multiple of 32
        if (ms->nsrc < 1 || ms->nsrc > OMNI_MAX_SRC || R != 1 || S != 1 || stride != 1 || pad != 0 || tile == 5) return OMNI_ERR_ARG;
        p.coff[0] = 0;
        for (int q = 0; q < OMNI_MAX_SRC; ++q) {
            const bool live = q < ms->nsrc;
            if (live && (ms->xs[q] == nullptr || ms->cs[q] <= 0 || (ms->cs[q] % 32) != 0)) return OMNI_ERR_ARG;
            p.xs[q] = live ? (const float*)ms->xs[q] : nullptr;
            p.coff[q + 1] = p.coff[q] + (live ? ms->cs[q] : 0);
            if (live && (long)N * H * W * ms->cs[q] * 4 >= (1L << 31)) return OMNI_ERR_ARG;
        }
        if (p.coff[ms->nsrc] != C || ldx != C) return OMNI_ERR_ARG;
        p.nsrc = ms->nsrc;
        p.x = p.xs[0];
    }
    if (bad_geom(p) || (ldx & 3) || ldx < C || ldo < K || tile < 0 || tile > 5 || splits_req < 0) return OMNI_ERR_ARG;
    if (splits_req > 1 && ldo != K) return OMNI_ERR_ARG;
    const long M = (long)N * p.OH * p.OW;
    if (M == 0) return OMNI_OK;
    hipStream_t st = (hipStream_t)stream;
    const long Kd = (long)R * S * C;
    // conv_fwd_kernel addresses x and w through buffer resources with 32-bit BYTE offsets (OMNI_OOB = 0x80000000 marks a lane as
    // out of range): an operand of 2 GiB or more would wrap into the resource and read wrong data silently -- refuse it
    if (ms == nullptr && (((long)N * H * W - 1) * ldx + C) * 4 >= (1L << 31)) return OMNI_ERR_ARG;
    if ((long)K * Kd * 4 >= (1L << 31)) return OMNI_ERR_ARG;
    const long nslab = (Kd + 31) / 32;
    const long t128 = ((M + 127) / 128) * ((K + 127) / 128);
    const long t64 = ((M + 63) / 64) * ((K + 63) / 64);
    long splits = 1;
    if (tile == 0) {
        if (K > 64 && t128 >= 256) {
            tile = 1;
        } else if (K > 64 && t128 >= 64 && Kd >= 2048 && ldo == K) {
            // big GEMM with few 128x128 tiles but a deep reduction (fc1: 2048 x 12544 -> 1024): keep the large tile and split K
            tile = 1;
            splits = (512 + t128 - 1) / t128;
            if (splits > nslab / 8) splits = nslab / 8;
        } else if (K > 32 && (K > 64 || ((M + 127) / 128) < 256)) {
            tile = 2;
            if (t64 < 512 && nslab >= 16 && ldo == K) {
                splits = (1024 + t64 - 1) / t64;
                if (splits > nslab / 8) splits = nslab / 8;
                if (splits > 32) splits = 32;
            }
        } else {
            tile = K > 32 ? 3 : 4;
        }
        if (splits < 1) splits = 1;
    }
    if (splits_req >= 1) splits = splits_req;
    if (splits > nslab) splits = nslab;
    const int bm = tile == 1 ? 128 : (tile == 2 || tile == 5) ? 64 : tile == 3 ? 128 : 256;
    const int bn = tile == 1 ? 128 : (tile == 2 || tile == 5) ? 64 : tile == 3 ? 64 : 32;
    const long tiles = ((M + bm - 1) / bm) * ((K + bn - 1) / bn);
    if (det.plan != nullptr) {
        det_plan(det, tile, tiles, splits, (long)bm * bn);
        return OMNI_OK;
    }
    const bool ordered = det.ctr != nullptr && splits > 1;      // split partials meet in the workspace, summed in split order
    if (ordered) {
        if (!det_fits(det, tiles, splits, (long)bm * bn)) return OMNI_ERR_ARG;
        p.ws = det.ws;
        p.ctr = det.ctr;
    }
    if (splits > 1 && !ordered) omni_memset_async(out, 0, sizeof(float) * (size_t)M * K, st);
    if (stats != nullptr && (splits == 1 || ordered) && bias == nullptr && !relu) {      // statistics only from complete, raw outputs
        const long rows = (M + bm - 1) / bm;
        if (rows <= stats_rows) {
            p.stats = stats;
            if (nblk_out) *nblk_out = (int)rows;
        }
    }
#define OMNI_FWD(BM_, BN_, WM_, WN_, BK_, PF_)                                                                           \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_fwd_kernel<BM_, BN_, WM_, WN_, BK_, PF_>),                                     \
                       dim3((unsigned)(((M + BM_ - 1) / BM_) * ((K + BN_ - 1) / BN_)), (unsigned)splits), dim3(256), 0, st, p)
    // (round 5: the 64 x 64 launches of the direct convolutions gain nothing from a deeper ring -- PF 2 / 3 on every one of them:
    // 10.97 / 10.99 ms per step against 10.97, profiles/r05_ab_conv64_pf.log; tile 5 stays an explicit choice)
    if (tile == 1) OMNI_FWD(128, 128, 2, 2, 32, 1);
    else if (tile == 2) OMNI_FWD(64, 64, 2, 2, 32, 1);
    else if (tile == 5) OMNI_FWD(64, 64, 2, 2, 32, 3);
    else if (tile == 3) OMNI_FWD(128, 64, 2, 2, 32, 1);
    else OMNI_FWD(256, 32, 4, 1, 16, 1);   // tiny channel counts (stem, level0/1, RPN 16-wide heads): slab depth 16 measured faster
#undef OMNI_FWD
    if (splits > 1 && relu && !ordered) {     // (an ordered split applies the ReLU in its last-arriving workgroup)
        const long n4 = M * K / 4;
        long g = (n4 + 255) / 256;
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(relu_inplace_kernel, dim3((unsigned)g), dim3(256), 0, st, out, n4);
    }
    return omni_launch_status();
}

int omni_conv2d_fwd_algo(const float* x, const float* w, const float* bias, float* out, int N, int H, int W, int C, int K,
                         int R, int S, int stride, int pad, int ldx, int ldo, int relu, int tile, int splits_req, void* stream) {
    return conv2d_fwd_impl(x, w, bias, out, N, H, W, C, K, R, S, stride, pad, ldx, ldo, relu, tile, splits_req, nullptr, 0, nullptr, stream);
}

// Forward convolution (no bias, no ReLU) that ALSO emits the BatchNorm batch statistics of its output: stats
// [stats_rows][2][K] floats receives per-m-tile partial sums / sums of squares; *nblk_out = number of partial rows written
// (0: the launcher chose a split reduction or the buffer is too small -- the caller then runs the separate statistics pass).
int omni_conv2d_fwd_stats(const float* x, const float* w, float* out, int N, int H, int W, int C, int K, int R, int S, int stride,
                          int pad, int ldx, int ldo, float* stats, int stats_rows, int* nblk_out, void* stream) {
    return conv2d_fwd_impl(x, w, nullptr, out, N, H, W, C, K, R, S, stride, pad, ldx, ldo, 0, 0, 0, stats, stats_rows, nblk_out, stream);
}

// Deterministic form of omni_conv2d_fwd_algo / omni_conv2d_fwd_stats (round 4): a split reduction meets in `ws` (ws_floats floats)
// and is summed in split order by the tile's last-arriving workgroup (ctr: n_ctr zeroed unsigned counters, left zero), bias and ReLU
// applied there; BatchNorm statistics (stats != nullptr) then also come from split launches.  plan != nullptr: report {tile, splits,
// counters, workspace floats} of the launch the arguments describe and return without launching.
int omni_conv2d_fwd_det(const float* x, const float* w, const float* bias, float* out, int N, int H, int W, int C, int K, int R, int S,
                        int stride, int pad, int ldx, int ldo, int relu, int tile, int splits_req, float* stats, int stats_rows,
                        int* nblk_out, float* ws, long long ws_floats, int* ctr, int n_ctr, long long* plan, void* stream) {
    return conv2d_fwd_impl(x, w, bias, out, N, H, W, C, K, R, S, stride, pad, ldx, ldo, relu, tile, splits_req, stats, stats_rows, nblk_out,
                           stream, DetArgs{ws, ws_floats, (unsigned*)ctr, n_ctr, plan});
}

// The same for an input that is the channel concatenation of nsrc <= 6 dense NHWC tensors xs[s] (N, H, W, cs[s]), cs[s] % 32 == 0
// -- the DLA Root's conv1x1(torch.cat(children, 1)) (dla.py:166-172) WITHOUT the concatenated copy: a reduction slab reads its
// channels from the source that holds them.  1 x 1, stride 1, no padding; same tiles, splits, epilogues (bias / ReLU / BatchNorm
// statistics) and bit-identical results as omni_conv2d_fwd_det on the concatenated tensor.  ctr == NULL && plan == NULL: the
// non-deterministic form (a split reduction then meets through atomics in a zeroed output).
int omni_conv2d_fwd_multi_det(const void* const* xs, const int* cs, int nsrc, const float* w, const float* bias, float* out, int N, int H,
                              int W, int K, int ldo, int relu, int tile, int splits_req, float* stats, int stats_rows, int* nblk_out,
                              float* ws, long long ws_floats, int* ctr, int n_ctr, long long* plan, void* stream) {
    if (xs == nullptr || cs == nullptr || nsrc < 1 || nsrc > OMNI_MAX_SRC) return OMNI_ERR_ARG;
    long C = 0;
    for (int q = 0; q < nsrc; ++q) C += cs[q] > 0 ? cs[q] : 0;
    if (C <= 0 || C > (1 << 20)) return OMNI_ERR_ARG;
    const MultiSrc ms{xs, cs, nsrc};
    return conv2d_fwd_impl((const float*)xs[0], w, bias, out, N, H, W, (int)C, K, 1, 1, 1, 0, (int)C, ldo, relu, tile, splits_req, stats, stats_rows,
                           nblk_out, stream, DetArgs{ws, ws_floats, (unsigned*)ctr, n_ctr, plan}, &ms);
}

// Tile choice: 128x128 when that already fills the 256 CUs, otherwise 64x64 (4x the workgroups); when even
// that leaves CUs idle and the reduction is deep (DLA level 4/5, FC heads with few rows) the reduction is
// split over gridDim.y with an atomic epilogue into a zeroed output.  Slab depth 32 (one barrier per 64
// MFMAs per wave) measured +23 % over 16 on the 3x3 256->256 @128x128 shape (96 -> 119 TFLOP/s).
int omni_conv2d_fwd(const float* x, const float* w, const float* bias, float* out, int N, int H, int W, int C, int K,
                    int R, int S, int stride, int pad, int ldx, int ldo, int relu, void* stream) {
    return omni_conv2d_fwd_algo(x, w, bias, out, N, H, W, C, K, R, S, stride, pad, ldx, ldo, relu, 0, 0, stream);
}

// tile: 0 auto | 1 = 128x128 | 2 = 64x64 | 3 = 128x64 | 4 = 256x32; splits: 0 auto | >= 1 explicit (> 1: atomic epilogue; into
// a dx zeroed here when accumulate == 0, which needs lddx == C, or on top of dx's content when accumulate != 0 -- a gradient
// fan-in target, see functional.fanout: neither a zero-fill nor a separate add kernel)
static int conv2d_dgrad_impl(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int K, int R, int S,
                             int stride, int pad, int lddy, int lddx, int accumulate, int tile, int splits_req, void* stream,
                             const DetArgs& det) {
    ConvP p{dy, w, nullptr, dx, N, H, W, C, (H + 2 * pad - R) / stride + 1, (W + 2 * pad - S) / stride + 1, K,
            R, S, stride, pad, lddy, lddx, 0, 0, accumulate, 1};
    p.stats = nullptr;
    p.ws = nullptr;
    p.ctr = nullptr;
    if (bad_geom(p) || (K & 3) || (lddy & 3) || lddy < K || lddx < C || tile < 0 || tile > 4 || splits_req < 0) return OMNI_ERR_ARG;
    if (splits_req > 1 && lddx != C && !accumulate) return OMNI_ERR_ARG;
    if ((long)N * H * W == 0) return OMNI_OK;
    hipStream_t st = (hipStream_t)stream;
    // one launch covers the stride^2 parity classes (grid.z); tiles are sized for the largest class (0, 0)
    const unsigned ncls = (unsigned)(stride * stride);
    const long M = (long)N * ((H + stride - 1) / stride) * ((W + stride - 1) / stride);
    const long Kd = (long)((R + stride - 1) / stride) * ((S + stride - 1) / stride) * K;
    const long nslab = (Kd + 31) / 32;
    const long t128 = ((M + 127) / 128) * ((C + 127) / 128);
    const long t64 = ((M + 63) / 64) * ((C + 63) / 64);
    long splits = 1;
    if (tile == 0) {
        if (C > 64 && t128 * ncls >= 256) {
            tile = 1;
        } else if (C > 32 && (C > 64 || ((M + 127) / 128) * ncls < 256)) {
            tile = 2;
            if (t64 * ncls < 512 && nslab >= 16 && (lddx == C || accumulate)) {
                splits = (1024 + t64 * ncls - 1) / (t64 * ncls);
                if (splits > nslab / 8) splits = nslab / 8;
                if (splits > 32) splits = 32;
                if (splits < 1) splits = 1;
            }
        } else {
            tile = C > 32 ? 3 : 4;
        }
    }
    if (splits_req >= 1) splits = splits_req;
    if (splits > nslab) splits = nslab;
    const int bm = tile == 1 ? 128 : tile == 2 ? 64 : tile == 3 ? 128 : 256, bn = tile == 1 ? 128 : tile == 2 ? 64 : tile == 3 ? 64 : 32;
    const long tiles = ((M + bm - 1) / bm) * ((C + bn - 1) / bn) * ncls;      // counter index: blockIdx.z * gridDim.x + blockIdx.x
    if (det.plan != nullptr) {
        det_plan(det, tile, tiles, splits, (long)bm * bn);
        return OMNI_OK;
    }
    const bool ordered = det.ctr != nullptr && splits > 1;
    if (ordered) {
        if (!det_fits(det, tiles, splits, (long)bm * bn)) return OMNI_ERR_ARG;
        p.ws = det.ws;
        p.ctr = det.ctr;
    }
    if (splits > 1 && !accumulate && !ordered) omni_memset_async(dx, 0, sizeof(float) * (size_t)N * H * W * C, st);
    // deep-prefetch form (PF = 2) when a 32-bit byte offset reaches both operands; OMNI_DGRAD_PF=0: the classic body (A/B knob)
    static const int dgrad_pf = [] { const char* e = getenv("OMNI_DGRAD_PF"); return e ? atoi(e) : 2; }();
    const bool dpf_ok = dgrad_pf >= 2 && (long)N * p.OH * p.OW * lddy * 4 < (1L << 31) - (1L << 24) && (long)K * R * S * C * 4 < (1L << 31) - (1L << 24);
#define OMNI_DGRAD_PF(BM_, BN_, WM_, WN_, BK_)                                                                           \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_dgrad_kernel<BM_, BN_, WM_, WN_, BK_, 2>),                                     \
                       dim3((unsigned)(((M + BM_ - 1) / BM_) * ((C + BN_ - 1) / BN_)), (unsigned)splits, ncls), dim3(256), 0, st, p)
    if (dpf_ok && tile >= 1 && tile <= 3) {
        if (tile == 1) OMNI_DGRAD_PF(128, 128, 2, 2, 32);
        else if (tile == 2) OMNI_DGRAD_PF(64, 64, 2, 2, 32);
        else OMNI_DGRAD_PF(128, 64, 2, 2, 32);
        return omni_launch_status();
    }
#undef OMNI_DGRAD_PF
#define OMNI_DGRAD(BM_, BN_, WM_, WN_, BK_)                                                                              \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_dgrad_kernel<BM_, BN_, WM_, WN_, BK_>),                                        \
                       dim3((unsigned)(((M + BM_ - 1) / BM_) * ((C + BN_ - 1) / BN_)), (unsigned)splits, ncls), dim3(256), 0, st, p)
    if (tile == 1) OMNI_DGRAD(128, 128, 2, 2, 32);
    else if (tile == 2) OMNI_DGRAD(64, 64, 2, 2, 32);
    else if (tile == 3) OMNI_DGRAD(128, 64, 2, 2, 32);
    else OMNI_DGRAD(256, 32, 4, 1, 16);
#undef OMNI_DGRAD
    return omni_launch_status();
}

int omni_conv2d_dgrad_algo(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int K, int R, int S,
                           int stride, int pad, int lddy, int lddx, int accumulate, int tile, int splits_req, void* stream) {
    return conv2d_dgrad_impl(dy, w, dx, N, H, W, C, K, R, S, stride, pad, lddy, lddx, accumulate, tile, splits_req, stream,
                             DetArgs{nullptr, 0, nullptr, 0, nullptr});
}

// deterministic form (see omni_conv2d_fwd_det): an ordered split writes dx (or dx + the carry when accumulate != 0) once, from the
// last-arriving workgroup of each tile; no zero-fill, no atomics
int omni_conv2d_dgrad_det(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int K, int R, int S, int stride, int pad,
                          int lddy, int lddx, int accumulate, int tile, int splits_req, float* ws, long long ws_floats, int* ctr, int n_ctr,
                          long long* plan, void* stream) {
    return conv2d_dgrad_impl(dy, w, dx, N, H, W, C, K, R, S, stride, pad, lddy, lddx, accumulate, tile, splits_req, stream,
                             DetArgs{ws, ws_floats, (unsigned*)ctr, n_ctr, plan});
}

// dx[N,H,W,C] (=|+= when accumulate) = conv_transpose(dy[N,OH,OW,K], w[K,R,S,C]).
int omni_conv2d_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int K, int R, int S,
                      int stride, int pad, int lddy, int lddx, int accumulate, void* stream) {
    return omni_conv2d_dgrad_algo(dy, w, dx, N, H, W, C, K, R, S, stride, pad, lddy, lddx, accumulate, 0, 0, stream);
}

// dw[K,R,S,C] = sum over output pixels of dy (x) x.  accumulate == 0: dw is overwritten (zeroed here when the
// reduction is split); accumulate != 0: the result is atomically ADDED to dw -- this is how weight gradients land
// directly in the flat gradient bucket without an extra add kernel per parameter.
// tile: 0 auto | 1 = 128x128 | 2 = 64x64 | 3 = 128x64 | 4 = 32x128 (BM over K, BN over the (r, s, c) extent)
// tile shape, tile count, reduction split and pixels per split of a weight-gradient launch: a function of the problem alone, shared by
// the single-problem launcher and the multi-problem one (which keeps every problem's structure, hence its bits)
struct WgradGeom {
    int bm, bn, tiles, pps;
    long splits;
};
// fc1-class weight gradients ([1024 x 2048]^T [2048 x 12544]: few "pixels", a very wide (tap, c) extent): 8 x 196 = 1568 tiles of
// 128 x 64 are 2.04 rounds of the 768 resident workgroups -- a third round of 32 workgroups costs ~100 us; 3136 tiles of 64 x 64 at four
// per CU end together: 589 -> 534 us (0.57 -> 0.63 of peak), the cube head's 512-ROI launch 157 -> 142 us (tools/probes/fc_wgrad_tiles.py,
// profiles/r06_fc_wgrad_tiles.log).  Same slab order per output element.  OMNI_WGRAD_FC_TILE64=0: the 128 x 64 tile.
static inline bool wgrad_fc_tile64() {
    static const bool on = [] { const char* e = getenv("OMNI_WGRAD_FC_TILE64"); return e == nullptr || atoi(e) != 0; }();
    return on;
}
static inline WgradGeom wgrad_geom(int K, int Nn, long P, int tile) {
    constexpr int WBK = 32;
    // tile: 128x128 for wide layers, 128x64 when the (tap, c) extent is only 64 wide, 64x64 for K <= 64,
    // 32x128 for the <= 32-channel stem layers (a 64-row tile would spend >= half its MFMAs on padding)
    WgradGeom g;
    if (tile == 1) { g.bm = 128; g.bn = 128; }
    else if (tile == 2) { g.bm = 64; g.bn = 64; }
    else if (tile == 3) { g.bm = 128; g.bn = 64; }
    else if (tile == 4) { g.bm = 32; g.bn = 128; }
    else if (K > 64 && Nn >= 4096 && P <= 4096 && wgrad_fc_tile64()) { g.bm = 64; g.bn = 64; }      // fc1-class: see wgrad_fc_tile64()
    else if (K > 64) { g.bm = 128; g.bn = (Nn > 64 && P >= 32768) ? 128 : 64; }
    else if (K > 32) { g.bm = 64; g.bn = 64; }
    else { g.bm = 32; g.bn = 128; }
    g.tiles = ((K + g.bm - 1) / g.bm) * ((Nn + g.bn - 1) / g.bn);
    // aim at ~1024 workgroups, at least 256 pixels (8 slabs) per split
    long splits = (1024 + g.tiles - 1) / g.tiles;
    const long max_splits = (P + 255) / 256;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int pps = (int)((P + splits - 1) / splits);
    pps = (pps + WBK - 1) / WBK * WBK;
    g.pps = pps;
    g.splits = (P + pps - 1) / pps;
    return g;
}

static int conv2d_wgrad_impl(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int K, int R, int S,
                             int stride, int pad, int ldx, int lddy, int accumulate, int tile, void* stream, const DetArgs& det,
                             const MultiSrc* ms = nullptr) {
    // tile + 16: the same tile with the XCD-contiguous workgroup order; tile + 32: with the plain order; 0..4: the launcher decides
    int xcd_order = -1;
    if (tile >= 32) { xcd_order = 0; tile -= 32; }
    else if (tile >= 16) { xcd_order = 1; tile -= 16; }
    if (tile < 0 || tile > 4) return OMNI_ERR_ARG;
    ConvP p{x, dy, nullptr, dw, N, H, W, C, (H + 2 * pad - R) / stride + 1, (W + 2 * pad - S) / stride + 1, K,
            R, S, stride, pad, ldx, 0, lddy, 0, accumulate, 1};
    p.stats = nullptr;
    p.ws = nullptr;
    p.ctr = nullptr;
    p.nsrc = 0;
    if (ms != nullptr) {      // x is the channel concatenation of ms->xs[]: 1 x 1 / stride 1
        if (ms->nsrc < 1 || ms->nsrc > OMNI_MAX_SRC || R != 1 || S != 1 || stride != 1 || pad != 0) return OMNI_ERR_ARG;
        p.coff[0] = 0;
        for (int q = 0; q < OMNI_MAX_SRC; ++q) {
            const bool live = q < ms->nsrc;
            if (live && (ms->xs[q] == nullptr || ms->cs[q] <= 0 || (ms->cs[q] & 3))) return OMNI_ERR_ARG;
            p.xs[q] = live ? (const float*)ms->xs[q] : nullptr;
            p.coff[q + 1] = p.coff[q] + (live ? ms->cs[q] : 0);
        }
        if (p.coff[ms->nsrc] != C || ldx != C) return OMNI_ERR_ARG;
        p.nsrc = ms->nsrc;
        p.x = p.xs[0];
    }
    if (bad_geom(p) || (K & 3) || (ldx & 3) || (lddy & 3) || ldx < C || lddy < K) return OMNI_ERR_ARG;
    const long P = (long)N * p.OH * p.OW;
    const int Nn = R * S * C;
    if (P == 0 && det.plan != nullptr) { det.plan[0] = tile; det.plan[1] = 1; det.plan[2] = det.plan[3] = 0; return OMNI_OK; }
    if (P == 0) {
        if (!accumulate) omni_memset_async(dw, 0, sizeof(float) * (size_t)K * Nn, (hipStream_t)stream);
        return OMNI_OK;
    }
    constexpr int WBK = 32;
    const WgradGeom geo = wgrad_geom(K, Nn, P, tile);
    const int bm = geo.bm, bn = geo.bn, tiles = geo.tiles, pps = geo.pps;
    const long splits = geo.splits;
    if (det.plan != nullptr) {
        det_plan(det, tile, tiles, splits, (long)bm * bn);
        return OMNI_OK;
    }
    const bool ordered = det.ctr != nullptr && splits > 1;
    if (ordered && !det_fits(det, tiles, splits, (long)bm * bn)) return OMNI_ERR_ARG;
    if (det.ctr != nullptr) {             // deterministic: ordered split sums, plain read-modify-write for accumulate
        p.ws = det.ws;
        p.ctr = det.ctr;
    }
    if (splits > 1 && !accumulate && !ordered) omni_memset_async(dw, 0, sizeof(float) * (size_t)K * Nn, (hipStream_t)stream);
    if (xcd_order < 0) xcd_order = (WGRAD_XCD_ORDER_FC && R == 1 && S == 1 && H == 1 && W == 1 && splits == 1) ? 1 : 0;
    p.relu = xcd_order;
    // deep-prefetch form (conv_wgrad_pf_kernel): single-source problems whose operands a 32-bit buffer offset reaches
    static const int wgrad_pf = [] { const char* e = getenv("OMNI_WGRAD_PF"); return e ? atoi(e) : 2; }();      // A/B knob: 0 = the classic body
    // (16 MiB of head-room: the byte cursors run up to PF slabs past the last pixel before their loads are masked)
    const bool pf_ok = wgrad_pf >= 2 && ms == nullptr && (long)P * lddy * 4 < (1L << 31) - (1L << 24) && (long)N * H * W * ldx * 4 < (1L << 31) - (1L << 24);
    // MEASURED and left OFF (profiles/r06_ab_wgrad_xcd_splits.log): 10.66-10.68 ms with, 10.66-10.68 without; the 3x3/s2 64->128 launch
    // 52 us either way -- the re-fetched ranges come out of the memory-side cache, not HBM
    static const int wgrad_xcd_splits = [] { const char* e = getenv("OMNI_WGRAD_XCD_SPLITS"); return e ? atoi(e) : 0; }();      // A/B knob
    const int xcd_splits = (wgrad_xcd_splits && splits >= 8 && tiles > 1) ? 1 : 0;
#define OMNI_WGRAD_PF(BM_, BN_, WM_, WN_, PF_)                                                                          \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_wgrad_pf_kernel<BM_, BN_, WM_, WN_, WBK, PF_>), dim3(tiles, (unsigned)splits), dim3(256), 0, \
                       (hipStream_t)stream, p, pps, xcd_splits)
    if (pf_ok) {
        // (wgrad_pf == 3, A/B: three slabs in flight on the tiles whose registers allow it without losing a wave per SIMD)
        if (bm == 128 && bn == 128) OMNI_WGRAD_PF(128, 128, 2, 2, 2);
        else if (bm == 128) OMNI_WGRAD_PF(128, 64, 2, 2, 2);
        else if (bm == 64) { if (wgrad_pf >= 3) OMNI_WGRAD_PF(64, 64, 2, 2, 3); else OMNI_WGRAD_PF(64, 64, 2, 2, 2); }
        else { if (wgrad_pf >= 3) OMNI_WGRAD_PF(32, 128, 1, 4, 3); else OMNI_WGRAD_PF(32, 128, 1, 4, 2); }
        return omni_launch_status();
    }
#undef OMNI_WGRAD_PF
#define OMNI_WGRAD(BM_, BN_, WM_, WN_)                                                                                  \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_wgrad_kernel<BM_, BN_, WM_, WN_, WBK>), dim3(tiles, (unsigned)splits), dim3(256), 0, \
                       (hipStream_t)stream, p, pps)
    if (bm == 128 && bn == 128) OMNI_WGRAD(128, 128, 2, 2);
    else if (bm == 128) OMNI_WGRAD(128, 64, 2, 2);
    else if (bm == 64) OMNI_WGRAD(64, 64, 2, 2);
    else OMNI_WGRAD(32, 128, 1, 4);
#undef OMNI_WGRAD
    return omni_launch_status();
}

int omni_conv2d_wgrad_algo(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int K, int R, int S,
                           int stride, int pad, int ldx, int lddy, int accumulate, int tile, void* stream) {
    return conv2d_wgrad_impl(x, dy, dw, N, H, W, C, K, R, S, stride, pad, ldx, lddy, accumulate, tile, stream, DetArgs{nullptr, 0, nullptr, 0, nullptr});
}

// deterministic form: ctr must be non-null even for a launch that is not split (it selects the plain read-modify-write epilogue of an
// accumulating launch; one counter is enough then)
int omni_conv2d_wgrad_det(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int K, int R, int S, int stride, int pad,
                          int ldx, int lddy, int accumulate, int tile, float* ws, long long ws_floats, int* ctr, int n_ctr, long long* plan,
                          void* stream) {
    return conv2d_wgrad_impl(x, dy, dw, N, H, W, C, K, R, S, stride, pad, ldx, lddy, accumulate, tile, stream,
                             DetArgs{ws, ws_floats, (unsigned*)ctr, n_ctr, plan});
}

// Round 6: n direct weight gradients (dense tensors: ldx = C, lddy = K; square filters) in as few launches as their tile shapes allow
// -- normally ONE (conv_wgrad_multi_kernel; <= 12 problems per launch, problems of another tile shape go to a launch of their own).
// Problem i: dw[i] (K, R, R, C) (+)= dy[i] (N, OH, OW, K)^T x[i] (N, H, W, C); nsrc[i] > 0: x is the channel concatenation of the
// nsrc[i] tensors xs[i * 6 + s] of widths cs[i * 6 + s] (1 x 1 / stride 1, as omni_conv2d_wgrad_multi_det).  Deterministic form only
// (ctr != NULL): every problem keeps the split structure, workspace layout and counters of its own omni_conv2d_wgrad_det launch
// (bit-identical results), at offsets inside ws / ctr; plan != NULL: plan[2] = counters, plan[3] = workspace floats, nothing launched.
int omni_conv2d_wgrad_batch_det(const void* const* x, const void* const* dy, const void* const* dw, const int* N, const int* H, const int* W,
                                const int* C, const int* K, const int* R, const int* stride, const int* pad, const int* accumulate,
                                const void* const* xs, const int* cs, const int* nsrc, int n, float* ws, long long ws_floats, int* ctr,
                                int n_ctr, long long* plan, void* stream) {
    if (n <= 0 || n > 64 || x == nullptr || dy == nullptr || dw == nullptr) return OMNI_ERR_ARG;
    if (plan == nullptr && ctr == nullptr) return OMNI_ERR_ARG;
    struct One { ConvP p; WgradGeom g; long ws_off; int ctr_off; long work; };
    One one[64];
    long ws_need = 0;
    long ctr_need = 1;        // (counter 0: the unsplit accumulating launches' marker, see omni_conv2d_wgrad_det)
    int live = 0;
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < n; ++i) {
        const int S = R[i];
        ConvP p{(const float*)x[i], (const float*)dy[i], nullptr, (float*)dw[i], N[i], H[i], W[i], C[i], (H[i] + 2 * pad[i] - R[i]) / stride[i] + 1,
                (W[i] + 2 * pad[i] - S) / stride[i] + 1, K[i], R[i], S, stride[i], pad[i], C[i], 0, K[i], 0, accumulate[i], 1};
        p.stats = nullptr; p.ws = nullptr; p.ctr = nullptr; p.nsrc = 0;
        p.xb = p.wb = p.ob = 0;
        const int ns = nsrc != nullptr ? nsrc[i] : 0;
        if (ns > 0) {
            if (ns > OMNI_MAX_SRC || xs == nullptr || cs == nullptr || R[i] != 1 || stride[i] != 1 || pad[i] != 0) return OMNI_ERR_ARG;
            p.coff[0] = 0;
            for (int q = 0; q < OMNI_MAX_SRC; ++q) {
                const bool ok = q < ns;
                const int c = ok ? cs[i * OMNI_MAX_SRC + q] : 0;
                if (ok && (xs[i * OMNI_MAX_SRC + q] == nullptr || c <= 0 || (c & 3))) return OMNI_ERR_ARG;
                p.xs[q] = ok ? (const float*)xs[i * OMNI_MAX_SRC + q] : nullptr;
                p.coff[q + 1] = p.coff[q] + c;
            }
            if (p.coff[ns] != C[i]) return OMNI_ERR_ARG;
            p.nsrc = ns;
            p.x = p.xs[0];
        } else {
            for (int q = 0; q < OMNI_MAX_SRC; ++q) { p.xs[q] = nullptr; p.coff[q] = 0; }
            p.coff[OMNI_MAX_SRC] = 0;
        }
        if (p.x == nullptr || p.w == nullptr || p.out == nullptr || bad_geom(p) || (K[i] & 3) || (C[i] & 3)) return OMNI_ERR_ARG;
        const long P = (long)N[i] * p.OH * p.OW;
        const int Nn = R[i] * S * C[i];
        if (P == 0) {
            if (plan == nullptr && !accumulate[i]) omni_memset_async((void*)dw[i], 0, sizeof(float) * (size_t)K[i] * Nn, st);
            continue;
        }
        One& o = one[live++];
        o.p = p;
        o.g = wgrad_geom(K[i], Nn, P, 0);
        o.ws_off = ws_need;
        o.ctr_off = (int)ctr_need;
        o.work = (long)o.g.pps;
        if (o.g.splits > 1) {
            ws_need += omni_split_ws_floats(o.g.tiles, o.g.splits, (long)o.g.bm * o.g.bn);
            ctr_need += omni_split_counters(o.g.tiles, o.g.splits);
        }
    }
    if (plan != nullptr) { plan[0] = 0; plan[1] = 0; plan[2] = ctr_need; plan[3] = ws_need; return OMNI_OK; }
    if (live == 0) return OMNI_OK;
    if (n_ctr < ctr_need || (ws_need > 0 && (ws == nullptr || ws_floats < ws_need))) return OMNI_ERR_ARG;
    // longest workgroups first (dispatch follows the id): the short ones fill the tail
    for (int a = 1; a < live; ++a) {
        const One v = one[a];
        int b = a;
        for (; b > 0 && one[b - 1].work < v.work; --b) one[b] = one[b - 1];
        one[b] = v;
    }
    static const int shapes[4][2] = {{128, 128}, {128, 64}, {64, 64}, {32, 128}};
    for (int sh = 0; sh < 4; ++sh) {
        int j0 = 0;
        while (true) {
            ConvWMulti m;
            m.n = 0;
            long first = 0;
            int j = j0;
            for (; j < live && m.n < WGRAD_MULTI_MAX; ++j) {
                One& o = one[j];
                if (o.g.bm != shapes[sh][0] || o.g.bn != shapes[sh][1]) continue;
                const long blocks = (long)o.g.tiles * o.g.splits;
                if (first + blocks > 0x7fffffff) return OMNI_ERR_ARG;
                o.p.ws = o.g.splits > 1 ? ws + o.ws_off : ws;
                o.p.ctr = (unsigned*)ctr + (o.g.splits > 1 ? o.ctr_off : 0);
                o.p.relu = 0;
                m.p[m.n] = o.p;
                m.pps[m.n] = o.g.pps;
                m.tiles[m.n] = o.g.tiles;
                m.splits[m.n] = (int)o.g.splits;
                m.first[m.n] = (int)first;
                first += blocks;
                ++m.n;
            }
            j0 = j;
            if (m.n == 0) break;
            for (int q = m.n; q <= WGRAD_MULTI_MAX; ++q) m.first[q] = (int)first;
            for (int q = m.n; q < WGRAD_MULTI_MAX; ++q) { m.pps[q] = 32; m.tiles[q] = 1; m.splits[q] = 1; m.p[q] = m.p[0]; }
            constexpr int WBK = 32;
            if (sh == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_wgrad_multi_kernel<128, 128, 2, 2, WBK>), dim3((unsigned)first), dim3(256), 0, st, m);
            else if (sh == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_wgrad_multi_kernel<128, 64, 2, 2, WBK>), dim3((unsigned)first), dim3(256), 0, st, m);
            else if (sh == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_wgrad_multi_kernel<64, 64, 2, 2, WBK>), dim3((unsigned)first), dim3(256), 0, st, m);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_wgrad_multi_kernel<32, 128, 1, 4, WBK>), dim3((unsigned)first), dim3(256), 0, st, m);
            if (j0 >= live) break;
        }
    }
    return omni_launch_status();
}

// omni_conv2d_wgrad_det for an input that is the channel concatenation of nsrc <= 6 dense NHWC tensors (see omni_conv2d_fwd_multi_det):
// dw (K, 1, 1, sum cs) = dy^T x without the concatenated copy; cs[s] % 4 == 0.  ctr == NULL && plan == NULL: non-deterministic form.
int omni_conv2d_wgrad_multi_det(const void* const* xs, const int* cs, int nsrc, const float* dy, float* dw, int N, int H, int W, int K,
                                int lddy, int accumulate, int tile, float* ws, long long ws_floats, int* ctr, int n_ctr, long long* plan,
                                void* stream) {
    if (xs == nullptr || cs == nullptr || nsrc < 1 || nsrc > OMNI_MAX_SRC) return OMNI_ERR_ARG;
    long C = 0;
    for (int q = 0; q < nsrc; ++q) C += cs[q] > 0 ? cs[q] : 0;
    if (C <= 0 || C > (1 << 20)) return OMNI_ERR_ARG;
    const MultiSrc ms{xs, cs, nsrc};
    return conv2d_wgrad_impl((const float*)xs[0], dy, dw, N, H, W, (int)C, K, 1, 1, 1, 0, (int)C, lddy, accumulate, tile, stream,
                             DetArgs{ws, ws_floats, (unsigned*)ctr, n_ctr, plan}, &ms);
}

int omni_conv2d_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int K, int R, int S,
                      int stride, int pad, int ldx, int lddy, int accumulate, void* stream) {
    return omni_conv2d_wgrad_algo(x, dy, dw, N, H, W, C, K, R, S, stride, pad, ldx, lddy, accumulate, 0, stream);
}

// ---- batched GEMMs of the Winograd path (csrc/winograd.hip): `batch` independent dense problems in one launch ----
// out[b] (M x K) = x[b] (M x C) * w[b] (K x C)^T
// algo: 0 auto | 1 = persistent 128x128 workgroups walking the (problem, tile) list (`workgroups` of them, 0 = 512; needs
// C % 32 == 0) | 2 = one 128x128 tile per workgroup | 3 = one 64x64 tile per workgroup | 4 = 64x64 tiles with 2-4 slabs of
// buffer-load prefetch in flight (gemm_nt_pf_kernel; needs C % 64 == 0) | 5 = algo 4's tiles walked by `workgroups` persistent
// workgroups (0 = 1024, a multiple of 8) with the slab stream running through the tile boundaries (gemm_nt_pfp_kernel)
int omni_gemm_batched_fwd_algo(const float* x, const float* w, float* out, int batch, int M, int C, int K, int algo, int workgroups,
                               void* stream) {
    if (batch <= 0 || M < 0 || C <= 0 || K <= 0 || (C & 3) || algo < 0 || algo > 5 || workgroups < 0) return OMNI_ERR_ARG;
    if ((algo == 1 || algo == 5) && ((C % 32) != 0 || (workgroups & 7))) return OMNI_ERR_ARG;
    if (M == 0) return OMNI_OK;
    ConvP p{x, w, nullptr, out, M, 1, 1, C, 1, 1, K, 1, 1, 1, 0, C, K, 0, 0, 0, 1, (long)M * C, (long)K * C, (long)M * K};
    const long t128 = (((long)M + 127) / 128) * ((K + 127) / 128);
    // measured per shape (tools/sweep_batched_gemm.py, hipGraph replay): the persistent kernel from 1024 128x128 tiles up, 64x64 tiles
    // below (36x[1024x256]x[256x256]^T: 60 us against 76 us with one 128x128 tile per workgroup)
    const bool auto_choice = algo == 0;
    if (algo == 0) algo = (K > 64 && (C % 32) == 0 && t128 * batch >= 1024 && !(FWD64_DEEP_PREFETCH && (C % 64) == 0)) ? 1 : 3;
    if (algo == 1) {
        // >= 2 items per resident workgroup: persistent kernel with the prefetch carried across items
        GemmP g{x, w, out, batch, M, K, C, (M + 127) / 128, (K + 127) / 128};
        hipLaunchKernelGGL(gemm_nt_persistent_kernel, dim3(workgroups ? workgroups : 512), dim3(256), 0, (hipStream_t)stream, g);
        return omni_launch_status();
    }
    if (algo == 3 && FWD64_DEEP_PREFETCH && (C % 64) == 0 && (long)M * C * 4 < (1L << 31) && (long)K * C * 4 < (1L << 31)) algo = 4;
    // many-tile launches: the persistent form from OMNI_GEMM_PERSIST_MIN_ITEMS 64x64 tiles up (A/B knob; 0 = never)
    static const long persist_min = [] { const char* e = getenv("OMNI_GEMM_PERSIST_MIN_ITEMS"); return e ? atol(e) : 0L; }();
    static const int persist_wgs = [] { const char* e = getenv("OMNI_GEMM_PERSIST_WGS"); return e ? atoi(e) : 0; }();
    // outputs of at least this many MB leave through `nt` stores (A/B knob; 0 = never)
    static const long nt_min_mb = [] { const char* e = getenv("OMNI_GEMM_NT_OUT_MIN_MB"); return e ? atol(e) : 0L; }();
    const int nt_out = (nt_min_mb > 0 && (long)batch * M * K * 4 >= nt_min_mb * (1L << 20)) ? 1 : 0;
    if (algo == 4 && auto_choice && persist_min > 0 && (((long)M + 63) / 64) * ((K + 63) / 64) * batch >= persist_min &&
        (long)batch * M * C * 4 < (1L << 31) && (long)batch * K * C * 4 < (1L << 31) && (persist_wgs & 7) == 0) {
        algo = 5;
        workgroups = persist_wgs;
    }
    if (algo == 5) {
        const long items = (((long)M + 63) / 64) * ((K + 63) / 64) * batch;
        if ((C % 64) != 0 || (long)batch * M * C * 4 >= (1L << 31) || (long)batch * K * C * 4 >= (1L << 31) || items > 0x7fffffff)
            return OMNI_ERR_ARG;
        GemmTP g{x, w, out, M, K, C, (long)M * C, (long)K * C, (long)M * K, nullptr, nullptr, batch, nt_out};
        long wgs = workgroups ? workgroups : 1024;
        if (wgs > items) wgs = (items + 7) / 8 * 8;
        if ((C % 128) == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_nt_pfp_kernel<4>), dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, g);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_nt_pfp_kernel<2>), dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, g);
        return omni_launch_status();
    }
    if (algo == 2)   // else 64x64 tiles: 4x the workgroups (measured on the 256ch @32x32 layers)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_fwd_kernel<128, 128, 2, 2, 32>), dim3((unsigned)t128, 1, (unsigned)batch), dim3(256), 0,
                           (hipStream_t)stream, p);
    else if (algo == 3)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_fwd_kernel<64, 64, 2, 2, 32>),
                           dim3((unsigned)((((long)M + 63) / 64) * ((K + 63) / 64)), 1, (unsigned)batch), dim3(256), 0,
                           (hipStream_t)stream, p);
    else {           // 64x64 tiles, PF slabs of buffer-load prefetch in flight (gemm_nt_pf_kernel)
        if ((C % 64) != 0 || (long)M * C * 4 >= (1L << 31) || (long)K * C * 4 >= (1L << 31)) return OMNI_ERR_ARG;
        GemmTP g{x, w, out, M, K, C, (long)M * C, (long)K * C, (long)M * K, nullptr, nullptr, batch, nt_out};
        const long wgs = (((long)M + 63) / 64) * ((K + 63) / 64) * batch;
        if (wgs > 0x7fffffff) return OMNI_ERR_ARG;
        const dim3 grid((unsigned)wgs);
        if ((C % 128) == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_nt_pf_kernel<4>), grid, dim3(256), 0, (hipStream_t)stream, g);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_nt_pf_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, g);
    }
    return omni_launch_status();
}

int omni_gemm_batched_fwd(const float* x, const float* w, float* out, int batch, int M, int C, int K, void* stream) {
    return omni_gemm_batched_fwd_algo(x, w, out, batch, M, C, K, 0, 0, stream);
}

// dw[b] (K x C) = dy[b] (M x K)^T * x[b] (M x C)      (overwrites dw)
// algo: 0 auto | 1 = the tile kernels of the implicit-GEMM weight gradient (128x128 / 64x64, one slab of prefetch) | 2 = 64x64 tiles
// with 2-4 slabs of buffer-load prefetch in flight (gemm_tn_pf_kernel)
constexpr bool WGRAD64_DEEP_PREFETCH = true;    // gemm_tn_pf_kernel (profiles/r03_sweep_batched_gemm.log: 3-66 % faster on every Winograd weight-gradient shape)

// A/B knob (OMNI_WGRAD_LDS_PAD, bytes of unused dynamic LDS per workgroup): bounds how many weight-gradient workgroups a CU hosts, so
// that workgroups of the critical-path stream always find registers / LDS free beside them
static inline unsigned tn_lds_pad() {
    static const unsigned pad = [] {
        const char* e = getenv("OMNI_WGRAD_LDS_PAD");
        const long v = e != nullptr ? atol(e) : 0;
        return (unsigned)(v < 0 ? 0 : v > 32768 ? 32768 : v);
    }();
    return pad;
}

// 64x64 tiles of gemm_tn_pf_kernel: row splits of >= 8 slabs, aiming at >= 512 workgroups
static inline void tn_pf_plan(int batch, int M, int C, int K, int& tiles, long& splits, int& rps) {
    tiles = ((K + 63) / 64) * ((C + 63) / 64);
    splits = (512 + (long)tiles * batch - 1) / ((long)tiles * batch);
    const long max_splits = ((long)M + 255) / 256;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    rps = (int)(((long)M + splits - 1) / splits);
    rps = (rps + 127) / 128 * 128;                                   // multiple of 32 * PF for PF = 4
    splits = ((long)M + rps - 1) / rps;
}

static int gemm_batched_wgrad_impl(const float* x, const float* dy, float* dw, int batch, int M, int C, int K, int algo, void* stream,
                                   const DetArgs& det) {
    if (batch <= 0 || M < 0 || C <= 0 || K <= 0 || (C & 3) || (K & 3) || algo < 0 || algo > 2) return OMNI_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (M == 0) {
        if (det.plan != nullptr) { det.plan[0] = algo; det.plan[1] = 1; det.plan[2] = det.plan[3] = 0; return OMNI_OK; }
        omni_memset_async(dw, 0, sizeof(float) * (size_t)batch * K * C, st);
        return OMNI_OK;
    }
    const bool fits = (long)M * C * 4 < (1L << 31) && (long)M * K * 4 < (1L << 31);
    if (algo == 2 && !fits) return OMNI_ERR_ARG;
    if (algo == 0) algo = (WGRAD64_DEEP_PREFETCH && fits) ? 2 : 1;
    if (algo == 2) {
        int tiles, rps;
        long splits;
        tn_pf_plan(batch, M, C, K, tiles, splits, rps);
        if (det.plan != nullptr) {
            det_plan(det, algo, (long)tiles * batch, splits, 64 * 64);
            return OMNI_OK;
        }
        const bool ordered = det.ctr != nullptr && splits > 1;
        if (ordered && !det_fits(det, (long)tiles * batch, splits, 64 * 64)) return OMNI_ERR_ARG;
        if (splits > 1 && !ordered) omni_memset_async(dw, 0, sizeof(float) * (size_t)batch * K * C, st);
        GemmTP g{dy, x, dw, M, K, C, (long)M * K, (long)M * C, (long)K * C, ordered ? det.ws : nullptr, ordered ? det.ctr : nullptr};
        const long wgs = (long)tiles * splits * batch;
        if (wgs > 0x7fffffff) return OMNI_ERR_ARG;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_tn_pf_kernel<4>), dim3((unsigned)wgs), dim3(256), 0, st, g, rps, (int)splits);
        return omni_launch_status();
    }
    ConvP p{x, dy, nullptr, dw, M, 1, 1, C, 1, 1, K, 1, 1, 1, 0, C, 0, K, 0, 0, 1, (long)M * C, (long)M * K, (long)K * C};
    constexpr int WBK = 32;
    const bool wide = K > 64 && C > 64;
    const int bm = wide ? 128 : 64, bn = wide ? 128 : 64;
    const int tiles = ((K + bm - 1) / bm) * ((C + bn - 1) / bn);
    long splits = (1024 + (long)tiles * batch - 1) / ((long)tiles * batch);
    const long max_splits = ((long)M + 255) / 256;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int pps = (int)(((long)M + splits - 1) / splits);
    pps = (pps + WBK - 1) / WBK * WBK;
    splits = ((long)M + pps - 1) / pps;
    p.stats = nullptr;
    p.ws = nullptr;
    p.ctr = nullptr;
    if (det.plan != nullptr) {
        det_plan(det, algo, (long)tiles * batch, splits, (long)bm * bn);
        return OMNI_OK;
    }
    const bool ordered = det.ctr != nullptr && splits > 1;
    if (ordered) {
        if (!det_fits(det, (long)tiles * batch, splits, (long)bm * bn)) return OMNI_ERR_ARG;
        p.ws = det.ws;
        p.ctr = det.ctr;
    }
    if (splits > 1 && !ordered) omni_memset_async(dw, 0, sizeof(float) * (size_t)batch * K * C, st);
    if (wide)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_wgrad_kernel<128, 128, 2, 2, WBK>), dim3(tiles, (unsigned)splits, (unsigned)batch),
                           dim3(256), 0, st, p, pps);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_wgrad_kernel<64, 64, 2, 2, WBK>), dim3(tiles, (unsigned)splits, (unsigned)batch),
                           dim3(256), 0, st, p, pps);
    return omni_launch_status();
}

int omni_gemm_batched_wgrad_algo(const float* x, const float* dy, float* dw, int batch, int M, int C, int K, int algo, void* stream) {
    return gemm_batched_wgrad_impl(x, dy, dw, batch, M, C, K, algo, stream, DetArgs{nullptr, 0, nullptr, 0, nullptr});
}

int omni_gemm_batched_wgrad_det(const float* x, const float* dy, float* dw, int batch, int M, int C, int K, int algo, float* ws,
                                long long ws_floats, int* ctr, int n_ctr, long long* plan, void* stream) {
    return gemm_batched_wgrad_impl(x, dy, dw, batch, M, C, K, algo, stream, DetArgs{ws, ws_floats, (unsigned*)ctr, n_ctr, plan});
}

// n <= 16 independent batched weight-gradient GEMMs of different shapes in ONE launch (gemm_tn_multi_kernel).  Problem i:
// dw[i][b] (K[i] x C[i]) = dy[i][b] (M[i] x K[i])^T x[i][b] (M[i] x C[i]), b < batch[i]; each with the tiles, row splits and (ctr !=
// nullptr) ordered split reduction of omni_gemm_batched_wgrad_det(algo 2) on that problem alone -- the results are bit-identical
// to n separate calls.  Workspace / counters: the problems' regions one after the other; plan (nullable) reports [0] = 2,
// [1] = 0, [2] = counters, [3] = workspace floats of the whole call and returns without launching.
int omni_gemm_batched_wgrad_multi(const void* const* x, const void* const* dy, const void* const* dw, const int* batch, const int* M,
                                  const int* C, const int* K, int n, float* ws, long long ws_floats, int* ctr, int n_ctr,
                                  long long* plan, void* stream) {
    if (n <= 0 || n > TN_MULTI_MAX || x == nullptr || dy == nullptr || dw == nullptr) return OMNI_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    struct One { int i, tiles, rps; long splits, ws_off, ctr_off; };
    One one[TN_MULTI_MAX];
    long ws_need = 0, ctr_need = 0;
    int live = 0;
    for (int i = 0; i < n; ++i) {
        if (batch[i] <= 0 || M[i] < 0 || C[i] <= 0 || K[i] <= 0 || (C[i] & 3) || (K[i] & 3) || dw[i] == nullptr ||
            (M[i] > 0 && (x[i] == nullptr || dy[i] == nullptr)))
            return OMNI_ERR_ARG;
        if ((long)M[i] * C[i] * 4 >= (1L << 31) || (long)M[i] * K[i] * 4 >= (1L << 31)) return OMNI_ERR_ARG;
        if (M[i] == 0) {
            if (plan == nullptr) omni_memset_async((void*)dw[i], 0, sizeof(float) * (size_t)batch[i] * K[i] * C[i], st);
            continue;
        }
        One& o = one[live++];
        o.i = i;
        tn_pf_plan(batch[i], M[i], C[i], K[i], o.tiles, o.splits, o.rps);
        o.ws_off = ws_need;
        o.ctr_off = ctr_need;
        if (o.splits > 1) {
            ws_need += omni_split_ws_floats((long)o.tiles * batch[i], o.splits, 64 * 64);
            ctr_need += omni_split_counters((long)o.tiles * batch[i], o.splits);
        }
    }
    if (plan != nullptr) {
        plan[0] = 2; plan[1] = 0; plan[2] = ctr_need; plan[3] = ws_need;
        return OMNI_OK;
    }
    if (live == 0) return OMNI_OK;
    const bool ordered = ctr != nullptr;
    if (ordered && ctr_need > 0 && (ws == nullptr || ws_floats < ws_need || n_ctr < ctr_need)) return OMNI_ERR_ARG;
    // longest workgroups first (dispatch follows the id): the short ones fill the tail
    for (int a = 1; a < live; ++a) {
        const One v = one[a];
        int b = a;
        for (; b > 0 && one[b - 1].rps < v.rps; --b) one[b] = one[b - 1];
        one[b] = v;
    }
    GemmTnMulti t;
    t.n = live;
    long first = 0;
    for (int j = 0; j < live; ++j) {
        const One& o = one[j];
        const int i = o.i;
        const bool split = o.splits > 1;
        if (split && !ordered) omni_memset_async((void*)dw[i], 0, sizeof(float) * (size_t)batch[i] * K[i] * C[i], st);
        t.p[j] = GemmTP{(const float*)dy[i], (const float*)x[i], (float*)dw[i], M[i], K[i], C[i], (long)M[i] * K[i], (long)M[i] * C[i],
                        (long)K[i] * C[i], (split && ordered) ? ws + o.ws_off : nullptr, (split && ordered) ? (unsigned*)ctr + o.ctr_off : nullptr};
        t.rps[j] = o.rps;
        t.splits[j] = (int)o.splits;
        const long items = (long)o.tiles * o.splits * batch[i];
        if (items > 0x3fffffff) return OMNI_ERR_ARG;
        t.items[j] = (int)items;
        t.first[j] = (int)first;
        first += (items + 7) / 8 * 8;
        if (first > 0x7fffffff) return OMNI_ERR_ARG;
    }
    t.first[live] = (int)first;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_tn_multi_kernel<4>), dim3((unsigned)first), dim3(256), tn_lds_pad(), st, t);
    return omni_launch_status();
}

int omni_gemm_batched_wgrad(const float* x, const float* dy, float* dw, int batch, int M, int C, int K, void* stream) {
    return omni_gemm_batched_wgrad_algo(x, dy, dw, batch, M, C, K, 0, stream);
}

// ---- grouped convolution: nn.Conv2d(C, K, R, stride, pad, groups=G, bias=False), the 3x3 of DLA's BottleneckX
// (cubercnn/modeling/backbone/dla.py:112-153; 32 or 64 groups of 4..32 channels).  The G independent convolutions read / write
// channel slices of the same NHWC tensors: the implicit-GEMM kernels already take the pixel pitch (ldx / ldo) separately from
// the channel count, so every group is one launch of them on offset pointers; w (K, R, S, C/G) keeps a group's filters contiguous.
int omni_grouped_conv2d_fwd(const float* x, const float* w, float* out, int N, int H, int W, int C, int K, int R, int S, int stride,
                            int pad, int groups, void* stream) {
    if (groups <= 0 || C % groups || K % groups || ((C / groups) & 3) || ((K / groups) & 3)) return OMNI_ERR_ARG;
    const int Cg = C / groups, Kg = K / groups;
    for (int g = 0; g < groups; ++g) {
        const int rc = conv2d_fwd_impl(x + g * Cg, w + (size_t)g * Kg * R * S * Cg, nullptr, out + g * Kg, N, H, W, Cg, Kg, R, S, stride,
                                       pad, C, K, 0, 0, 0, nullptr, 0, nullptr, stream);
        if (rc != OMNI_OK) return rc;
    }
    return OMNI_OK;
}

int omni_grouped_conv2d_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int K, int R, int S, int stride,
                              int pad, int groups, void* stream) {
    if (groups <= 0 || C % groups || K % groups || ((C / groups) & 3) || ((K / groups) & 3)) return OMNI_ERR_ARG;
    const int Cg = C / groups, Kg = K / groups;
    for (int g = 0; g < groups; ++g) {
        const int rc = omni_conv2d_dgrad_algo(dy + g * Kg, w + (size_t)g * Kg * R * S * Cg, dx + g * Cg, N, H, W, Cg, Kg, R, S, stride, pad,
                                              K, C, 0, 0, 0, stream);
        if (rc != OMNI_OK) return rc;
    }
    return OMNI_OK;
}

// dw (K, R, S, C/G) overwritten
int omni_grouped_conv2d_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int K, int R, int S, int stride,
                              int pad, int groups, void* stream) {
    if (groups <= 0 || C % groups || K % groups || ((C / groups) & 3) || ((K / groups) & 3)) return OMNI_ERR_ARG;
    const int Cg = C / groups, Kg = K / groups;
    for (int g = 0; g < groups; ++g) {
        const int rc = omni_conv2d_wgrad_algo(x + g * Cg, dy + g * Kg, dw + (size_t)g * Kg * R * S * Cg, N, H, W, Cg, Kg, R, S, stride, pad,
                                              C, K, 0, 0, stream);
        if (rc != OMNI_OK) return rc;
    }
    return OMNI_OK;
}

}  // extern "C"
