// gemm_split.hip -- OPT-IN EXPERIMENT (round 6; SURVEY.md section 7 "3 x bf16 split"): the batched NT GEMM of the
// Winograd point products,  out[b] (M x N) = A[b] (M x K) * B[b] (N x K)^T  with fp32 operands and fp32 results, computed on the
// BF16 matrix cores from an error-free split of every operand:
//
//     x = x_hi + x_lo (+ x_lo2),   x_hi = bf16(x),  x_lo = bf16(x - x_hi),  x_lo2 = bf16(x - x_hi - x_lo)      (all subtractions exact)
//
//   TERMS == 3:  a * b ~ a_hi b_hi + a_hi b_lo + a_lo b_hi                       (dropped: a_lo b_lo, |.| <= 2^-16 |a b|; operands to 2^-17)
//   TERMS == 6:  ... + a_hi b_lo2 + a_lo2 b_hi + a_lo b_lo                       (dropped terms <= 2^-24 |a b|: fp32-grade products)
//
// every partial product accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (16x the multiply rate of v_mfma_f32_32x32x2_f32 on gfx950:
// 2.5 PFLOP/s dense against 157.3 TFLOP/s).  This is NOT the product's arithmetic: north_star asks for fp32 box parameters within
// 1e-4 of the reference's fp32 path, and the measured line of bench.py stays on the fp32-input MFMA kernels.  The experiment answers
// one question -- what would the contraction time be if the split were admissible -- and reports its own error against float64 next
// to the fp32-MFMA kernel's on the same inputs (bench.py `bf16_split`, tests/test_gemm_split.py).
//
// Kernel: 128 x 128 tile per workgroup, four waves of 64 x 64 (2 x 2 MFMA blocks), reduction slab 32.  A thread converts the fp32
// slab it fetched (register-staged prefetch, one slab ahead) into the bf16 planes while it stores them to LDS ([row][40] bf16: 80 B
// rows, conflict-free 16-byte fragment reads); a lane's MFMA fragment is 8 consecutive bf16 of its row.
#include <device_rt.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short us4 __attribute__((ext_vector_type(4)));
#ifndef OMNI_HIPEMU
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
#endif

__device__ __forceinline__ unsigned short bf16_rne(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_val(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

struct SplitP {
    const float* A;
    const float* B;
    float* out;
    int batch, M, N, K;
};

constexpr int SBM = 128, SBN = 128, SBK = 32, SROW = SBK + 8;        // bf16 elements per LDS row (80 bytes)

template <int TERMS>
__global__ void __launch_bounds__(256) OMNI_WAVES_PER_EU(2) gemm_nt_split_kernel(SplitP p) {
    constexpr int PL = TERMS == 3 ? 2 : 3;                           // bf16 planes per operand
    __shared__ __attribute__((aligned(16))) unsigned short s_a[PL][SBM * SROW];
    __shared__ __attribute__((aligned(16))) unsigned short s_b[PL][SBN * SROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_m = (p.M + SBM - 1) / SBM, tiles_n = (p.N + SBN - 1) / SBN, per = tiles_m * tiles_n;
    // one contiguous chunk of the (problem, tile) sequence per XCD (workgroup ids go round-robin over the 8 XCDs): the column tiles
    // that re-read an A panel share an L2 (PMC before: 340 MB fetched for 161 MB of operands)
    int item = (int)blockIdx.x;
    {
        const int total = (int)gridDim.x;
        if (total >= 8) {
            const int q = total / 8, r = total % 8, xcd = item % 8, k = item / 8;
            item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
        }
    }
    const int prob = item / per, tix = item - prob * per;
    const int m0 = (tix / tiles_n) * SBM, n0 = (tix % tiles_n) * SBN;
    const float* A = p.A + (long)prob * p.M * p.K;
    const float* B = p.B + (long)prob * p.N * p.K;
    float* out = p.out + (long)prob * p.M * p.N;

    // global -> registers: 128 rows x 8 float4 per operand = 1024 float4, 4 per thread
    const int kq = tid & 7, lrow = tid >> 3;                           // float4 column, first row (32 rows per pass)
    float4 ra[4], rb[4];
    auto load_slab = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + lrow + 32 * i, n = n0 + lrow + 32 * i;
            ra[i] = m < p.M ? *reinterpret_cast<const float4*>(A + (long)m * p.K + k0 + 4 * kq) : make_float4(0.f, 0.f, 0.f, 0.f);
            rb[i] = n < p.N ? *reinterpret_cast<const float4*>(B + (long)n * p.K + k0 + 4 * kq) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto split_store = [&](unsigned short (*plane)[SBM * SROW], int row, const float4& v) {
#ifdef OMNI_HIPEMU
        float x[4] = {v.x, v.y, v.z, v.w};
        us4 h, l, l2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned short hi = bf16_rne(x[e]);
            const float r1 = x[e] - bf16_val(hi);                      // exact
            const unsigned short lo = bf16_rne(r1);
            h[e] = hi; l[e] = lo;
            if (PL == 3) l2[e] = bf16_rne(r1 - bf16_val(lo));          // exact difference, rounded once
        }
        *reinterpret_cast<us4*>(&plane[0][row * SROW + 4 * kq]) = h;
        *reinterpret_cast<us4*>(&plane[1][row * SROW + 4 * kq]) = l;
        if (PL == 3) *reinterpret_cast<us4*>(&plane[2][row * SROW + 4 * kq]) = l2;
#else
        // v_cvt_pk_bf16_f32 (round-to-nearest-even, two values per instruction); the residuals are exact fp32 differences
        const f32x4v x = {v.x, v.y, v.z, v.w};
        const bf16x4 h = __builtin_convertvector(x, bf16x4);
        const f32x4v r1 = x - __builtin_convertvector(h, f32x4v);
        const bf16x4 l = __builtin_convertvector(r1, bf16x4);
        *reinterpret_cast<bf16x4*>(&plane[0][row * SROW + 4 * kq]) = h;
        *reinterpret_cast<bf16x4*>(&plane[1][row * SROW + 4 * kq]) = l;
        if (PL == 3) {
            const f32x4v r2 = r1 - __builtin_convertvector(l, f32x4v);
            *reinterpret_cast<bf16x4*>(&plane[2][row * SROW + 4 * kq]) = __builtin_convertvector(r2, bf16x4);
        }
#endif
    };
    auto store_slab = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            split_store(s_a, lrow + 32 * i, ra[i]);
            split_store(s_b, lrow + 32 * i, rb[i]);
        }
    };

    // two accumulator sets: the hi x hi products run the same fp32 chain the fp32-MFMA kernel runs (one rounding per k-step of a full
    // -size partial sum); the correction products (2^-8 .. 2^-16 of it) meet in a chain of their own, whose roundings are that much
    // smaller, and join once at the end.  With one set the corrections' roundings land on the full-size sum: measured 2x the error.
    f32x16 acc[2][2], cor[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; cor[i][j][r] = 0.f; }

    const int l31 = lane & 31, kh = lane >> 5;
    const int nk = p.K / SBK;
    load_slab(0);
    for (int kt = 0; kt < nk; ++kt) {
        store_slab();
        __syncthreads();
        if (kt + 1 < nk) load_slab((kt + 1) * SBK);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {                               // two k-steps of 16 per slab; this lane: k = 16 ks + 8 kh + [0, 8)
            const int ko = 16 * ks + 8 * kh;
            // B fragments of both column blocks are read once per k-step, A fragments once per row block: 9 fragments live at a time
            // (with all 12 live beside the 128 accumulator registers the compiler shuttled the accumulators between VGPRs and AGPRs
            // around every MFMA: 512 copies per slab, round 6 ISA / PMC: SQ_INSTS_VALU 3.6x the fp32 kernel's)
            const unsigned short* fb[2][PL];
#ifndef OMNI_HIPEMU
            bf16x8 vb[2][PL];
#endif
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pl = 0; pl < PL; ++pl) {
                    fb[j][pl] = &s_b[pl][(wn * 64 + j * 32 + l31) * SROW + ko];
#ifndef OMNI_HIPEMU
                    vb[j][pl] = *reinterpret_cast<const bf16x8*>(fb[j][pl]);
#endif
                }
#ifdef OMNI_HIPEMU
#define OMNI_MMA(c, pa, j, pb) c = mfma_bf16_32x32x16(fa[pa], fb[j][pb], c)
#else
#define OMNI_MMA(c, pa, j, pb) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[pa], vb[j][pb], c, 0, 0, 0)
#endif
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned short* fa[PL];
#ifndef OMNI_HIPEMU
                bf16x8 va[PL];
#endif
#pragma unroll
                for (int pl = 0; pl < PL; ++pl) {
                    fa[pl] = &s_a[pl][(wm * 64 + i * 32 + l31) * SROW + ko];
#ifndef OMNI_HIPEMU
                    va[pl] = *reinterpret_cast<const bf16x8*>(fa[pl]);
#endif
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (TERMS == 6) {
                        OMNI_MMA(cor[i][j], 1, j, 1);                  // lo lo
                        OMNI_MMA(cor[i][j], 0, j, PL - 1);             // hi lo2
                        OMNI_MMA(cor[i][j], PL - 1, j, 0);             // lo2 hi
                    }
                    OMNI_MMA(cor[i][j], 0, j, 1);                      // hi lo
                    OMNI_MMA(cor[i][j], 1, j, 0);                      // lo hi
                    OMNI_MMA(acc[i][j], 0, j, 0);                      // hi hi
                }
            }
#undef OMNI_MMA
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (m < p.M && n < p.N) out[(long)m * p.N + n] = acc[i][j][r] + cor[i][j][r];
            }
        }
}

}  // namespace

extern "C" {

// out[b] (M x N) = A[b] (M x K) B[b] (N x K)^T for b < batch, dense fp32 operands (K % 32 == 0), through the bf16 split described at the
// top of csrc/gemm_split.hip.  terms: 3 or 6.  EXPERIMENT: not on the product's default path (kernels/wino.py OMNI_GEMM_SPLIT).
int omni_gemm_batched_split(const float* A, const float* B, float* out, int batch, int M, int N, int K, int terms, void* stream) {
    if (A == nullptr || B == nullptr || out == nullptr || batch < 0 || M < 0 || N < 0 || K <= 0 || (K & 31) || (terms != 3 && terms != 6))
        return OMNI_ERR_ARG;
    const long items = (long)batch * ((M + SBM - 1) / SBM) * ((N + SBN - 1) / SBN);
    if (items == 0) return OMNI_OK;
    if (items > 0x7fffffff) return OMNI_ERR_ARG;
    SplitP p{A, B, out, batch, M, N, K};
    if (terms == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_nt_split_kernel<3>), dim3((unsigned)items), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_nt_split_kernel<6>), dim3((unsigned)items), dim3(256), 0, (hipStream_t)stream, p);
    return omni_launch_status();
}

}  // extern "C"
