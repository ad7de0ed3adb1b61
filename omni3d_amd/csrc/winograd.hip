// winograd.hip -- Winograd F(2x2, 3x3) transforms for the 3x3 / stride-1 / pad-1 convolutions with wide channel
// counts (FPN output convs and the shared RPN conv on p2 / p3: 38 % of the step's direct-convolution flops).
//
// Reference call sites: detectron2 FPN output convs (built at /root/reference/cubercnn/modeling/backbone/dla.py:500-506)
// and StandardRPNHead.conv (configs/Base.yaml:49) -- plain nn.Conv2d(256, 256, 3, padding=1) upstream.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A       per 4x4 input tile d (stride 2), 3x3 filter g, 2x2 output tile Y
// The element-wise product summed over channels is 16 independent GEMMs  M[xi] (T x K) = V[xi] (T x C) * U[xi]^T (K x C),
// T = N * H/2 * W/2 tiles, run by the batched entry points of conv_gemm.hip on the fp32 MFMA path: 2.25x fewer MFMA
// flops than the direct implicit GEMM.  The kernels here are the HBM-bound transforms around those GEMMs (float4 per
// lane along the channel dimension, consecutive lanes = consecutive channels):
//   wino_in_kernel   d (NHWC, zero padding 1)    -> V  [16][T][C]     (B^T d B)
//   wino_out_kernel  M [16][T][K] (+bias, ReLU)   -> y (NHWC)          (A^T M A)
//   wino_dy_kernel   dy (NHWC)                    -> dM [16][T][K]     (A dy A^T, the adjoint of wino_out)
//   wino_w_kernel    g [K][3][3][C]               -> U  [16][K][C]     (G g G^T) and/or U'[16][C][K] of the 180-degree
//                                                    rotated, channel-transposed filter (data gradient = the same conv)
//   wino_dw_kernel   dU [16][K][C]                -> dg [K][3][3][C]   (G^T dU G, the adjoint of wino_w), = or +=
// Backward: dx = Winograd conv of dy with U' (no overlap-add), dU[xi] = dM[xi]^T V[xi] (V kept from the forward).
#include <device_rt.h>

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 z4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

inline int ew_grid(long total) {
    long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    return (int)(g < 1 ? 1 : g);
}

// affine != nullptr: the input is the RAW output of the convolution below and the BatchNorm(+ReLU) between the two layers is applied
// on load -- d = relu?(x * scale + shift) with (scale, shift) = affine[0:C], affine[C:2C], zero outside the image like the padding of
// the normalised tensor -- so that tensor is never written or read (the expression of bn_apply_body).
__device__ __forceinline__ float4 wino_ld(const float* __restrict__ p, bool ok, bool aff, float4 sc, float4 sh, int relu) {
    if (!ok) return z4();
    float4 v = ld4(p);
    if (aff) {
        v = v * sc + sh;
        if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    }
    return v;
}

// plane_in > 0: the (point, tile, channel) array is a row range of a wider one -- the tiles of several tensors side by side (the RPN's
// shared convolution over the FPN levels, one GEMM for all of them) -- V points at this tensor's first row and the point planes are
// plane_in floats apart; 0: the array of this tensor alone (planes T * C apart).  Same convention in the kernels below.
__global__ void __launch_bounds__(256) wino_in_kernel(const float* __restrict__ x, float* __restrict__ V, int N, int H, int W,
                                                      int C, const float* __restrict__ affine, int relu, long plane_in) {
    const int C4 = C >> 2, TH = H >> 1, TW = W >> 1;
    const bool aff = affine != nullptr;
    const long T = (long)N * TH * TW, total = T * C4, plane = plane_in > 0 ? plane_in : T * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const long t = i / C4;
        const int tx = (int)(t % TW);
        const int ty = (int)((t / TW) % TH);
        const int n = (int)(t / ((long)TW * TH));
        const float4 sc = aff ? ld4(affine + 4 * c4) : z4(), sh = aff ? ld4(affine + C + 4 * c4) : z4();
        float4 d[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ih = 2 * ty - 1 + r;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int iw = 2 * tx - 1 + s;
                const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                d[r][s] = wino_ld(x + (((long)n * H + ih) * W + iw) * C + 4 * c4, ok, aff, sc, sh, relu);
            }
        }
        float4 u[4][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {       // B^T d
            u[0][s] = d[0][s] - d[2][s];
            u[1][s] = d[1][s] + d[2][s];
            u[2][s] = d[2][s] - d[1][s];
            u[3][s] = d[1][s] - d[3][s];
        }
        float* o = V + t * C + 4 * c4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {       // (.) B
            st4(o + (long)(4 * r + 0) * plane, u[r][0] - u[r][2]);
            st4(o + (long)(4 * r + 1) * plane, u[r][1] + u[r][2]);
            st4(o + (long)(4 * r + 2) * plane, u[r][2] - u[r][1]);
            st4(o + (long)(4 * r + 3) * plane, u[r][1] - u[r][3]);
        }
    }
}

// Per-channel sum / sum of squares of everything a workgroup wrote (BatchNorm statistics of the Winograd layers,
// cf. the conv_fwd epilogue in conv_gemm.hip): every thread keeps ONE channel group for its whole grid-stride walk
// (256 % K4 == 0), so it accumulates privately and the 256 / K4 threads of a group meet once in LDS at the end.
__device__ __forceinline__ void block_channel_stats(float4 sv, float4 sq, int K, float* __restrict__ stats) {
    __shared__ float4 r0[256], r1[256];
    const int K4 = K >> 2, t = threadIdx.x;
    r0[t] = sv;
    r1[t] = sq;
    __syncthreads();
    if (t < K4) {
        float4 a = r0[t], b = r1[t];
        for (int q = t + K4; q < 256; q += K4) {
            a = make_float4(a.x + r0[q].x, a.y + r0[q].y, a.z + r0[q].z, a.w + r0[q].w);
            b = make_float4(b.x + r1[q].x, b.y + r1[q].y, b.z + r1[q].z, b.w + r1[q].w);
        }
        st4(stats + (long)blockIdx.x * 2 * K + 4 * t, a);
        st4(stats + (long)blockIdx.x * 2 * K + K + 4 * t, b);
    }
}
#define OMNI_ACC_STATS(v) do { sv = make_float4(sv.x + (v).x, sv.y + (v).y, sv.z + (v).z, sv.w + (v).w); \
                               sq = make_float4(sq.x + (v).x * (v).x, sq.y + (v).y * (v).y, sq.z + (v).z * (v).z, sq.w + (v).w * (v).w); } while (0)

__global__ void __launch_bounds__(256) wino_out_kernel(const float* __restrict__ M, const float* __restrict__ bias,
                                                       float* __restrict__ y, int N, int H, int W, int K, int relu,
                                                       float* __restrict__ stats, const float* __restrict__ carry, long ldc, long plane_in) {
    const int K4 = K >> 2, TH = H >> 1, TW = W >> 1;
    const long T = (long)N * TH * TW, total = T * K4, plane = plane_in > 0 ? plane_in : T * K;
    float4 sv = z4(), sq = z4();
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k4 = (int)(i % K4);
        const long t = i / K4;
        const int tx = (int)(t % TW);
        const int ty = (int)((t / TW) % TH);
        const int n = (int)(t / ((long)TW * TH));
        const float* m = M + t * K + 4 * k4;
        float4 s[2][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {       // A^T M
            const float4 m0 = ld4(m + (long)(0 + c) * plane), m1 = ld4(m + (long)(4 + c) * plane);
            const float4 m2 = ld4(m + (long)(8 + c) * plane), m3 = ld4(m + (long)(12 + c) * plane);
            s[0][c] = m0 + m1 + m2;
            s[1][c] = m1 - m2 - m3;
        }
        const float4 b = bias != nullptr ? ld4(bias + 4 * k4) : z4();
#pragma unroll
        for (int r = 0; r < 2; ++r) {       // (.) A
            float4 y0 = s[r][0] + s[r][1] + s[r][2] + b;
            float4 y1 = s[r][1] - s[r][2] - s[r][3] + b;
            if (relu) {
                y0 = make_float4(fmaxf(y0.x, 0.f), fmaxf(y0.y, 0.f), fmaxf(y0.z, 0.f), fmaxf(y0.w, 0.f));
                y1 = make_float4(fmaxf(y1.x, 0.f), fmaxf(y1.y, 0.f), fmaxf(y1.z, 0.f), fmaxf(y1.w, 0.f));
            }
            const long pix = ((long)n * H + 2 * ty + r) * W + 2 * tx;
            if (carry != nullptr) {     // gradient fan-in: what other consumers of this tensor already contributed (pixel pitch ldc)
                y0 = y0 + ld4(carry + pix * ldc + 4 * k4);
                y1 = y1 + ld4(carry + (pix + 1) * ldc + 4 * k4);
            }
            float* o = y + pix * K + 4 * k4;
            st4(o, y0);
            st4(o + K, y1);
            OMNI_ACC_STATS(y0);
            OMNI_ACC_STATS(y1);
        }
    }
    if (stats != nullptr) block_channel_stats(sv, sq, K, stats);
}

__global__ void __launch_bounds__(256) wino_dy_kernel(const float* __restrict__ dy, float* __restrict__ dM, int N, int H, int W,
                                                      int K) {
    const int K4 = K >> 2, TH = H >> 1, TW = W >> 1;
    const long T = (long)N * TH * TW, total = T * K4, plane = T * K;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k4 = (int)(i % K4);
        const long t = i / K4;
        const int tx = (int)(t % TW);
        const int ty = (int)((t / TW) % TH);
        const int n = (int)(t / ((long)TW * TH));
        const float* g = dy + (((long)n * H + 2 * ty) * W + 2 * tx) * K + 4 * k4;
        const float4 g00 = ld4(g), g01 = ld4(g + K), g10 = ld4(g + (long)W * K), g11 = ld4(g + (long)W * K + K);
        // u = A dy  (rows: dy0, dy0 + dy1, dy0 - dy1, -dy1)
        float4 u[4][2];
        u[0][0] = g00;        u[0][1] = g01;
        u[1][0] = g00 + g10;  u[1][1] = g01 + g11;
        u[2][0] = g00 - g10;  u[2][1] = g01 - g11;
        u[3][0] = z4() - g10; u[3][1] = z4() - g11;
        float* o = dM + t * K + 4 * k4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {       // (.) A^T
            st4(o + (long)(4 * r + 0) * plane, u[r][0]);
            st4(o + (long)(4 * r + 1) * plane, u[r][0] + u[r][1]);
            st4(o + (long)(4 * r + 2) * plane, u[r][0] - u[r][1]);
            st4(o + (long)(4 * r + 3) * plane, z4() - u[r][1]);
        }
    }
}

// U[xi][k][c] = (G g G^T)[xi]; optionally also U'[xi][c][k], the transform of the 180-degree rotated, channel-transposed filter
// (what the data gradient convolves dy with): rotating g permutes the rows of G g by pi = (3, 1, 2, 0), so
// U'[4i + j][c][k] = U[4 pi(i) + pi(j)][k][c] -- no second pass over g.
// 32 x 32 (k, c) tiles (round 4; see wino4_w_body32): a thread owns four (k, c) pairs, k = k0 + tid / 32 + 8 j; U' through LDS four
// points at a time ([4][32][33] floats)
__device__ __forceinline__ void wino_w_body32(const float* __restrict__ g, float* __restrict__ U, float* __restrict__ Uf, int K, int C,
                                              int bid, float* __restrict__ smem) {
    float (*s_t)[32][33] = reinterpret_cast<float (*)[32][33]>(smem);
    const long total = (long)K * C;
    const int tiles_c = (C + 31) / 32;
    const int k0 = (bid / tiles_c) * 32, c0 = (bid % tiles_c) * 32;
    const int kq = threadIdx.x >> 5, cc = threadIdx.x & 31;
    const int c = c0 + cc;
    float u[4][4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = k0 + kq + 8 * j;
        const bool ok = k < K && c < C;
        float w[3][3], a[4][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) w[r][s] = ok ? g[((long)k * 9 + r * 3 + s) * C + c] : 0.f;
#pragma unroll
        for (int s = 0; s < 3; ++s) {       // G g
            a[0][s] = w[0][s];
            a[1][s] = 0.5f * (w[0][s] + w[1][s] + w[2][s]);
            a[2][s] = 0.5f * (w[0][s] - w[1][s] + w[2][s]);
            a[3][s] = w[2][s];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {       // (.) G^T
            u[j][r][0] = a[r][0];
            u[j][r][1] = 0.5f * (a[r][0] + a[r][1] + a[r][2]);
            u[j][r][2] = 0.5f * (a[r][0] - a[r][1] + a[r][2]);
            u[j][r][3] = a[r][2];
        }
        if (U != nullptr && ok) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < 4; ++t) U[(long)(4 * r + t) * total + (long)k * C + c] = u[j][r][t];
        }
    }
    if (Uf != nullptr) {
        const int okk = k0 + cc;                     // transposed side: lanes run along k
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int pr = (r == 0) ? 3 : (r == 3) ? 0 : r;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int pt = (t == 0) ? 3 : (t == 3) ? 0 : t;
                    s_t[t][cc][kq + 8 * j] = u[j][pr][pt];
                }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ol = kq + 8 * j, oc = c0 + ol;
                if (oc < C && okk < K) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) Uf[(long)(4 * r + t) * total + (long)oc * K + okk] = s_t[t][ol][cc];
                }
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ void wino_w_body(const float* __restrict__ g, float* __restrict__ U, float* __restrict__ Uf, int K, int C,
                                            int bid, float (*s_t)[16][17]) {
    // one 16 (k) x 16 (c) tile per block (bid); U' goes through an LDS transpose (s_t: [16][16][17]) so that its rows (k contiguous)
    // are written in 64-byte runs instead of 4-byte scatters
    const long total = (long)K * C;
    const int tiles_c = (C + 15) / 16;
    const int k0 = (bid / tiles_c) * 16, c0 = (bid % tiles_c) * 16;
    const int kk = threadIdx.x >> 4, cc = threadIdx.x & 15;
    const int k = k0 + kk, c = c0 + cc;
    const bool ok = k < K && c < C;
    float w[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) w[r][s] = ok ? g[((long)k * 9 + r * 3 + s) * C + c] : 0.f;
    float a[4][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {       // G g
        a[0][s] = w[0][s];
        a[1][s] = 0.5f * (w[0][s] + w[1][s] + w[2][s]);
        a[2][s] = 0.5f * (w[0][s] - w[1][s] + w[2][s]);
        a[3][s] = w[2][s];
    }
    float u[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {       // (.) G^T
        u[r][0] = a[r][0];
        u[r][1] = 0.5f * (a[r][0] + a[r][1] + a[r][2]);
        u[r][2] = 0.5f * (a[r][0] - a[r][1] + a[r][2]);
        u[r][3] = a[r][2];
    }
    if (U != nullptr && ok) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < 4; ++t) U[(long)(4 * r + t) * total + (long)k * C + c] = u[r][t];
    }
    if (Uf != nullptr) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int pr = (r == 0) ? 3 : (r == 3) ? 0 : r, pt = (t == 0) ? 3 : (t == 3) ? 0 : t;
                s_t[4 * r + t][cc][kk] = u[pr][pt];
            }
        __syncthreads();
        const int oc = c0 + kk, okk = k0 + cc;       // this thread now writes (c = c0 + kk, k = k0 + cc)
        if (oc < C && okk < K) {
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) Uf[(long)xi * total + (long)oc * K + okk] = s_t[xi][kk][cc];
        }
    }
}
__global__ void __launch_bounds__(256) wino_w_kernel(const float* __restrict__ g, float* __restrict__ U, float* __restrict__ Uf,
                                                     int K, int C, int tile32) {
    __shared__ float s_t[16][16][17];           // 17.4 KB >= the 32-tile form's [4][32][33]
    if (tile32) wino_w_body32(g, U, Uf, K, C, (int)blockIdx.x, &s_t[0][0][0]);
    else wino_w_body(g, U, Uf, K, C, (int)blockIdx.x, s_t);
}

// v[r][s] = (G^T dU G)[r][s] of filter element i = k * C + c
__device__ __forceinline__ void wino_dw_elem(const float* __restrict__ dU, long total, long i, float (&v)[3][3]) {
#pragma clang fp contract(off)
    float u[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int s = 0; s < 4; ++s) u[r][s] = dU[(long)(4 * r + s) * total + i];
    float e[3][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {       // G^T dU
        e[0][s] = u[0][s] + 0.5f * (u[1][s] + u[2][s]);
        e[1][s] = 0.5f * (u[1][s] - u[2][s]);
        e[2][s] = 0.5f * (u[1][s] + u[2][s]) + u[3][s];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {       // (.) G
        v[r][0] = e[r][0] + 0.5f * (e[r][1] + e[r][2]);
        v[r][1] = 0.5f * (e[r][1] - e[r][2]);
        v[r][2] = 0.5f * (e[r][1] + e[r][2]) + e[r][3];
    }
}

__global__ void __launch_bounds__(256) wino_dw_kernel(const float* __restrict__ dU, float* __restrict__ dg, int K, int C,
                                                      int accumulate) {
#pragma clang fp contract(off)
    const long total = (long)K * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C), k = (int)(i / C);
        float v[3][3];
        wino_dw_elem(dU, total, i, v);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float* o = dg + ((long)k * 9 + r * 3) * C + c;
            if (accumulate) { o[0] += v[r][0]; o[C] += v[r][1]; o[2 * (long)C] += v[r][2]; }
            else { o[0] = v[r][0]; o[C] = v[r][1]; o[2 * (long)C] = v[r][2]; }
        }
    }
}

// ================================================================================================================
// F(4x4, 3x3): 6x6 input tiles (stride 4), 36 Winograd points, 2.25 multiplies per output instead of 4 (and 9 direct);
// the transformed tensors are only 2.25x (not 4x) the size of the activations.
//
// Interpolation points (round 3): {0, 1, -1, 1/2, -2, inf} instead of Lavin & Gray's {0, 1, -1, 2, -2, inf}.  The symmetric
// +-2 pair makes the transforms mix magnitudes up to 25x (B^T rows like [4 0 -5 0 1 0]); replacing +2 by +1/2 keeps every
// entry of B^T within 5/2 and of G within 16/15.  Measured in fp32 against float64 (256-channel reduction, N(0,1) data,
// tools/winograd_points.py): forward / data gradient rms error 3.1e-6 -> 2.0e-6 of the output rms (a direct fp32 convolution:
// 0.9e-6), weight gradient over 1024 tiles 4.2e-6 -> 3.3e-6 (direct: 1.8e-6).  Matrices (exact, Cook-Toom with the Lagrange
// denominators folded into G):
//   B^T = [1 -3/2 -2 3/2 1 0; 0 -1 1/2 5/2 1 0; 0 1 -5/2 1/2 1 0; 0 -2 -1 2 1 0; 0 1/2 -1 -1/2 1 0; 0 1 -3/2 -2 3/2 1]
//   G   = [1 0 0; 1/3 1/3 1/3; -1/3 1/3 -1/3; -16/15 -8/15 -4/15; 1/15 -2/15 4/15; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 1/2 -2 0; 0 1 1 1/4 4 0; 0 1 -1 1/8 -8 1]
// Layers with < 256 tiles of 4x4 or extents not divisible by 4 keep F(2x2, 3x3).
// ================================================================================================================
template <typename V>
__device__ __forceinline__ void bt6(const V (&d)[6], V (&t)[6]) {
    t[0] = d[0] + (d[3] - d[1]) * 1.5f - d[2] * 2.f + d[4];
    t[1] = d[2] * 0.5f + d[3] * 2.5f - d[1] + d[4];
    t[2] = d[1] - d[2] * 2.5f + d[3] * 0.5f + d[4];
    t[3] = (d[3] - d[1]) * 2.f - d[2] + d[4];
    t[4] = (d[1] - d[3]) * 0.5f - d[2] + d[4];
    t[5] = d[1] + (d[4] - d[2]) * 1.5f - d[3] * 2.f + d[5];
}
template <typename V>
__device__ __forceinline__ void at6(const V (&m)[6], V (&y)[4]) {
    y[0] = m[0] + m[1] + m[2] + m[3] + m[4];
    y[1] = m[1] - m[2] + m[3] * 0.5f - m[4] * 2.f;
    y[2] = m[1] + m[2] + m[3] * 0.25f + m[4] * 4.f;
    y[3] = m[1] - m[2] + m[3] * 0.125f - m[4] * 8.f + m[5];
}
template <typename V>
__device__ __forceinline__ void a6(const V (&y)[4], V (&u)[6]) {      // u = A y (adjoint of at6)
    u[0] = y[0];
    u[1] = y[0] + y[1] + y[2] + y[3];
    u[2] = y[0] - y[1] + y[2] - y[3];
    u[3] = y[0] + y[1] * 0.5f + y[2] * 0.25f + y[3] * 0.125f;
    u[4] = y[0] - y[1] * 2.f + y[2] * 4.f - y[3] * 8.f;
    u[5] = y[3];
}

__global__ void __launch_bounds__(256) wino4_in_kernel(const float* __restrict__ x, float* __restrict__ V, int N, int H, int W,
                                                       int C, const float* __restrict__ affine, int relu, long plane_in) {
    const int C4 = C >> 2, TH = H >> 2, TW = W >> 2;
    const bool aff = affine != nullptr;
    const long T = (long)N * TH * TW, total = T * C4, plane = plane_in > 0 ? plane_in : T * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const long t = i / C4;
        const int tx = (int)(t % TW);
        const int ty = (int)((t / TW) % TH);
        const int n = (int)(t / ((long)TW * TH));
        const float4 sc = aff ? ld4(affine + 4 * c4) : z4(), sh = aff ? ld4(affine + C + 4 * c4) : z4();
        float4 u[6][6];     // u[r][s] = (B^T d) row r, column s -- built column by column
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            const int iw = 4 * tx - 1 + s;
            float4 d[6], tcol[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const int ih = 4 * ty - 1 + r;
                const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                d[r] = wino_ld(x + (((long)n * H + ih) * W + iw) * C + 4 * c4, ok, aff, sc, sh, relu);
            }
            bt6(d, tcol);
#pragma unroll
            for (int r = 0; r < 6; ++r) u[r][s] = tcol[r];
        }
        float* o = V + t * C + 4 * c4;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            float4 v[6];
            bt6(u[r], v);       // (.) B  ==  B^T applied along the row
#pragma unroll
            for (int s = 0; s < 6; ++s) st4(o + (long)(6 * r + s) * plane, v[s]);
        }
    }
}

// BatchNorm-BACKWARD statistics of the layer below a data-gradient transform (bnb.x != nullptr): the tensor written here is the
// gradient dy of a BatchNorm(+ReLU) output, and that layer's backward pass starts with the per-channel sums of dz and dz * xhat
// (dz = dy masked by the ReLU, xhat = (x - mean) * rstd of the BatchNorm INPUT x) -- bn_reduce_kernel<1>, one more pass over dy
// and x.  Here every dy value is in a register already: the thread reads x at the same offset and accumulates both sums, one
// partial row per workgroup like the forward statistics.
struct BnBwdStats {
    const float* x;             // BatchNorm input, same (N, H, W, K) layout as the tensor written here
    const float* mean_rstd;     // 2K
    const float* scale_shift;   // 2K, or nullptr: no ReLU on that layer
};
__global__ void __launch_bounds__(256) wino4_out_kernel(const float* __restrict__ M, const float* __restrict__ bias,
                                                        float* __restrict__ y, int N, int H, int W, int K, int relu,
                                                        float* __restrict__ stats, BnBwdStats bnb,
                                                        const float* __restrict__ carry, long ldc, long plane_in) {
    const int K4 = K >> 2, TH = H >> 2, TW = W >> 2;
    const long T = (long)N * TH * TW, total = T * K4, plane = plane_in > 0 ? plane_in : T * K;
    float4 sv = z4(), sq = z4();
    float4 mu = z4(), rs = z4(), sc = z4(), sh = z4();
    if (bnb.x != nullptr) {            // (the statistics modes keep ONE channel group per thread: 256 % K4 == 0)
        const int kc = 4 * (int)(threadIdx.x % K4);
        mu = ld4(bnb.mean_rstd + kc); rs = ld4(bnb.mean_rstd + K + kc);
        if (bnb.scale_shift != nullptr) { sc = ld4(bnb.scale_shift + kc); sh = ld4(bnb.scale_shift + K + kc); }
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k4 = (int)(i % K4);
        const long t = i / K4;
        const int tx = (int)(t % TW);
        const int ty = (int)((t / TW) % TH);
        const int n = (int)(t / ((long)TW * TH));
        const float* m = M + t * K + 4 * k4;
        float4 s[4][6];     // A^T M
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            float4 col[6], o4[4];
#pragma unroll
            for (int r = 0; r < 6; ++r) col[r] = ld4(m + (long)(6 * r + c) * plane);
            at6(col, o4);
#pragma unroll
            for (int r = 0; r < 4; ++r) s[r][c] = o4[r];
        }
        const float4 b = bias != nullptr ? ld4(bias + 4 * k4) : z4();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float4 row[4];
            at6(s[r], row);
            const long pix = ((long)n * H + 4 * ty + r) * W + 4 * tx;
            const long off = pix * K + 4 * k4;
            float* o = y + off;
            if (carry != nullptr) {     // gradient fan-in (see wino_out_kernel); never combined with bias / ReLU / statistics
                float4 cv[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) cv[c] = ld4(carry + (pix + c) * ldc + 4 * k4);
#pragma unroll
                for (int c = 0; c < 4; ++c) st4(o + (long)c * K, row[c] + cv[c]);
            } else if (bnb.x != nullptr) {
                float4 xv[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) xv[c] = ld4(bnb.x + off + (long)c * K);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float4 v = row[c];
                    st4(o + (long)c * K, v);
                    float4 g = v;
                    if (bnb.scale_shift != nullptr) {      // y > 0 of that layer, recomputed like bn_apply_body computes it
                        const float4 yv = xv[c] * sc + sh;
                        g = make_float4(yv.x > 0.f ? v.x : 0.f, yv.y > 0.f ? v.y : 0.f, yv.z > 0.f ? v.z : 0.f, yv.w > 0.f ? v.w : 0.f);
                    }
                    const float4 xh = (xv[c] - mu) * rs;
                    sv = sv + g;
                    sq = sq + g * xh;
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float4 v = row[c] + b;
                    if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                    st4(o + (long)c * K, v);
                    OMNI_ACC_STATS(v);
                }
            }
        }
    }
    if (stats != nullptr) block_channel_stats(sv, sq, K, stats);
}

__global__ void __launch_bounds__(256) wino4_dy_kernel(const float* __restrict__ dy, float* __restrict__ dM, int N, int H, int W,
                                                       int K) {
    const int K4 = K >> 2, TH = H >> 2, TW = W >> 2;
    const long T = (long)N * TH * TW, total = T * K4, plane = T * K;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k4 = (int)(i % K4);
        const long t = i / K4;
        const int tx = (int)(t % TW);
        const int ty = (int)((t / TW) % TH);
        const int n = (int)(t / ((long)TW * TH));
        const float* g = dy + (((long)n * H + 4 * ty) * W + 4 * tx) * K + 4 * k4;
        float4 u[6][4];     // A dy
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float4 col[4], o6[6];
#pragma unroll
            for (int r = 0; r < 4; ++r) col[r] = ld4(g + ((long)r * W + c) * K);
            a6(col, o6);
#pragma unroll
            for (int r = 0; r < 6; ++r) u[r][c] = o6[r];
        }
        float* o = dM + t * K + 4 * k4;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            float4 row[6];
            a6(u[r], row);      // (.) A^T
#pragma unroll
            for (int c = 0; c < 6; ++c) st4(o + (long)(6 * r + c) * plane, row[c]);
        }
    }
}

__device__ __forceinline__ void g6(const float (&g)[3], float (&a)[6]) {
    a[0] = g[0];
    a[1] = (g[0] + g[1] + g[2]) * (1.f / 3.f);
    a[2] = (g[1] - g[0] - g[2]) * (1.f / 3.f);
    a[3] = -(g[0] * 4.f + g[1] * 2.f + g[2]) * (4.f / 15.f);
    a[4] = (g[0] - g[1] * 2.f + g[2] * 4.f) * (1.f / 15.f);
    a[5] = g[2];
}
__device__ __forceinline__ void gt6(const float (&u)[6], float (&e)[3]) {      // e = G^T u (adjoint of g6)
#pragma clang fp contract(off)      // one rounding per written operation: the chained form (wino_dw_multi_kernel) must equal the separate launches bit for bit
    e[0] = u[0] + (u[1] - u[2]) * (1.f / 3.f) - u[3] * (16.f / 15.f) + u[4] * (1.f / 15.f);
    e[1] = (u[1] + u[2]) * (1.f / 3.f) - u[3] * (8.f / 15.f) - u[4] * (2.f / 15.f);
    e[2] = (u[1] - u[2]) * (1.f / 3.f) + (u[4] - u[3]) * (4.f / 15.f) + u[5];
}

// U[36][K][C] and / or U'[36][C][K] (rotated, channel-transposed filter); 16 x 16 (k, c) tile per block, U' through an LDS
// transpose (as wino_w_kernel)
__device__ __forceinline__ void wino4_w_body(const float* __restrict__ g, float* __restrict__ U, float* __restrict__ Uf, int K, int C,
                                             int bid, float (*s_t)[16][17]) {
    const long total = (long)K * C;
    const int tiles_c = (C + 15) / 16;
    const int k0 = (bid / tiles_c) * 16, c0 = (bid % tiles_c) * 16;
    const int kk = threadIdx.x >> 4, cc = threadIdx.x & 15;
    const int k = k0 + kk, c = c0 + cc;
    const bool ok = k < K && c < C;
    float w[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) w[r][s] = ok ? g[((long)k * 9 + r * 3 + s) * C + c] : 0.f;
#pragma unroll
    for (int flip = 0; flip < 2; ++flip) {
        float* dst = flip ? Uf : U;
        if (dst == nullptr) continue;            // uniform
        float a[6][3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            float col[3], o6[6];
#pragma unroll
            for (int r = 0; r < 3; ++r) col[r] = flip ? w[2 - r][2 - s] : w[r][s];
            g6(col, o6);
#pragma unroll
            for (int r = 0; r < 6; ++r) a[r][s] = o6[r];
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            float row[6];
            g6(a[r], row);
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                if (!flip) { if (ok) dst[(long)(6 * r + s) * total + (long)k * C + c] = row[s]; }
                else s_t[6 * r + s][cc][kk] = row[s];
            }
        }
        if (flip) {
            __syncthreads();
            const int oc = c0 + kk, okk = k0 + cc;
            if (oc < C && okk < K) {
#pragma unroll
                for (int xi = 0; xi < 36; ++xi) dst[(long)xi * total + (long)oc * K + okk] = s_t[xi][kk][cc];
            }
        }
    }
}
// The same transform on 32 x 32 (k, c) tiles (round 4): every store instruction of a wave covers two 128-byte runs instead of four
// 64-byte ones -- this launch writes 260 MB per training step (36 planes of U and of U' for every 3x3 filter) and ran at 2.3 TB/s.
// A thread owns four (k, c) pairs, k = k0 + tid / 32 + 8 j; U' goes through LDS one row of six points at a time ([6][32][33] floats).
__device__ __forceinline__ void wino4_w_body32(const float* __restrict__ g, float* __restrict__ U, float* __restrict__ Uf, int K, int C,
                                               int bid, float* __restrict__ smem) {
    float (*s_t)[32][33] = reinterpret_cast<float (*)[32][33]>(smem);
    const long total = (long)K * C;
    const int tiles_c = (C + 31) / 32;
    const int k0 = (bid / tiles_c) * 32, c0 = (bid % tiles_c) * 32;
    const int kq = threadIdx.x >> 5, cc = threadIdx.x & 31;
    const int c = c0 + cc;
    float w[4][3][3];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = k0 + kq + 8 * j;
        const bool ok = k < K && c < C;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) w[j][r][s] = ok ? g[((long)k * 9 + r * 3 + s) * C + c] : 0.f;
    }
    if (U != nullptr) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + kq + 8 * j;
            const bool ok = k < K && c < C;
            float a[6][3];
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                float col[3], o6[6];
#pragma unroll
                for (int r = 0; r < 3; ++r) col[r] = w[j][r][s];
                g6(col, o6);
#pragma unroll
                for (int r = 0; r < 6; ++r) a[r][s] = o6[r];
            }
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                float row[6];
                g6(a[r], row);
#pragma unroll
                for (int s = 0; s < 6; ++s)
                    if (ok) U[(long)(6 * r + s) * total + (long)k * C + c] = row[s];
            }
        }
    }
    if (Uf != nullptr) {
        float a[4][6][3];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                float col[3], o6[6];
#pragma unroll
                for (int r = 0; r < 3; ++r) col[r] = w[j][2 - r][2 - s];
                g6(col, o6);
#pragma unroll
                for (int r = 0; r < 6; ++r) a[j][r][s] = o6[r];
            }
        const int okk = k0 + cc;                     // transposed side: lanes run along k
#pragma unroll
        for (int r = 0; r < 6; ++r) {                // (unrolled: `a` stays in registers)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float row[6];
                float ar[3] = {a[j][r][0], a[j][r][1], a[j][r][2]};
                g6(ar, row);
#pragma unroll
                for (int s = 0; s < 6; ++s) s_t[s][cc][kq + 8 * j] = row[s];
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ol = kq + 8 * j, oc = c0 + ol;
                if (oc < C && okk < K) {
#pragma unroll
                    for (int s = 0; s < 6; ++s) Uf[(long)(6 * r + s) * total + (long)oc * K + okk] = s_t[s][ol][cc];
                }
            }
            __syncthreads();
        }
    }
}
// OMNI_WINO_W_TILE32=0: the 16 x 16 tiles of rounds 1-3 (A/B knob)
static inline int wino4_w_tile32() {
    static const int on = [] { const char* e = getenv("OMNI_WINO_W_TILE32"); return (e == nullptr || atoi(e) != 0) ? 1 : 0; }();
    return on;
}
static inline int wino4_w_tiles(int K, int C) {
    return wino4_w_tile32() ? ((K + 31) / 32) * ((C + 31) / 32) : ((K + 15) / 16) * ((C + 15) / 16);
}
__global__ void __launch_bounds__(256) wino4_w_kernel(const float* __restrict__ g, float* __restrict__ U, float* __restrict__ Uf,
                                                      int K, int C, int tile32) {
    __shared__ float s_t[36][16][17];
    if (tile32) wino4_w_body32(g, U, Uf, K, C, (int)blockIdx.x, &s_t[0][0][0]);
    else wino4_w_body(g, U, Uf, K, C, (int)blockIdx.x, s_t);
}

// Every filter of a forward pass in ONE launch: the weights are fixed for the duration of a step, so the ~20 transform launches of
// the DLA-34 + FPN + RPN forward (6 us each, latency-bound, on the un-overlapped forward path) collapse into one.  A workgroup finds
// its (filter, 16 x 16 tile) by scanning the table's workgroup prefix.
constexpr int WINO_MULTI_MAX = 48;
struct WinoWTable {
    const float* g[WINO_MULTI_MAX];
    float* U[WINO_MULTI_MAX];
    float* Uf[WINO_MULTI_MAX];
    int K[WINO_MULTI_MAX], C[WINO_MULTI_MAX], tile[WINO_MULTI_MAX];
    int wg0[WINO_MULTI_MAX + 1];
    int n, tile32;
};
__global__ void __launch_bounds__(256) wino_w_multi_kernel(WinoWTable t) {
    __shared__ float s_t[36][16][17];
    int e = 0;
    while (e + 1 < t.n && (int)blockIdx.x >= t.wg0[e + 1]) ++e;
    const int bid = (int)blockIdx.x - t.wg0[e];
    if (t.tile[e] == 2 && t.tile32) wino_w_body32(t.g[e], t.U[e], t.Uf[e], t.K[e], t.C[e], bid, &s_t[0][0][0]);
    else if (t.tile[e] == 2) wino_w_body(t.g[e], t.U[e], t.Uf[e], t.K[e], t.C[e], bid, s_t);
    else if (t.tile32) wino4_w_body32(t.g[e], t.U[e], t.Uf[e], t.K[e], t.C[e], bid, &s_t[0][0][0]);
    else wino4_w_body(t.g[e], t.U[e], t.Uf[e], t.K[e], t.C[e], bid, s_t);
}

__device__ __forceinline__ void wino4_dw_elem(const float* __restrict__ dU, long total, long i, float (&v)[3][3]) {
#pragma clang fp contract(off)
    float e[3][6];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        float col[6], o3[3];
#pragma unroll
        for (int r = 0; r < 6; ++r) col[r] = dU[(long)(6 * r + s) * total + i];
        gt6(col, o3);
#pragma unroll
        for (int r = 0; r < 3; ++r) e[r][s] = o3[r];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) gt6(e[r], v[r]);
}

__global__ void __launch_bounds__(256) wino4_dw_kernel(const float* __restrict__ dU, float* __restrict__ dg, int K, int C,
                                                       int accumulate) {
#pragma clang fp contract(off)
    const long total = (long)K * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C), k = (int)(i / C);
        float v[3][3];
        wino4_dw_elem(dU, total, i, v);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float* o = dg + ((long)k * 9 + r * 3) * C + c;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                if (accumulate) o[(long)s * C] += v[r][s];
                else o[(long)s * C] = v[r][s];
            }
        }
    }
}

// The transforms back of several layers' weight gradients in one launch, each ADDED into its gradient view (the tail of the
// weight-gradient stream's batched launch, omni_gemm_batched_wgrad_multi).  Sources that add into the SAME view -- the RPN's shared
// 3x3 convolution sees one weight gradient per FPN level -- are chained inside one thread, in the order they were given, so the
// result is what the separate launches, one after the other, would have left.
constexpr int WINO_DW_MAX = 16;
struct WinoDwTable {
    const float* dU[WINO_DW_MAX];        // sources, grouped by destination
    int tile[WINO_DW_MAX];               // per source
    float* dg[WINO_DW_MAX];              // per destination
    int K[WINO_DW_MAX], C[WINO_DW_MAX], src0[WINO_DW_MAX + 1], wg0[WINO_DW_MAX + 1];
    int n;
};
__global__ void __launch_bounds__(256) wino_dw_multi_kernel(WinoDwTable t) {
#pragma clang fp contract(off)
    int e = 0;
    while (e + 1 < t.n && (int)blockIdx.x >= t.wg0[e + 1]) ++e;
    const int C = t.C[e];
    const long total = (long)t.K[e] * C;
    const long i = (long)((int)blockIdx.x - t.wg0[e]) * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C), k = (int)(i / C);
    float* o = t.dg[e] + (long)k * 9 * C + c;
    float acc[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) acc[r][s] = o[(long)(r * 3 + s) * C];
    for (int q = t.src0[e]; q < t.src0[e + 1]; ++q) {
        float v[3][3];
        if (t.tile[q] == 2) wino_dw_elem(t.dU[q], total, i, v);
        else wino4_dw_elem(t.dU[q], total, i, v);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) acc[r][s] += v[r][s];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) o[(long)(r * 3 + s) * C] = acc[r][s];
}

// ---- backward: both transforms of dy in one pass ---------------------------------------------------------------------
// The data gradient needs V_dy = B^T dy B (the (tile+2)^2 window, like wino_in) and the weight gradient needs
// dM = A dy A^T (the central tile x tile block of the same window): one kernel reads the window once and writes both.
__global__ void __launch_bounds__(256) wino_dy_in_kernel(const float* __restrict__ dy, float* __restrict__ dM, float* __restrict__ Vd,
                                                         int N, int H, int W, int K, long plane_in) {
    const int K4 = K >> 2, TH = H >> 1, TW = W >> 1;
    const long T = (long)N * TH * TW, total = T * K4, plane = plane_in > 0 ? plane_in : T * K;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k4 = (int)(i % K4);
        const long t = i / K4;
        const int tx = (int)(t % TW);
        const int ty = (int)((t / TW) % TH);
        const int n = (int)(t / ((long)TW * TH));
        float4 d[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ih = 2 * ty - 1 + r;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int iw = 2 * tx - 1 + s;
                const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                d[r][s] = ok ? ld4(dy + (((long)n * H + ih) * W + iw) * K + 4 * k4) : z4();
            }
        }
        const long o = t * K + 4 * k4;
        {   // V_dy = B^T d B
            float4 u[4][4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                u[0][s] = d[0][s] - d[2][s];
                u[1][s] = d[1][s] + d[2][s];
                u[2][s] = d[2][s] - d[1][s];
                u[3][s] = d[1][s] - d[3][s];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                st4(Vd + o + (long)(4 * r + 0) * plane, u[r][0] - u[r][2]);
                st4(Vd + o + (long)(4 * r + 1) * plane, u[r][1] + u[r][2]);
                st4(Vd + o + (long)(4 * r + 2) * plane, u[r][2] - u[r][1]);
                st4(Vd + o + (long)(4 * r + 3) * plane, u[r][1] - u[r][3]);
            }
        }
        {   // dM = A g A^T of the central 2x2 block
            const float4 g00 = d[1][1], g01 = d[1][2], g10 = d[2][1], g11 = d[2][2];
            float4 u[4][2];
            u[0][0] = g00;        u[0][1] = g01;
            u[1][0] = g00 + g10;  u[1][1] = g01 + g11;
            u[2][0] = g00 - g10;  u[2][1] = g01 - g11;
            u[3][0] = z4() - g10; u[3][1] = z4() - g11;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                st4(dM + o + (long)(4 * r + 0) * plane, u[r][0]);
                st4(dM + o + (long)(4 * r + 1) * plane, u[r][0] + u[r][1]);
                st4(dM + o + (long)(4 * r + 2) * plane, u[r][0] - u[r][1]);
                st4(dM + o + (long)(4 * r + 3) * plane, z4() - u[r][1]);
            }
        }
    }
}

__global__ void __launch_bounds__(256) wino4_dy_in_kernel(const float* __restrict__ dy, float* __restrict__ dM, float* __restrict__ Vd,
                                                          int N, int H, int W, int K, long plane_in) {
    const int K4 = K >> 2, TH = H >> 2, TW = W >> 2;
    const long T = (long)N * TH * TW, total = T * K4, plane = plane_in > 0 ? plane_in : T * K;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k4 = (int)(i % K4);
        const long t = i / K4;
        const int tx = (int)(t % TW);
        const int ty = (int)((t / TW) % TH);
        const int n = (int)(t / ((long)TW * TH));
        float4 d[6][6];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int ih = 4 * ty - 1 + r;
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const int iw = 4 * tx - 1 + s;
                const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                d[r][s] = ok ? ld4(dy + (((long)n * H + ih) * W + iw) * K + 4 * k4) : z4();
            }
        }
        const long o = t * K + 4 * k4;
        {   // dM = A g A^T of the central 4x4 block
            float4 u[6][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float4 col[4], o6[6];
#pragma unroll
                for (int r = 0; r < 4; ++r) col[r] = d[1 + r][1 + c];
                a6(col, o6);
#pragma unroll
                for (int r = 0; r < 6; ++r) u[r][c] = o6[r];
            }
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                float4 row[6];
                a6(u[r], row);
#pragma unroll
                for (int c = 0; c < 6; ++c) st4(dM + o + (long)(6 * r + c) * plane, row[c]);
            }
        }
        {   // V_dy = B^T d B
            float4 u[6][6];
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                float4 col[6], tcol[6];
#pragma unroll
                for (int r = 0; r < 6; ++r) col[r] = d[r][s];
                bt6(col, tcol);
#pragma unroll
                for (int r = 0; r < 6; ++r) u[r][s] = tcol[r];
            }
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                float4 v[6];
                bt6(u[r], v);
#pragma unroll
                for (int s = 0; s < 6; ++s) st4(Vd + o + (long)(6 * r + s) * plane, v[s]);
            }
        }
    }
}

inline bool bad(int N, int H, int W, int C) { return N < 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3); }

}  // namespace

extern "C" {

// omni_wino_in of relu?(x * scale + shift): affine = [scale (C) | shift (C)] of the BatchNorm between the convolution that wrote x and
// this one (nullable: plain omni_wino_in) -- the normalised activation is consumed without ever being stored.
static int wino_in_impl(const float* x, const float* affine, int relu, float* V, int N, int H, int W, int C, int tile, long plane, void* stream) {
    if (bad(N, H, W, C) || (tile != 2 && tile != 4) || (H % tile) || (W % tile)) return OMNI_ERR_ARG;
    const long tiles = (long)N * (H / tile) * (W / tile), total = tiles * (C / 4);
    if (plane != 0 && plane < tiles * C) return OMNI_ERR_ARG;
    if (total == 0) return OMNI_OK;
    if (tile == 2) hipLaunchKernelGGL(wino_in_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, V, N, H, W, C, affine, relu, plane);
    else hipLaunchKernelGGL(wino4_in_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, V, N, H, W, C, affine, relu, plane);
    return omni_launch_status();
}

int omni_wino_in_affine(const float* x, const float* affine, int relu, float* V, int N, int H, int W, int C, int tile, void* stream) {
    return wino_in_impl(x, affine, relu, V, N, H, W, C, tile, 0, stream);
}

// Row-range forms (round 4): the Winograd-domain array is a row range of a wider (points, rows_total, channels) array that holds the
// tiles of SEVERAL tensors side by side -- the five FPN levels under the RPN's shared 3x3 convolution -- so that ONE batched GEMM
// (omni_gemm_batched_fwd / _wgrad over rows_total rows) serves all of them.  V / Mt / dM / Vd point at the tensor's first row,
// `plane` = rows_total * channels floats between point planes.
int omni_wino_in_rows(const float* x, float* V, int N, int H, int W, int C, int tile, long long plane, void* stream) {
    if (plane <= 0) return OMNI_ERR_ARG;
    return wino_in_impl(x, nullptr, 0, V, N, H, W, C, tile, (long)plane, stream);
}

int omni_wino_in(const float* x, float* V, int N, int H, int W, int C, int tile, void* stream) {
    return omni_wino_in_affine(x, nullptr, 0, V, N, H, W, C, tile, stream);
}

static int wino_out_impl(const float* M, const float* bias, float* y, int N, int H, int W, int K, int relu, int tile, float* stats,
                         int stats_rows, int* nblk_out, void* stream, BnBwdStats bnb = BnBwdStats{nullptr, nullptr, nullptr},
                         const float* carry = nullptr, long ldc = 0, long plane = 0) {
    if (nblk_out) *nblk_out = 0;
    if (bad(N, H, W, K) || (tile != 2 && tile != 4) || (H % tile) || (W % tile)) return OMNI_ERR_ARG;
    const long tiles = (long)N * (H / tile) * (W / tile), total = tiles * (K / 4);
    if (plane != 0 && plane < tiles * K) return OMNI_ERR_ARG;
    if (total == 0) return OMNI_OK;
    int grid = ew_grid(total);
    // statistics: one partial row per workgroup; needs a fixed channel group per thread (256 % (K/4) == 0), raw outputs
    float* st_ptr = nullptr;
    if (stats != nullptr && bias == nullptr && !relu && K >= 4 && (256 % (K / 4)) == 0 && (bnb.x == nullptr || tile == 4)) {
        if (grid > stats_rows) grid = stats_rows;           // fewer, longer-running workgroups rather than no fusion
        if (grid >= 1) { st_ptr = stats; if (nblk_out) *nblk_out = grid; }
        else grid = ew_grid(total);
    }
    if (st_ptr == nullptr) bnb = BnBwdStats{nullptr, nullptr, nullptr};
    if (tile == 2) hipLaunchKernelGGL(wino_out_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, M, bias, y, N, H, W, K, relu, st_ptr,
                                      carry, ldc, plane);
    else hipLaunchKernelGGL(wino4_out_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, M, bias, y, N, H, W, K, relu, st_ptr, bnb,
                            carry, ldc, plane);
    return omni_launch_status();
}

// row-range form of omni_wino_out (bias / ReLU) and of omni_wino_out_carry (carry != NULL: no bias, no ReLU); see omni_wino_in_rows
int omni_wino_out_rows(const float* M, const float* bias, const float* carry, long long ldc, float* y, int N, int H, int W, int K, int relu,
                       int tile, long long plane, void* stream) {
    if (plane <= 0) return OMNI_ERR_ARG;
    if (carry != nullptr && (bias != nullptr || relu || ldc < K || (ldc & 3) || (((unsigned long long)carry) & 15))) return OMNI_ERR_ARG;
    return wino_out_impl(M, bias, y, N, H, W, K, relu, tile, nullptr, 0, nullptr, stream, BnBwdStats{nullptr, nullptr, nullptr}, carry,
                         carry != nullptr ? (long)ldc : 0, (long)plane);
}

int omni_wino_out(const float* M, const float* bias, float* y, int N, int H, int W, int K, int relu, int tile, void* stream) {
    return wino_out_impl(M, bias, y, N, H, W, K, relu, tile, nullptr, 0, nullptr, stream);
}

// Data-gradient output transform with gradient fan-in: y = A^T M A + carry, carry an NHWC tensor of the same extent whose pixels
// are ldc floats apart (ldc >= K, ldc % 4 == 0, 16-byte aligned: a whole tensor, or a channel slice of a wider one).  The tensor this
// convolution's input gradient is added to (another consumer's gradient of the same activation) is read here instead of by a
// separate add kernel.  carry may alias nothing that is written (y is a different buffer).
int omni_wino_out_carry(const float* M, const float* carry, long long ldc, float* y, int N, int H, int W, int K, int tile, void* stream) {
    if (carry == nullptr || ldc < K || (ldc & 3) || (((unsigned long long)carry) & 15)) return OMNI_ERR_ARG;
    return wino_out_impl(M, nullptr, y, N, H, W, K, 0, tile, nullptr, 0, nullptr, stream, BnBwdStats{nullptr, nullptr, nullptr}, carry, (long)ldc);
}

// Data-gradient output transform that also emits the BACKWARD partial statistics of the BatchNorm whose output gradient it writes
// (see BnBwdStats): stats [rows][2][K] = per-workgroup (sum dz, sum dz * xhat); *nblk_out = rows written, 0 = not produced (F(2x2)
// tiles, channel count without a fixed group per thread) and the caller runs omni_bn_bwd.  scale_shift nullable (no ReLU).
int omni_wino_out_bn_bwd_stats(const float* M, float* y, int N, int H, int W, int K, int tile, const float* bn_x, const float* mean_rstd,
                               const float* scale_shift, float* stats, int stats_rows, int* nblk_out, void* stream) {
    if (bn_x == nullptr || mean_rstd == nullptr) return OMNI_ERR_ARG;
    return wino_out_impl(M, nullptr, y, N, H, W, K, 0, tile, stats, stats_rows, nblk_out, stream, BnBwdStats{bn_x, mean_rstd, scale_shift});
}

// Output transform (no bias, no ReLU) that also emits BatchNorm partial statistics [rows][2][K]; *nblk_out = rows written (0 = none)
int omni_wino_out_stats(const float* M, float* y, int N, int H, int W, int K, int tile, float* stats, int stats_rows, int* nblk_out,
                        void* stream) {
    return wino_out_impl(M, nullptr, y, N, H, W, K, 0, tile, stats, stats_rows, nblk_out, stream);
}

int omni_wino_dy(const float* dy, float* dM, int N, int H, int W, int K, int tile, void* stream) {
    if (bad(N, H, W, K) || (tile != 2 && tile != 4) || (H % tile) || (W % tile)) return OMNI_ERR_ARG;
    const long total = (long)N * (H / tile) * (W / tile) * (K / 4);
    if (total == 0) return OMNI_OK;
    if (tile == 2) hipLaunchKernelGGL(wino_dy_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, dy, dM, N, H, W, K);
    else hipLaunchKernelGGL(wino4_dy_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, dy, dM, N, H, W, K);
    return omni_launch_status();
}

static int wino_dy_in_impl(const float* dy, float* dM, float* Vd, int N, int H, int W, int K, int tile, long plane, void* stream) {
    if (bad(N, H, W, K) || (tile != 2 && tile != 4) || (H % tile) || (W % tile)) return OMNI_ERR_ARG;
    const long tiles = (long)N * (H / tile) * (W / tile), total = tiles * (K / 4);
    if (plane != 0 && plane < tiles * K) return OMNI_ERR_ARG;
    if (total == 0) return OMNI_OK;
    if (tile == 2) hipLaunchKernelGGL(wino_dy_in_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, dy, dM, Vd, N, H, W, K, plane);
    else hipLaunchKernelGGL(wino4_dy_in_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, dy, dM, Vd, N, H, W, K, plane);
    return omni_launch_status();
}

int omni_wino_dy_in(const float* dy, float* dM, float* Vd, int N, int H, int W, int K, int tile, void* stream) {
    return wino_dy_in_impl(dy, dM, Vd, N, H, W, K, tile, 0, stream);
}

// row-range form (dM and Vd are row ranges of two arrays of the same extent; see omni_wino_in_rows)
int omni_wino_dy_in_rows(const float* dy, float* dM, float* Vd, int N, int H, int W, int K, int tile, long long plane, void* stream) {
    if (plane <= 0) return OMNI_ERR_ARG;
    return wino_dy_in_impl(dy, dM, Vd, N, H, W, K, tile, (long)plane, stream);
}

int omni_wino_weights(const float* g, float* U, float* U_flip, int K, int C, int tile, void* stream) {
    if (K <= 0 || C <= 0 || (U == nullptr && U_flip == nullptr) || (tile != 2 && tile != 4)) return OMNI_ERR_ARG;
    const unsigned wt = (unsigned)wino4_w_tiles(K, C);       // one 32 x 32 (or, OMNI_WINO_W_TILE32=0, 16 x 16) (k, c) tile per workgroup
    if (tile == 2) hipLaunchKernelGGL(wino_w_kernel, dim3(wt), dim3(256), 0, (hipStream_t)stream, g, U, U_flip, K, C, wino4_w_tile32());
    else hipLaunchKernelGGL(wino4_w_kernel, dim3(wt), dim3(256), 0, (hipStream_t)stream, g, U, U_flip, K, C, wino4_w_tile32());
    return omni_launch_status();
}

// omni_wino_weights for n (<= 48) filters in one launch: g / U / U_flip are HOST arrays of device pointers (U[i], U_flip[i] nullable, not
// both), K / C / tile HOST int arrays.
int omni_wino_weights_multi(const void* const* g, const void* const* U, const void* const* U_flip, const int* K, const int* C,
                            const int* tile, int n, void* stream) {
    if (n <= 0 || n > WINO_MULTI_MAX || g == nullptr || U == nullptr || U_flip == nullptr) return OMNI_ERR_ARG;
    WinoWTable t;
    t.n = n;
    t.tile32 = wino4_w_tile32();
    t.wg0[0] = 0;
    for (int i = 0; i < n; ++i) {
        if (K[i] <= 0 || C[i] <= 0 || (U[i] == nullptr && U_flip[i] == nullptr) || (tile[i] != 2 && tile[i] != 4) || g[i] == nullptr)
            return OMNI_ERR_ARG;
        t.g[i] = (const float*)g[i];
        t.U[i] = (float*)U[i];
        t.Uf[i] = (float*)U_flip[i];
        t.K[i] = K[i]; t.C[i] = C[i]; t.tile[i] = tile[i];
        t.wg0[i + 1] = t.wg0[i] + wino4_w_tiles(K[i], C[i]);      // (both transforms take the same tile size)
    }
    hipLaunchKernelGGL(wino_w_multi_kernel, dim3((unsigned)t.wg0[n]), dim3(256), 0, (hipStream_t)stream, t);
    return omni_launch_status();
}

int omni_wino_dweights(const float* dU, float* dg, int K, int C, int accumulate, int tile, void* stream) {
    if (K <= 0 || C <= 0 || (tile != 2 && tile != 4)) return OMNI_ERR_ARG;
    if (tile == 2) hipLaunchKernelGGL(wino_dw_kernel, dim3(ew_grid((long)K * C)), dim3(256), 0, (hipStream_t)stream, dU, dg, K, C, accumulate);
    else hipLaunchKernelGGL(wino4_dw_kernel, dim3(ew_grid((long)K * C)), dim3(256), 0, (hipStream_t)stream, dU, dg, K, C, accumulate);
    return omni_launch_status();
}

// dg[i] (K[i],3,3,C[i]) += G^T dU[i] G for n <= 16 sources in ONE launch; sources with the same dg (same K, C) are added in the order given
int omni_wino_dweights_multi(const void* const* dU, const void* const* dg, const int* K, const int* C, const int* tile, int n, void* stream) {
    if (n <= 0 || n > WINO_DW_MAX || dU == nullptr || dg == nullptr) return OMNI_ERR_ARG;
    WinoDwTable t;
    int dest[WINO_DW_MAX], nd = 0;             // dest[i] = destination entry of source i
    for (int i = 0; i < n; ++i) {
        if (K[i] <= 0 || C[i] <= 0 || (tile[i] != 2 && tile[i] != 4) || dU[i] == nullptr || dg[i] == nullptr) return OMNI_ERR_ARG;
        int d = 0;
        while (d < nd && t.dg[d] != (float*)dg[i]) ++d;
        if (d == nd) {
            t.dg[nd] = (float*)dg[i]; t.K[nd] = K[i]; t.C[nd] = C[i];
            ++nd;
        } else if (t.K[d] != K[i] || t.C[d] != C[i]) {
            return OMNI_ERR_ARG;
        }
        dest[i] = d;
    }
    t.n = nd;
    t.wg0[0] = 0;
    int q = 0;
    for (int d = 0; d < nd; ++d) {
        t.src0[d] = q;
        for (int i = 0; i < n; ++i)
            if (dest[i] == d) { t.dU[q] = (const float*)dU[i]; t.tile[q] = tile[i]; ++q; }
        const long wgs = ((long)t.K[d] * t.C[d] + 255) / 256;
        if (t.wg0[d] + wgs > 0x7fffffff) return OMNI_ERR_ARG;
        t.wg0[d + 1] = t.wg0[d] + (int)wgs;
    }
    t.src0[nd] = q;
    hipLaunchKernelGGL(wino_dw_multi_kernel, dim3((unsigned)t.wg0[nd]), dim3(256), 0, (hipStream_t)stream, t);
    return omni_launch_status();
}

}  // extern "C"
