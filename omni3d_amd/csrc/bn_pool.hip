// bn_pool.hip -- HBM-bound NHWC kernels of the DLA-34 / FPN bottom-up: training-mode BatchNorm
// (batch statistics, affine, running-stat update) fused with ReLU and the residual add, its
// backward, 2x2 max-pool, the stride-2 subsample that makes p6, the FPN nearest-2x top-down add,
// and image normalisation.
//
// Reference call sites:
//   nn.BatchNorm2d + ReLU + `out += residual`   /root/reference/cubercnn/modeling/backbone/dla.py:46-66,162-172,214,244
//   nn.MaxPool2d(2,2)                           dla.py:209      F.max_pool2d(k=1,s=2) dla.py:474
//   FPN top-down `lateral + interpolate(prev, 2.0, "nearest")`   detectron2 FPN built at dla.py:500-506
//   (img - PIXEL_MEAN) / PIXEL_STD + zero padding               GeneralizedRCNN.preprocess_image (rcnn3d.py:46)
//
// All of these move each byte once or twice and do a handful of flops per element: the roofline
// is HBM (~6.3 TB/s achievable).  Layout is NHWC fp32, every lane moves 16 B (float4 = 4
// channels), consecutive lanes walk the channel dimension first, so a wave reads 1 KiB contiguous.
#include <device_rt.h>

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 f4(float a) { return make_float4(a, a, a, a); }
// (float4 + - * are the element-wise operators of HIP's vector types)
__device__ __forceinline__ float4 relu4(float4 a) { return make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f)); }
__device__ __forceinline__ float4 mask4(float4 g, float4 y) {
    return make_float4(y.x > 0.f ? g.x : 0.f, y.y > 0.f ? g.y : 0.f, y.z > 0.f ? g.z : 0.f, y.w > 0.f ? g.w : 0.f);
}

// ---- per-channel reductions over the P = N*H*W pixels of an NHWC tensor -------------------------
// MODE 0: (sum x, sum x^2)                                  -> forward statistics
// MODE 1: (sum dz, sum dz * xhat), dz = dy masked by y > 0  -> backward reductions
// grid-stride over pixel rows; 256 threads = rows x C/4 column groups; LDS tree over rows; one
// fp64 atomic per (block, channel, quantity).
// One tile = up to 256 channel quads starting at quad `cbase` (C <= 1024 is a single tile; wider tensors, e.g. the 1152-channel
// expansions of MNASNet, take several).
template <int MODE>
__device__ __forceinline__ void bn_reduce_tile(const float* __restrict__ x, const float* __restrict__ dy,
                                               const float* __restrict__ y, const float* __restrict__ mean_rstd,
                                               int P, int C, int relu, double* __restrict__ acc, int cbase, int C4,
                                               float4* __restrict__ s0, float4* __restrict__ s1, long lddy) {
    const int rows = 256 / C4;
    const int t = threadIdx.x;
    const int col = cbase + t % C4, row = t / C4;
    const bool active = row < rows;
    float4 a0 = f4(0.f), a1 = f4(0.f);
    float4 mu = f4(0.f), rs = f4(0.f), sc = f4(0.f), sh = f4(0.f);
    if (MODE == 1 && active) { mu = ld4(mean_rstd + 4 * col); rs = ld4(mean_rstd + C + 4 * col); }
    // relu == 2: `y` holds the forward pass's (scale, shift) instead of the output -- the mask y > 0 of a layer without residual is
    // recomputed as x * scale + shift > 0 with the expression of bn_apply_body, and the output tensor is not read at all
    if (MODE == 1 && active && relu == 2) { sc = ld4(y + 4 * col); sh = ld4(y + C + 4 * col); }
    if (active) {
        const long step = (long)gridDim.x * rows;
        long p = (long)blockIdx.x * rows + row;
        // 4 pixel rows per trip: the loads of a trip are independent, so 4 (MODE 1: up to 12) are in flight per lane
        for (; p + 3 * step < P; p += 4 * step) {
            float4 xv[4], g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) xv[u] = ld4(x + (p + u * step) * C + 4 * col);
            if (MODE == 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { a0 = a0 + xv[u]; a1 = a1 + xv[u] * xv[u]; }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) g[u] = ld4(dy + (p + u * step) * lddy + 4 * col);
                if (relu == 1) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) g[u] = mask4(g[u], ld4(y + (p + u * step) * C + 4 * col));
                } else if (relu == 2) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) g[u] = mask4(g[u], xv[u] * sc + sh);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) { a0 = a0 + g[u]; a1 = a1 + g[u] * ((xv[u] - mu) * rs); }
            }
        }
        for (; p < P; p += step) {
            const long off = p * C + 4 * col;
            const float4 xv = ld4(x + off);
            if (MODE == 0) {
                a0 = a0 + xv;
                a1 = a1 + xv * xv;
            } else {
                float4 g = ld4(dy + p * lddy + 4 * col);
                if (relu == 1) g = mask4(g, ld4(y + off));
                else if (relu == 2) g = mask4(g, xv * sc + sh);
                a0 = a0 + g;
                a1 = a1 + g * ((xv - mu) * rs);
            }
        }
    }
    s0[t] = a0;
    s1[t] = a1;
    __syncthreads();
    if (t < C4) {
        float4 r0 = s0[t], r1 = s1[t];
        for (int r = 1; r < rows; ++r) { r0 = r0 + s0[r * C4 + t]; r1 = r1 + s1[r * C4 + t]; }
        // per-block partials (no atomics: deterministic, and 1024 blocks hammering 2C addresses was the
        // bottleneck of this kernel); reduced over blocks by bn_partial_sum_kernel
        float* a = reinterpret_cast<float*>(acc) + (long)blockIdx.x * 2 * C;
        st4(a + 4 * (cbase + t), r0);
        st4(a + C + 4 * (cbase + t), r1);
    }
}
template <int MODE>
__device__ __forceinline__ void bn_reduce_body(const float* __restrict__ x, const float* __restrict__ dy,
                                               const float* __restrict__ y, const float* __restrict__ mean_rstd,
                                               int P, int C, int relu, double* __restrict__ acc, long lddy = 0) {
    __shared__ float4 s0[256], s1[256];
    if (lddy == 0) lddy = C;        // (dy: pixel pitch lddy floats -- a channel slice of a wider NHWC gradient is read in place)
    const int C4 = C >> 2;
    for (int cbase = 0; cbase < C4; cbase += 256) {
        const int width = C4 - cbase < 256 ? C4 - cbase : 256;
        bn_reduce_tile<MODE>(x, dy, y, mean_rstd, P, C, relu, acc, cbase, width, s0, s1, lddy);
        if (cbase + 256 < C4) __syncthreads();
    }
}
template <int MODE>
__global__ void __launch_bounds__(256) bn_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        const float* __restrict__ y, const float* __restrict__ mean_rstd,
                                                        int P, int C, int relu, double* __restrict__ acc, long lddy) {
    bn_reduce_body<MODE>(x, dy, y, mean_rstd, P, C, relu, acc, lddy);
}

// Sum the per-block partials of FIN_C channels (both quantities) in fp64: FIN_C column lanes x (256 / FIN_C) row groups + LDS
// reduction.  Returns (for threads t < FIN_C, channel c0 + t) s0 = sum partial[b][c], s1 = sum partial[b][C + c].
// Round 3: FIN_C 16 -> 4.  A finalize launch is pure latency on the critical path (56 of them per training step, 0.45 ms): with
// 16 channels per workgroup a 128-channel layer ran on 8 workgroups whose threads each walked nblk / 16 partial rows (up to 32
// dependent trips); with 4 it runs on 32 workgroups and a thread walks nblk / 64 rows.
constexpr int FIN_C = 4, FIN_RG = 256 / FIN_C;
__device__ __forceinline__ void colsum_fin(const float* __restrict__ partial, int nblk, int C, int c0, double& s0, double& s1) {
    __shared__ double sm0[256], sm1[256];
    const int t = threadIdx.x, cl = t % FIN_C, rg = t / FIN_C;
    const int c = c0 + cl;
    double a = 0.0, b = 0.0;
    if (c < C) {
        // 4 independent loads in flight per quantity: this loop is pure memory latency
        int blk = rg;
        for (; blk + 3 * FIN_RG < nblk; blk += 4 * FIN_RG) {
            const float* q = partial + (long)blk * 2 * C + c;
            const long st = (long)FIN_RG * 2 * C;
            const float a0 = q[0], a1 = q[st], a2 = q[2 * st], a3 = q[3 * st];
            const float b0 = q[C], b1 = q[st + C], b2 = q[2 * st + C], b3 = q[3 * st + C];
            a += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
            b += ((double)b0 + (double)b1) + ((double)b2 + (double)b3);
        }
        for (; blk < nblk; blk += FIN_RG) {
            a += (double)partial[(long)blk * 2 * C + c];
            b += (double)partial[(long)blk * 2 * C + C + c];
        }
    }
    sm0[t] = a;
    sm1[t] = b;
    __syncthreads();
    // tree over the row groups (fixed order: deterministic)
    for (int half = FIN_RG / 2; half >= 1; half >>= 1) {
        if (rg < half) { sm0[t] += sm0[t + half * FIN_C]; sm1[t] += sm1[t + half * FIN_C]; }
        __syncthreads();
    }
    s0 = sm0[t % FIN_C];
    s1 = sm1[t % FIN_C];
}

// forward finalize (one workgroup per FIN_C channels): mean, biased var -> rstd; scale/shift for the apply pass;
// running stats (momentum m, unbiased variance) exactly like torch.nn.functional.batch_norm(training=True).
__device__ __forceinline__ void bn_finalize_fwd_group(int grp, const float* __restrict__ partial, int nblk, int P, int C,
                                                      float eps, float momentum, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* __restrict__ mean_rstd,
                                                      float* __restrict__ scale_shift, float* __restrict__ running_mean,
                                                      float* __restrict__ running_var) {
    double sx, sxx;
    colsum_fin(partial, nblk, C, grp * FIN_C, sx, sxx);
    const int c = grp * FIN_C + threadIdx.x;
    if (threadIdx.x >= FIN_C || c >= C) return;
    const double mean = sx / P;
    double var = sxx / P - mean * mean;
    if (var < 0) var = 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    mean_rstd[c] = (float)mean;
    mean_rstd[C + c] = rstd;
    const float sc = gamma[c] * rstd;
    scale_shift[c] = sc;
    scale_shift[C + c] = beta[c] - (float)mean * sc;
    if (running_mean != nullptr) {
        const double unbiased = P > 1 ? var * P / (P - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}
__global__ void __launch_bounds__(256) bn_finalize_fwd_kernel(const float* __restrict__ partial, int nblk, int P, int C,
                                                              float eps, float momentum, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ mean_rstd,
                                                              float* __restrict__ scale_shift, float* __restrict__ running_mean,
                                                              float* __restrict__ running_var) {
    bn_finalize_fwd_group(blockIdx.x, partial, nblk, P, C, eps, momentum, gamma, beta, mean_rstd, scale_shift, running_mean, running_var);
}

// y = relu?( x * scale + shift (+ residual) )
// (ldy: pixel pitch of y in floats, 0 = dense; a Root child is written straight into its channel slice of the concatenated tensor)
__device__ __forceinline__ void bn_apply_body(const float* __restrict__ x, const float* __restrict__ scale_shift,
                                              const float* __restrict__ residual, float* __restrict__ y,
                                              long total4, int C, int relu, long ldy = 0) {
    const int C4 = C >> 2;
    const bool dense = ldy == 0 || ldy == C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const int col = (int)(i % C4);
        float4 v = ld4(x + 4 * i) * ld4(scale_shift + 4 * col) + ld4(scale_shift + C + 4 * col);
        if (residual != nullptr) v = v + ld4(residual + 4 * i);
        if (relu) v = relu4(v);
        st4(dense ? y + 4 * i : y + (i / C4) * ldy + 4 * col, v);
    }
}
__global__ void __launch_bounds__(256) bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale_shift,
                                                       const float* __restrict__ residual, float* __restrict__ y,
                                                       long total4, int C, int relu, long ldy) {
    bn_apply_body(x, scale_shift, residual, y, total4, C, relu, ldy);
}

// backward finalize: dgamma, dbeta out (or accumulated); coefficients for the apply pass:
//   dx = g_rstd * (dz - a - xhat * b),  g_rstd = gamma*rstd, a = dbeta/P, b = dgamma/P
__device__ __forceinline__ void bn_finalize_bwd_group(int grp, const float* __restrict__ partial, int nblk, int P, int C,
                                                      const float* __restrict__ gamma, const float* __restrict__ mean_rstd,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                      float* __restrict__ coef, int accumulate) {
    double db, dg;
    colsum_fin(partial, nblk, C, grp * FIN_C, db, dg);
    const int c = grp * FIN_C + threadIdx.x;
    if (threadIdx.x >= FIN_C || c >= C) return;
    dbeta[c] = accumulate ? dbeta[c] + (float)db : (float)db;
    dgamma[c] = accumulate ? dgamma[c] + (float)dg : (float)dg;
    coef[c] = gamma[c] * mean_rstd[C + c];
    coef[C + c] = (float)(db / P);
    coef[2 * C + c] = (float)(dg / P);
}
__global__ void __launch_bounds__(256) bn_finalize_bwd_kernel(const float* __restrict__ partial, int nblk, int P, int C,
                                                              const float* __restrict__ gamma, const float* __restrict__ mean_rstd,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ coef, int accumulate) {
    bn_finalize_bwd_group(blockIdx.x, partial, nblk, P, C, gamma, mean_rstd, dgamma, dbeta, coef, accumulate);
}

__device__ __forceinline__ void bn_bwd_apply_body(const float* __restrict__ x, const float* __restrict__ dy,
                                                  const float* __restrict__ y, const float* __restrict__ mean_rstd,
                                                  const float* __restrict__ coef, float* __restrict__ dx,
                                                  float* __restrict__ dres, long total4, int C, int relu,
                                                  const float* __restrict__ res_carry = nullptr, long ldc = 0, long lddy = 0) {
    const int C4 = C >> 2;
    const bool dense_dy = lddy == 0 || lddy == C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const int col = (int)(i % C4);
        float4 g = ld4(dense_dy ? dy + 4 * i : dy + (i / C4) * lddy + 4 * col);
        const float4 xv = ld4(x + 4 * i);
        if (relu == 1) g = mask4(g, ld4(y + 4 * i));
        else if (relu == 2) g = mask4(g, xv * ld4(y + 4 * col) + ld4(y + C + 4 * col));      // y = (scale, shift): see bn_reduce_tile
        // residual gradient (+ what other consumers of the residual tensor already contributed: gradient fan-in, pixel pitch ldc)
        if (dres != nullptr) st4(dres + 4 * i, res_carry != nullptr ? g + ld4(res_carry + (i / C4) * ldc + 4 * col) : g);
        const float4 xh = (xv - ld4(mean_rstd + 4 * col)) * ld4(mean_rstd + C + 4 * col);
        const float4 v = ld4(coef + 4 * col) * (g - ld4(coef + C + 4 * col) - xh * ld4(coef + 2 * C + 4 * col));
        st4(dx + 4 * i, v);
    }
}
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ y, const float* __restrict__ mean_rstd,
                                                           const float* __restrict__ coef, float* __restrict__ dx,
                                                           float* __restrict__ dres, long total4, int C, int relu,
                                                           const float* __restrict__ res_carry, long ldc, long lddy) {
    bn_bwd_apply_body(x, dy, y, mean_rstd, coef, dx, dres, total4, C, relu, res_carry, ldc, lddy);
}

// ---- round 5: the finalize folded into the apply pass (channel-group ownership) ---------------------------------------------
// The finalize launches were pure latency on the critical path (39 + 37 per DLA-34 step, a ~5 us slot each for a few hundred flops).
// Here a workgroup of the APPLY pass owns GC = 16 channels (one 64-byte run per pixel) of a pixel chunk and first sums the partial
// rows of exactly those channels itself: nblk x 128 bytes, every thread one or a few independent float4 loads -- one memory
// round trip instead of a dependent launch.  Nothing meets across workgroups, so there is no counter, no fence and no order to
// depend on; per channel the additions are colsum_fin's (same 4-row grouping, same tree over the 64 row groups), i.e. the
// coefficients are BIT-IDENTICAL to the three-launch path.  The workgroups of pixel chunk 0 also write mean / rstd / scale /
// shift / running statistics (forward) or dgamma / dbeta (backward).  Work items are walked in XCD-contiguous order: the
// channel groups of one pixel chunk run on the same XCD, so every 128-byte line is used whole inside one L2.
constexpr int GQ = 4, GC = 4 * GQ;
static_assert(256 / GQ == FIN_RG, "colsum_fin_quads walks the partial rows in colsum_fin's 64 row groups");
__device__ __forceinline__ void acc4(double (&a)[4], float4 v0, float4 v1, float4 v2, float4 v3) {
    a[0] += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
    a[1] += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
    a[2] += ((double)v0.z + (double)v1.z) + ((double)v2.z + (double)v3.z);
    a[3] += ((double)v0.w + (double)v1.w) + ((double)v2.w + (double)v3.w);
}
// totals of the GC channels starting at c0 over the nblk partial rows; thread t < GC returns (s0, s1) of channel c0 + t
__device__ __forceinline__ void colsum_fin_quads(const float* __restrict__ partial, int nblk, int C, int c0, double& s0, double& s1) {
    __shared__ double sm0[256][4], sm1[256][4];
    const int t = threadIdx.x, q = t % GQ, rg = t / GQ;
    const float* base = partial + c0 + 4 * q;
    const long st = (long)FIN_RG * 2 * C;
    double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
    int blk = rg;
#pragma unroll 2
    for (; blk + 3 * FIN_RG < nblk; blk += 4 * FIN_RG) {
        const float* p = base + (long)blk * 2 * C;
        const float4 a0 = ld4(p), a1 = ld4(p + st), a2 = ld4(p + 2 * st), a3 = ld4(p + 3 * st);
        const float4 b0 = ld4(p + C), b1 = ld4(p + st + C), b2 = ld4(p + 2 * st + C), b3 = ld4(p + 3 * st + C);
        acc4(a, a0, a1, a2, a3);
        acc4(b, b0, b1, b2, b3);
    }
    for (; blk < nblk; blk += FIN_RG) {
        const float4 av = ld4(base + (long)blk * 2 * C), bv = ld4(base + (long)blk * 2 * C + C);
        a[0] += (double)av.x; a[1] += (double)av.y; a[2] += (double)av.z; a[3] += (double)av.w;
        b[0] += (double)bv.x; b[1] += (double)bv.y; b[2] += (double)bv.z; b[3] += (double)bv.w;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { sm0[t][k] = a[k]; sm1[t][k] = b[k]; }
    __syncthreads();
    for (int half = FIN_RG / 2; half >= 1; half >>= 1) {
        if (rg < half) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { sm0[t][k] += sm0[t + half * GQ][k]; sm1[t][k] += sm1[t + half * GQ][k]; }
        }
        __syncthreads();
    }
    s0 = sm0[(t >> 2) & (GQ - 1)][t & 3];
    s1 = sm1[(t >> 2) & (GQ - 1)][t & 3];
}
// work item of this workgroup: (channel group, pixel chunk), XCD-contiguous; false = nothing to do (grid rounded up to 8)
__device__ __forceinline__ bool fin_item(int CG, int PC, int& cg, int& pc) {
    const int nitems = CG * PC, per = (nitems + 7) >> 3;
    const int item = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (item >= nitems || (int)(blockIdx.x >> 3) >= per) return false;
    cg = item % CG;
    pc = item / CG;
    return true;
}

__global__ void __launch_bounds__(256) bn_fin_apply_kernel(const float* __restrict__ x, const float* __restrict__ partial, int nblk, int P,
                                                           int C, float eps, float momentum, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ residual,
                                                           float* __restrict__ y, long ldy, float* __restrict__ mean_rstd,
                                                           float* __restrict__ scale_shift, float* __restrict__ running_mean,
                                                           float* __restrict__ running_var, int relu, int CG, int PC, int chunk) {
    int cg, pc;
    if (!fin_item(CG, PC, cg, pc)) return;
    const int c0 = cg * GC, t = threadIdx.x;
    __shared__ float s_ss[2 * GC];
    double sx, sxx;
    colsum_fin_quads(partial, nblk, C, c0, sx, sxx);
    if (t < GC) {       // (the arithmetic of bn_finalize_fwd_group)
        const int c = c0 + t;
        const double mean = sx / P;
        double var = sxx / P - mean * mean;
        if (var < 0) var = 0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[c] * rstd;
        const float sh = beta[c] - (float)mean * sc;
        s_ss[t] = sc;
        s_ss[GC + t] = sh;
        if (pc == 0) {
            mean_rstd[c] = (float)mean;
            mean_rstd[C + c] = rstd;
            scale_shift[c] = sc;
            scale_shift[C + c] = sh;
            if (running_mean != nullptr) {
                const double unbiased = P > 1 ? var * P / (P - 1) : var;
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
            }
        }
    }
    __syncthreads();
    const int col = t & (GQ - 1), prow = t >> 2;
    const float4 sc = ld4(s_ss + 4 * col), sh = ld4(s_ss + GC + 4 * col);
    const long pend = min((long)P, (long)(pc + 1) * chunk), coff = c0 + 4 * col;
    long p = (long)pc * chunk + prow;
    for (; p + 192 < pend; p += 256) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = ld4(x + (p + 64 * u) * C + coff) * sc + sh;
        if (residual != nullptr) {
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = v[u] + ld4(residual + (p + 64 * u) * C + coff);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) st4(y + (p + 64 * u) * ldy + coff, relu ? relu4(v[u]) : v[u]);
    }
    for (; p < pend; p += 64) {
        float4 v = ld4(x + p * C + coff) * sc + sh;
        if (residual != nullptr) v = v + ld4(residual + p * C + coff);
        st4(y + p * ldy + coff, relu ? relu4(v) : v);
    }
}

__global__ void __launch_bounds__(256) bn_fin_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy, long lddy,
                                                               const float* __restrict__ y, const float* __restrict__ mean_rstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ partial,
                                                               int nblk, int P, int C, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, int accumulate, float* __restrict__ dx,
                                                               float* __restrict__ dres, const float* __restrict__ res_carry, long ldc,
                                                               int relu, int CG, int PC, int chunk) {
    int cg, pc;
    if (!fin_item(CG, PC, cg, pc)) return;
    const int c0 = cg * GC, t = threadIdx.x;
    __shared__ float s_cf[3 * GC];
    double db, dg;
    colsum_fin_quads(partial, nblk, C, c0, db, dg);
    if (t < GC) {       // (the arithmetic of bn_finalize_bwd_group)
        const int c = c0 + t;
        if (pc == 0) {
            dbeta[c] = accumulate ? dbeta[c] + (float)db : (float)db;
            dgamma[c] = accumulate ? dgamma[c] + (float)dg : (float)dg;
        }
        s_cf[t] = gamma[c] * mean_rstd[C + c];
        s_cf[GC + t] = (float)(db / P);
        s_cf[2 * GC + t] = (float)(dg / P);
    }
    __syncthreads();
    const int col = t & (GQ - 1), prow = t >> 2;
    const long coff = c0 + 4 * col;
    const float4 k0 = ld4(s_cf + 4 * col), k1 = ld4(s_cf + GC + 4 * col), k2 = ld4(s_cf + 2 * GC + 4 * col);
    const float4 mu = ld4(mean_rstd + coff), rs = ld4(mean_rstd + C + coff);
    float4 msc = f4(0.f), msh = f4(0.f);
    if (relu == 2) { msc = ld4(y + coff); msh = ld4(y + C + coff); }       // y = (scale, shift): see bn_reduce_tile
    const long pend = min((long)P, (long)(pc + 1) * chunk);
    for (long p = (long)pc * chunk + prow; p < pend; p += 128) {
        // two pixels per trip: up to 8 independent loads in flight per lane
        const long p1 = p + 64;
        const bool two = p1 < pend;
        float4 g0 = ld4(dy + p * lddy + coff), g1 = two ? ld4(dy + p1 * lddy + coff) : f4(0.f);
        const float4 x0 = ld4(x + p * C + coff), x1 = two ? ld4(x + p1 * C + coff) : f4(0.f);
        if (relu == 1) {
            g0 = mask4(g0, ld4(y + p * C + coff));
            if (two) g1 = mask4(g1, ld4(y + p1 * C + coff));
        } else if (relu == 2) {
            g0 = mask4(g0, x0 * msc + msh);
            g1 = mask4(g1, x1 * msc + msh);
        }
        if (dres != nullptr) {
            st4(dres + p * C + coff, res_carry != nullptr ? g0 + ld4(res_carry + p * ldc + coff) : g0);
            if (two) st4(dres + p1 * C + coff, res_carry != nullptr ? g1 + ld4(res_carry + p1 * ldc + coff) : g1);
        }
        const float4 h0 = (x0 - mu) * rs, h1 = (x1 - mu) * rs;
        st4(dx + p * C + coff, k0 * (g0 - k1 - h0 * k2));
        if (two) st4(dx + p1 * C + coff, k0 * (g1 - k1 - h1 * k2));
    }
}

#ifndef OMNI_HIPEMU
// (A one-launch BatchNorm -- statistics, finalize and apply separated by grid-wide barriers built on agent-scope atomics -- was
// built and measured in round 2: inside a hipGraph a dependent launch costs ~2.8 us on MI355X while a grid barrier across the
// eight XCDs costs ~5 us with sc1 loads/stores and ~20 us with acquire/release fences (whole-L2 write-back + invalidate), so
// three launches win: 8.5 vs 14.6 us forward, 10.6 vs 18.7 us backward at 4 x 512 x 16 x 16.  tools/bench_bn.py reproduces the
// three-launch numbers.)
// bias gradient in one launch: column sums per workgroup, then one float atomic per (workgroup, channel) into db, which
// already holds the running gradient (accumulate) -- no partial rows, no finalize launch
__global__ void __launch_bounds__(256) bias_grad_atomic_kernel(const float* __restrict__ dy, int P, int C, float* __restrict__ db) {
    __shared__ float4 s0[256];
    const int C4 = C >> 2, rows = 256 / C4, t = threadIdx.x, col = t % C4, row = t / C4;
    float4 a = f4(0.f);
    if (row < rows) {
        const long step = (long)gridDim.x * rows;
        long p = (long)blockIdx.x * rows + row;
        for (; p + 3 * step < P; p += 4 * step) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = ld4(dy + (p + u * step) * C + 4 * col);
            a = a + ((v[0] + v[1]) + (v[2] + v[3]));
        }
        for (; p < P; p += step) a = a + ld4(dy + p * C + 4 * col);
    }
    s0[t] = a;
    __syncthreads();
    if (t < C4) {
        float4 r = s0[t];
        for (int k = 1; k < rows; ++k) r = r + s0[k * C4 + t];
        atomicAdd(db + 4 * t + 0, r.x); atomicAdd(db + 4 * t + 1, r.y); atomicAdd(db + 4 * t + 2, r.z); atomicAdd(db + 4 * t + 3, r.w);
    }
}
#endif

// ---- 2x2/s2 max-pool (NHWC); first maximum in window scan order wins, like ATen -----------------
__global__ void __launch_bounds__(256) maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N,
                                                           int H, int W, int C) {
    const int C4 = C >> 2, OH = H >> 1, OW = W >> 1;
    const long total = (long)N * OH * OW * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int col = (int)(i % C4);
        long q = i / C4;
        const int ow = (int)(q % OW); q /= OW;
        const int oh = (int)(q % OH);
        const int n = (int)(q / OH);
        const float* b = x + (((long)n * H + 2 * oh) * W + 2 * ow) * C + 4 * col;
        const float4 v00 = ld4(b), v01 = ld4(b + C), v10 = ld4(b + (long)W * C), v11 = ld4(b + (long)W * C + C);
        float4 m;
        m.x = fmaxf(fmaxf(v00.x, v01.x), fmaxf(v10.x, v11.x));
        m.y = fmaxf(fmaxf(v00.y, v01.y), fmaxf(v10.y, v11.y));
        m.z = fmaxf(fmaxf(v00.z, v01.z), fmaxf(v10.z, v11.z));
        m.w = fmaxf(fmaxf(v00.w, v01.w), fmaxf(v10.w, v11.w));
        st4(y + 4 * i, m);
    }
}

__device__ __forceinline__ void route(float v00, float v01, float v10, float v11, float g, float& d00, float& d01,
                                      float& d10, float& d11) {
    int k = 0;
    float m = v00;
    if (v01 > m) { m = v01; k = 1; }
    if (v10 > m) { m = v10; k = 2; }
    if (v11 > m) { m = v11; k = 3; }
    d00 = k == 0 ? g : 0.f; d01 = k == 1 ? g : 0.f; d10 = k == 2 ? g : 0.f; d11 = k == 3 ? g : 0.f;
}

__global__ void __launch_bounds__(256) maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ dx, int N, int H, int W, int C,
                                                           const float* __restrict__ carry, long ldc, long lddy) {
    const int C4 = C >> 2, OH = H >> 1, OW = W >> 1;
    const long total = (long)N * OH * OW * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int col = (int)(i % C4);
        long q = i / C4;
        const int ow = (int)(q % OW); q /= OW;
        const int oh = (int)(q % OH);
        const int n = (int)(q / OH);
        const long o = (((long)n * H + 2 * oh) * W + 2 * ow) * C + 4 * col;
        const float4 v00 = ld4(x + o), v01 = ld4(x + o + C), v10 = ld4(x + o + (long)W * C), v11 = ld4(x + o + (long)W * C + C);
        const float4 g = ld4(dy + (i / C4) * lddy + 4 * col);
        float4 d00, d01, d10, d11;
        route(v00.x, v01.x, v10.x, v11.x, g.x, d00.x, d01.x, d10.x, d11.x);
        route(v00.y, v01.y, v10.y, v11.y, g.y, d00.y, d01.y, d10.y, d11.y);
        route(v00.z, v01.z, v10.z, v11.z, g.z, d00.z, d01.z, d10.z, d11.z);
        route(v00.w, v01.w, v10.w, v11.w, g.w, d00.w, d01.w, d10.w, d11.w);
        if (carry != nullptr) {     // gradient fan-in: dx = routed dy + what the other consumers of x contributed (pixel pitch ldc)
            const long pc = (((long)n * H + 2 * oh) * W + 2 * ow) * ldc + 4 * col;
            d00 = d00 + ld4(carry + pc); d01 = d01 + ld4(carry + pc + ldc);
            d10 = d10 + ld4(carry + pc + (long)W * ldc); d11 = d11 + ld4(carry + pc + (long)W * ldc + ldc);
        }
        st4(dx + o, d00); st4(dx + o + C, d01); st4(dx + o + (long)W * C, d10); st4(dx + o + (long)W * C + C, d11);
    }
}

// ---- 2x2/s2 average pool (nn.AvgPool2d(2, 2), torchvision densenet `_Transition.pool`): DIR 0 forward, DIR 1 backward (every
//      input of a window gets dy / 4; an odd last row / column is outside every window and gets 0) ----------------------------
template <int DIR>
__global__ void __launch_bounds__(256) avgpool2_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int H, int W,
                                                       int C) {
    const int C4 = C >> 2, OH = H >> 1, OW = W >> 1;
    const long total = (long)N * OH * OW * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int col = (int)(i % C4);
        long q = i / C4;
        const int ow = (int)(q % OW); q /= OW;
        const int oh = (int)(q % OH);
        const int n = (int)(q / OH);
        const long o = (((long)n * H + 2 * oh) * W + 2 * ow) * C + 4 * col;
        if (DIR == 0) {
            const float4 a = ld4(src + o), b = ld4(src + o + C), c = ld4(src + o + (long)W * C), d = ld4(src + o + (long)W * C + C);
            float4 m;
            m.x = (a.x + b.x + c.x + d.x) * 0.25f; m.y = (a.y + b.y + c.y + d.y) * 0.25f;
            m.z = (a.z + b.z + c.z + d.z) * 0.25f; m.w = (a.w + b.w + c.w + d.w) * 0.25f;
            st4(dst + 4 * i, m);
        } else {
            float4 g = ld4(src + 4 * i);
            g.x *= 0.25f; g.y *= 0.25f; g.z *= 0.25f; g.w *= 0.25f;
            st4(dst + o, g); st4(dst + o + C, g); st4(dst + o + (long)W * C, g); st4(dst + o + (long)W * C + C, g);
        }
    }
}

// ---- stride-2 subsample (max_pool2d kernel 1, stride 2): y[n,oh,ow] = x[n,2oh,2ow] ----------------
// DIR 0: forward gather; DIR 1: backward scatter into a zero-initialised dx
// gradient fan-in form of the backward scatter: dx = carry everywhere (+ dy at the even pixels), one pass, no zero-fill
__global__ void __launch_bounds__(256) subsample2_bwd_carry_kernel(const float* __restrict__ dy, const float* __restrict__ carry, long ldc,
                                                                   float* __restrict__ dx, int N, int H, int W, int C) {
    const int C4 = C >> 2, OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const long total = (long)N * H * W * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int col = (int)(i % C4);
        long q = i / C4;
        const long pix = q;
        const int w = (int)(q % W); q /= W;
        const int h = (int)(q % H);
        const int n = (int)(q / H);
        float4 v = ld4(carry + pix * ldc + 4 * col);
        if (!((h | w) & 1)) v = v + ld4(dy + ((((long)n * OH + (h >> 1)) * OW + (w >> 1)) * C4 + col) * 4);
        st4(dx + 4 * i, v);
    }
}

template <int DIR>
__global__ void __launch_bounds__(256) subsample2_kernel(const float* __restrict__ src, float* __restrict__ dst, int N,
                                                         int H, int W, int C) {
    const int C4 = C >> 2, OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const long total = (long)N * OH * OW * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int col = (int)(i % C4);
        long q = i / C4;
        const int ow = (int)(q % OW); q /= OW;
        const int oh = (int)(q % OH);
        const int n = (int)(q / OH);
        const long big = (((long)n * H + 2 * oh) * W + 2 * ow) * C + 4 * col;
        if (DIR == 0) st4(dst + 4 * i, ld4(src + big));
        else st4(dst + big, ld4(src + 4 * i));
    }
}

// ---- FPN top-down: out = lateral + nearest_upsample_2x(top);  backward of the upsample ----------
__global__ void __launch_bounds__(256) upsample2_add_kernel(const float* __restrict__ lat, const float* __restrict__ top,
                                                            float* __restrict__ out, int N, int H, int W, int C) {
    const int C4 = C >> 2, TH = H >> 1, TW = W >> 1;
    const long total = (long)N * H * W * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int col = (int)(i % C4);
        long q = i / C4;
        const int w = (int)(q % W); q /= W;
        const int h = (int)(q % H);
        const int n = (int)(q / H);
        const float4 t = ld4(top + (((long)n * TH + (h >> 1)) * TW + (w >> 1)) * C + 4 * col);
        st4(out + 4 * i, ld4(lat + 4 * i) + t);
    }
}

__global__ void __launch_bounds__(256) upsample2_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dtop,
                                                            int N, int H, int W, int C, const float* __restrict__ carry, long ldc) {
    const int C4 = C >> 2, TH = H >> 1, TW = W >> 1;
    const long total = (long)N * TH * TW * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int col = (int)(i % C4);
        long q = i / C4;
        const int tw = (int)(q % TW); q /= TW;
        const int th = (int)(q % TH);
        const int n = (int)(q / TH);
        const long o = (((long)n * H + 2 * th) * W + 2 * tw) * C + 4 * col;
        float4 s = (ld4(dout + o) + ld4(dout + o + C)) + (ld4(dout + o + (long)W * C) + ld4(dout + o + (long)W * C + C));
        if (carry != nullptr) s = s + ld4(carry + (i / C4) * ldc + 4 * col);      // gradient fan-in (pixel pitch ldc)
        st4(dtop + 4 * i, s);
    }
}

// ---- image normalisation: uint8 planar (N,3,H,W) -> NHWC fp32 with 4 channels (4th = 0), the
//      bottom/right padding up to (PH, PW) is zero AFTER normalisation (ImageList.from_tensors) ---
// image_hw (nullable): device (N, 2) ints, the valid height / width of every image inside its H x W slot (a batch staged into
// fixed-size slots -- the size-bucketed graph replay of cubercnn/solver/autoreplay.py); pixels outside are written as zero padding
// imgs: the N images as SEPARATE planar uint8 tensors (3, H, W) -- the reference stacks them first (`torch.stack` inside
// ImageList.from_tensors); here the kernel reads image n through its own pointer (round 6: one copy launch less per step)
constexpr int PRE_MAXN = 64;
struct ImgPtrs {
    const unsigned char* p[PRE_MAXN];
};
template <bool MULTI>
__global__ void __launch_bounds__(256) preprocess_kernel(const unsigned char* __restrict__ img, ImgPtrs ptrs, float* __restrict__ out,
                                                         int N, int H, int W, int PH, int PW, float m0, float m1,
                                                         float m2, float s0, float s1, float s2, const int* __restrict__ image_hw) {
    const long total = (long)N * PH * PW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int w = (int)(i % PW);
        long q = i / PW;
        const int h = (int)(q % PH);
        const int n = (int)(q / PH);
        float4 v = f4(0.f);
        const int vh = image_hw != nullptr ? min(H, image_hw[2 * n]) : H, vw = image_hw != nullptr ? min(W, image_hw[2 * n + 1]) : W;
        if (h < vh && w < vw) {
            const unsigned char* b = (MULTI ? ptrs.p[n] : img + (long)n * 3 * H * W) + (long)h * W + w;
            v.x = ((float)b[0] - m0) / s0;
            v.y = ((float)b[(long)H * W] - m1) / s1;
            v.z = ((float)b[2L * H * W] - m2) / s2;
        }
        st4(out + 4 * i, v);
    }
}

// dz = dy where y > 0 else 0 (backward of the ReLU fused into a conv / linear epilogue)
__global__ void __launch_bounds__(256) relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                       float* __restrict__ dz, long total4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x)
        st4(dz + 4 * i, mask4(ld4(dy + 4 * i), ld4(y + 4 * i)));
}

__global__ void __launch_bounds__(256) colsum_finalize_kernel(const float* __restrict__ partial, int nblk, int C,
                                                              float* __restrict__ out, int accumulate) {
    double s0, s1;
    colsum_fin(partial, nblk, C, blockIdx.x * FIN_C, s0, s1);
    const int c = blockIdx.x * FIN_C + threadIdx.x;
    if (threadIdx.x < FIN_C && c < C) out[c] = accumulate ? out[c] + (float)s0 : (float)s0;
}

inline int ew_grid(long total) {
    long g = (total + 255) / 256;
    if (g > 256 * 8) g = 256 * 8;
    return (int)(g < 1 ? 1 : g);
}
inline int red_grid(int P, int C) {
    // >= 4 pixel rows per thread (one trip of the 4-deep load loop) before another workgroup is worth its partial-sum row;
    // <= 512 workgroups (the partial rows live in the caller's workspace).  Round 3: 16 -> 4 rows.  With 16 the level-3 / level-4
    // BatchNorms (8 / 4 MB tensors) ran their reductions on 128 / 64 workgroups at ~1.3 TB/s -- a latency chain, not a stream.
    const int rows = (C >> 2) >= 256 ? 1 : 256 / (C >> 2);
    long g = ((long)P + 4L * rows - 1) / (4L * rows);
    if (g > 512) g = 512;
    return (int)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" {

// pixel chunks of the fused finalize + apply launches: ~2 workgroups per CU over (channel groups x chunks), chunks of >= 64 pixels
// that are multiples of 64 (one pixel per 4 lanes and trip)
static inline void fin_geometry(int P, int C, int& CG, int& PC, int& chunk, int& grid) {
    CG = C / GC;
    long pc = 512 / CG;
    const long maxpc = ((long)P + 63) / 64;
    if (pc > maxpc) pc = maxpc;
    if (pc < 1) pc = 1;
    chunk = (int)((((long)P + pc - 1) / pc + 63) / 64 * 64);
    PC = (int)(((long)P + chunk - 1) / chunk);
    grid = (CG * PC + 7) / 8 * 8;
}
static inline bool fin_fusable(int C, int nblk, int fuse_rows) { return fuse_rows > 0 && nblk <= fuse_rows && (C % GC) == 0; }
static const int BN_FUSE_ROWS_DEFAULT = 512;

// Training-mode BatchNorm forward on NHWC x (P = N*H*W pixels, C channels, C % 4 == 0, C <= 4096).
// partial [nullable]: [nblk][2][C] per-block sums / sums of squares already made by the PRODUCER of x (conv / Winograd / stem
// epilogues); NULL: a statistics pass over x runs first (ws: >= 2*C*258 doubles of scratch).  ldy: pixel pitch of y in floats
// (>= C, % 4 == 0; y 16-byte aligned).  mean_rstd (2C), scale_shift (2C) are kept for backward.
// fuse_rows: the finalize is folded into the apply launch when there are <= fuse_rows partial rows and C % 16 == 0 (0 = never:
// the round 1-4 finalize launch); both forms give the same bits.
int omni_bn_fwd_algo(const float* x, const float* partial, int nblk, const float* gamma, const float* beta, const float* residual,
                     float* y, long long ldy, float* running_mean, float* running_var, float* mean_rstd, float* scale_shift, double* ws,
                     int P, int C, float eps, float momentum, int relu, int fuse_rows, void* stream) {
    if (P <= 0 || C <= 0 || (C & 3) || C > 4096 || (partial != nullptr && nblk <= 0) || (partial == nullptr && ws == nullptr))
        return OMNI_ERR_ARG;
    if (ldy == 0) ldy = C;
    if (ldy < C || (ldy & 3) || (((unsigned long long)y) & 15)) return OMNI_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (partial == nullptr) {
        nblk = red_grid(P, C);
        float* mine = reinterpret_cast<float*>(ws + 2 * C);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(bn_reduce_kernel<0>), dim3(nblk), dim3(256), 0, st, x, (const float*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, P, C, 0, reinterpret_cast<double*>(mine), 0L);
        partial = mine;
    }
    if (fin_fusable(C, nblk, fuse_rows)) {
        int CG, PC, chunk, grid;
        fin_geometry(P, C, CG, PC, chunk, grid);
        hipLaunchKernelGGL(bn_fin_apply_kernel, dim3(grid), dim3(256), 0, st, x, partial, nblk, P, C, eps, momentum, gamma, beta, residual,
                           y, (long)ldy, mean_rstd, scale_shift, running_mean, running_var, relu, CG, PC, chunk);
        return omni_launch_status();
    }
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3((C + FIN_C - 1) / FIN_C), dim3(256), 0, st, partial, nblk, P, C, eps, momentum, gamma,
                       beta, mean_rstd, scale_shift, running_mean, running_var);
    const long total4 = (long)P * (C >> 2);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(total4)), dim3(256), 0, st, x, (const float*)scale_shift, residual, y, total4, C, relu,
                       (long)ldy);
    return omni_launch_status();
}

int omni_bn_fwd(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                float* running_mean, float* running_var, float* mean_rstd, float* scale_shift, double* ws, int P, int C,
                float eps, float momentum, int relu, void* stream) {
    return omni_bn_fwd_algo(x, nullptr, 0, gamma, beta, residual, y, C, running_mean, running_var, mean_rstd, scale_shift, ws, P, C, eps,
                            momentum, relu, BN_FUSE_ROWS_DEFAULT, stream);
}

// The same forward with the statistics pass already done by the PRODUCER of x (conv / Winograd / stem epilogues):
// partial [nblk][2][C] floats = per-block sums and sums of squares.
int omni_bn_fwd_partials(const float* x, const float* partial, int nblk, const float* gamma, const float* beta, const float* residual,
                         float* y, float* running_mean, float* running_var, float* mean_rstd, float* scale_shift, int P, int C,
                         float eps, float momentum, int relu, void* stream) {
    if (partial == nullptr) return OMNI_ERR_ARG;
    return omni_bn_fwd_algo(x, partial, nblk, gamma, beta, residual, y, C, running_mean, running_var, mean_rstd, scale_shift, nullptr, P, C,
                            eps, momentum, relu, BN_FUSE_ROWS_DEFAULT, stream);
}

// The finalize half of omni_bn_fwd_partials alone: batch statistics -> mean_rstd (2C), scale_shift (2C), running statistics updated.
// For a BatchNorm(+ReLU) whose only consumer is a Winograd convolution: that layer's input transform applies (scale, shift) on load
// (omni_wino_in_affine) and the normalised tensor is never materialised.
int omni_bn_finalize_fwd(const float* partial, int nblk, const float* gamma, const float* beta, float* running_mean, float* running_var,
                         float* mean_rstd, float* scale_shift, int P, int C, float eps, float momentum, void* stream) {
    if (P <= 0 || C <= 0 || (C & 3) || C > 4096 || nblk <= 0) return OMNI_ERR_ARG;
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3((C + FIN_C - 1) / FIN_C), dim3(256), 0, (hipStream_t)stream, partial, nblk, P, C, eps,
                       momentum, gamma, beta, mean_rstd, scale_shift, running_mean, running_var);
    return omni_launch_status();
}

// Inference / frozen BN: y = relu?(x*scale + shift (+res)) with caller-provided scale_shift (2C).
int omni_bn_apply(const float* x, const float* scale_shift, const float* residual, float* y, int P, int C, int relu,
                  void* stream) {
    if (P <= 0 || C <= 0 || (C & 3)) return OMNI_ERR_ARG;
    const long total4 = (long)P * (C >> 2);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(total4)), dim3(256), 0, (hipStream_t)stream, x, scale_shift, residual,
                       y, total4, C, relu, 0L);
    return omni_launch_status();
}

// Backward.  dy is the gradient wrt the (post-ReLU) output y; dres [nullable] receives the
// gradient of the residual input.  ws: >= 2*C doubles, coef: 3*C floats of scratch.
static int bad_carry(const float* carry, long long ldc, int C) {
    return carry != nullptr && (ldc < C || (ldc & 3) || (((unsigned long long)carry) & 15));
}

// Backward of omni_bn_fwd(_algo) with gradient fan-in on the residual: dres = (masked dy) + res_carry, res_carry [nullable] an NHWC
// tensor of the same extent with pixel pitch ldc floats (what the other consumers of the residual tensor already contributed to its
// gradient); dy with pixel pitch lddy floats (a channel slice of the DLA Root's concatenated gradient is read where it lies).
// partial [nullable]: [nblk][2][C] reductions (sum dz, sum dz * xhat) already made by the kernel that produced dy
// (omni_wino_out_bn_bwd_stats; dense dy, no carry); NULL: the reduction pass runs first (ws: >= 2*C*258 doubles).
// fuse_rows: as omni_bn_fwd_algo (coef, 3C floats of scratch, is only touched by the unfused form).
int omni_bn_bwd_algo(const float* x, const float* dy, long long lddy, const float* y, const float* gamma, const float* mean_rstd,
                     const float* partial, int nblk, float* dx, float* dres, const float* res_carry, long long ldc, float* dgamma,
                     float* dbeta, double* ws, float* coef, int P, int C, int relu, int accumulate_param_grads, int fuse_rows,
                     void* stream) {
    if (P <= 0 || C <= 0 || (C & 3) || C > 4096 || relu < 0 || relu > 2 || (relu && y == nullptr)) return OMNI_ERR_ARG;
    if (lddy == 0) lddy = C;
    if (bad_carry(res_carry, ldc, C) || (res_carry != nullptr && dres == nullptr) || bad_carry(dy, lddy, C)) return OMNI_ERR_ARG;
    if ((partial != nullptr && (nblk <= 0 || res_carry != nullptr || lddy != C)) || (partial == nullptr && ws == nullptr)) return OMNI_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (partial == nullptr) {
        nblk = red_grid(P, C);
        float* mine = reinterpret_cast<float*>(ws + 2 * C);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(bn_reduce_kernel<1>), dim3(nblk), dim3(256), 0, st, x, dy, y, mean_rstd, P, C, relu,
                           reinterpret_cast<double*>(mine), (long)lddy);
        partial = mine;
    }
    if (fin_fusable(C, nblk, fuse_rows)) {
        int CG, PC, chunk, grid;
        fin_geometry(P, C, CG, PC, chunk, grid);
        hipLaunchKernelGGL(bn_fin_bwd_apply_kernel, dim3(grid), dim3(256), 0, st, x, dy, (long)lddy, y, mean_rstd, gamma, partial, nblk, P, C,
                           dgamma, dbeta, accumulate_param_grads, dx, dres, res_carry, (long)ldc, relu, CG, PC, chunk);
        return omni_launch_status();
    }
    if (coef == nullptr) return OMNI_ERR_ARG;
    hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3((C + FIN_C - 1) / FIN_C), dim3(256), 0, st, partial, nblk, P, C, gamma, mean_rstd,
                       dgamma, dbeta, coef, accumulate_param_grads);
    const long total4 = (long)P * (C >> 2);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(total4)), dim3(256), 0, st, x, dy, y, mean_rstd,
                       (const float*)coef, dx, dres, total4, C, relu, res_carry, (long)ldc, (long)lddy);
    return omni_launch_status();
}

int omni_bn_bwd_carry(const float* x, const float* dy, long long lddy, const float* y, const float* gamma, const float* mean_rstd, float* dx,
                      float* dres, const float* res_carry, long long ldc, float* dgamma, float* dbeta, double* ws, float* coef, int P,
                      int C, int relu, int accumulate_param_grads, void* stream) {
    return omni_bn_bwd_algo(x, dy, lddy, y, gamma, mean_rstd, nullptr, 0, dx, dres, res_carry, ldc, dgamma, dbeta, ws, coef, P, C, relu,
                            accumulate_param_grads, BN_FUSE_ROWS_DEFAULT, stream);
}

int omni_bn_bwd(const float* x, const float* dy, const float* y, const float* gamma, const float* mean_rstd, float* dx,
                float* dres, float* dgamma, float* dbeta, double* ws, float* coef, int P, int C, int relu,
                int accumulate_param_grads, void* stream) {
    return omni_bn_bwd_carry(x, dy, C, y, gamma, mean_rstd, dx, dres, nullptr, 0, dgamma, dbeta, ws, coef, P, C, relu,
                             accumulate_param_grads, stream);
}

// omni_bn_bwd with the reductions already done by the kernel that produced dy (omni_wino_out_bn_bwd_stats): finalize + apply.
int omni_bn_bwd_partials(const float* x, const float* dy, const float* y, const float* gamma, const float* mean_rstd,
                         const float* partial, int nblk, float* dx, float* dres, float* dgamma, float* dbeta, float* coef, int P,
                         int C, int relu, int accumulate_param_grads, void* stream) {
    if (partial == nullptr) return OMNI_ERR_ARG;
    return omni_bn_bwd_algo(x, dy, C, y, gamma, mean_rstd, partial, nblk, dx, dres, nullptr, 0, dgamma, dbeta, nullptr, coef, P, C, relu,
                            accumulate_param_grads, BN_FUSE_ROWS_DEFAULT, stream);
}

int omni_maxpool2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream) {
    if ((C & 3) || H < 2 || W < 2) return OMNI_ERR_ARG;
    const long total = (long)N * (H / 2) * (W / 2) * (C / 4);
    if (total == 0) return OMNI_OK;
    hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, C);
    return omni_launch_status();
}

// carry [nullable]: gradient fan-in, dx = routed dy + carry (NHWC, pixel pitch ldc floats; H and W even when given); dy with pixel
// pitch lddy floats
int omni_maxpool2_bwd_carry(const float* x, const float* dy, long long lddy, const float* carry, long long ldc, float* dx, int N, int H,
                            int W, int C, void* stream) {
    if ((C & 3) || H < 2 || W < 2 || bad_carry(carry, ldc, C) || (carry != nullptr && ((H & 1) || (W & 1))) || bad_carry(dy, lddy, C))
        return OMNI_ERR_ARG;
    const long total = (long)N * (H / 2) * (W / 2) * (C / 4);
    if (total == 0) return OMNI_OK;
    if ((H & 1) || (W & 1)) omni_memset_async(dx, 0, sizeof(float) * (size_t)N * H * W * C, (hipStream_t)stream);
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, N, H, W, C, carry, (long)ldc,
                       (long)lddy);
    return omni_launch_status();
}

int omni_maxpool2_bwd(const float* x, const float* dy, float* dx, int N, int H, int W, int C, void* stream) {
    return omni_maxpool2_bwd_carry(x, dy, C, nullptr, 0, dx, N, H, W, C, stream);
}

int omni_avgpool2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream) {
    if ((C & 3) || H < 2 || W < 2) return OMNI_ERR_ARG;
    const long total = (long)N * (H / 2) * (W / 2) * (C / 4);
    if (total == 0) return OMNI_OK;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(avgpool2_kernel<0>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, C);
    return omni_launch_status();
}

int omni_avgpool2_bwd(const float* dy, float* dx, int N, int H, int W, int C, void* stream) {
    if ((C & 3) || H < 2 || W < 2) return OMNI_ERR_ARG;
    const long total = (long)N * (H / 2) * (W / 2) * (C / 4);
    if (total == 0) return OMNI_OK;
    if ((H & 1) || (W & 1)) omni_memset_async(dx, 0, sizeof(float) * (size_t)N * H * W * C, (hipStream_t)stream);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(avgpool2_kernel<1>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, dy, dx, N, H, W, C);
    return omni_launch_status();
}

int omni_subsample2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream) {
    if (C & 3) return OMNI_ERR_ARG;
    const long total = (long)N * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1) * (C / 4);
    if (total == 0) return OMNI_OK;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(subsample2_kernel<0>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x,
                       y, N, H, W, C);
    return omni_launch_status();
}

int omni_subsample2_bwd(const float* dy, float* dx, int N, int H, int W, int C, void* stream) {
    if (C & 3) return OMNI_ERR_ARG;
    const long total = (long)N * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1) * (C / 4);
    omni_memset_async(dx, 0, sizeof(float) * (size_t)N * H * W * C, (hipStream_t)stream);
    if (total == 0) return OMNI_OK;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(subsample2_kernel<1>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, dy,
                       dx, N, H, W, C);
    return omni_launch_status();
}

// carry (required): gradient fan-in, dx = carry + scattered dy (NHWC (N,H,W,C), pixel pitch ldc floats)
int omni_subsample2_bwd_carry(const float* dy, const float* carry, long long ldc, float* dx, int N, int H, int W, int C, void* stream) {
    if ((C & 3) || carry == nullptr || bad_carry(carry, ldc, C)) return OMNI_ERR_ARG;
    const long total = (long)N * H * W * (C / 4);
    if (total == 0) return OMNI_OK;
    hipLaunchKernelGGL(subsample2_bwd_carry_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, dy, carry, (long)ldc, dx, N, H,
                       W, C);
    return omni_launch_status();
}

// out (N,H,W,C) = lat (N,H,W,C) + nearest-2x(top (N,H/2,W/2,C));  H, W even.
int omni_upsample2_add(const float* lat, const float* top, float* out, int N, int H, int W, int C, void* stream) {
    if ((C & 3) || (H & 1) || (W & 1)) return OMNI_ERR_ARG;
    const long total = (long)N * H * W * (C / 4);
    if (total == 0) return OMNI_OK;
    hipLaunchKernelGGL(upsample2_add_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, lat, top, out, N, H,
                       W, C);
    return omni_launch_status();
}

// dtop (N,H/2,W/2,C) = 2x2 block sums of dout (N,H,W,C).
// carry [nullable]: gradient fan-in, dtop = block sums + carry (N,H/2,W/2,C with pixel pitch ldc floats)
int omni_upsample2_bwd_carry(const float* dout, const float* carry, long long ldc, float* dtop, int N, int H, int W, int C, void* stream) {
    if ((C & 3) || (H & 1) || (W & 1) || bad_carry(carry, ldc, C)) return OMNI_ERR_ARG;
    const long total = (long)N * (H / 2) * (W / 2) * (C / 4);
    if (total == 0) return OMNI_OK;
    hipLaunchKernelGGL(upsample2_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, dout, dtop, N, H, W, C, carry, (long)ldc);
    return omni_launch_status();
}

int omni_upsample2_bwd(const float* dout, float* dtop, int N, int H, int W, int C, void* stream) {
    return omni_upsample2_bwd_carry(dout, nullptr, 0, dtop, N, H, W, C, stream);
}

// img uint8 (N,3,H,W) -> out fp32 NHWC (N,PH,PW,4): (v - mean)/std per channel, zero padded.
int omni_preprocess_masked(const unsigned char* img, const int* image_hw, float* out, int N, int H, int W, int PH, int PW, float m0,
                           float m1, float m2, float s0, float s1, float s2, void* stream) {
    if (PH < H || PW < W) return OMNI_ERR_ARG;
    const long total = (long)N * PH * PW;
    if (total == 0) return OMNI_OK;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(preprocess_kernel<false>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, img, ImgPtrs{}, out, N, H,
                       W, PH, PW, m0, m1, m2, s0, s1, s2, image_hw);
    return omni_launch_status();
}

// The same from N <= 64 separate images: imgs = HOST array of N device pointers, each a planar uint8 (3, H, W) image.
int omni_preprocess_multi(const void* const* imgs, const int* image_hw, float* out, int N, int H, int W, int PH, int PW, float m0,
                          float m1, float m2, float s0, float s1, float s2, void* stream) {
    if (PH < H || PW < W || N < 0 || N > PRE_MAXN || (N > 0 && imgs == nullptr)) return OMNI_ERR_ARG;
    const long total = (long)N * PH * PW;
    if (total == 0) return OMNI_OK;
    ImgPtrs ptrs{};
    for (int n = 0; n < N; ++n) {
        if (imgs[n] == nullptr) return OMNI_ERR_ARG;
        ptrs.p[n] = (const unsigned char*)imgs[n];
    }
    hipLaunchKernelGGL(HIP_KERNEL_NAME(preprocess_kernel<true>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned char*)nullptr, ptrs, out, N, H, W, PH, PW, m0, m1, m2, s0, s1, s2, image_hw);
    return omni_launch_status();
}

int omni_preprocess(const unsigned char* img, float* out, int N, int H, int W, int PH, int PW, float m0, float m1,
                    float m2, float s0, float s1, float s2, void* stream) {
    return omni_preprocess_masked(img, nullptr, out, N, H, W, PH, PW, m0, m1, m2, s0, s1, s2, stream);
}

// dz = dy * (y > 0), n elements (n % 4 == 0).
int omni_relu_bwd(const float* dy, const float* y, float* dz, long long n, void* stream) {
    if (n < 0 || (n & 3)) return OMNI_ERR_ARG;
    if (n == 0) return OMNI_OK;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(ew_grid(n >> 2)), dim3(256), 0, (hipStream_t)stream, dy, y, dz, (long)(n >> 2));
    return omni_launch_status();
}

// db[c] = sum over the P pixels of dy[p, c] (bias gradient of a conv / linear).  ws: 2*C doubles.
// accumulate: 0 = overwrite db | 1 = add to db (small tensors: one launch with <= 64 fp32 atomics per channel) | 3 = add to db
// deterministically (always partial rows + a fixed-order finalize)
int omni_bias_grad(const float* dy, int P, int C, float* db, double* ws, int accumulate, void* stream) {
    if (P <= 0 || C <= 0 || (C & 3) || C > 4096) return OMNI_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = red_grid(P, C);
    float* partial = reinterpret_cast<float*>(ws + 2 * C);
#ifndef OMNI_HIPEMU
    if (accumulate == 1 && nblk <= 64) {     // small tensors: db already holds the running gradient, add this call's column sums
        hipLaunchKernelGGL(bias_grad_atomic_kernel, dim3(nblk), dim3(256), 0, st, dy, P, C, db);   // with <= 64 float atomics per channel
        return omni_launch_status();
    }
#endif
    hipLaunchKernelGGL(HIP_KERNEL_NAME(bn_reduce_kernel<0>), dim3(nblk), dim3(256), 0, st, dy, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, P, C, 0, reinterpret_cast<double*>(partial), 0L);
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((C + FIN_C - 1) / FIN_C), dim3(256), 0, st, (const float*)partial, nblk, C, db,
                       accumulate);
    return omni_launch_status();
}

}  // extern "C"
