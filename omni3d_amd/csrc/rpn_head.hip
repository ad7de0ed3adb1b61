// rpn_head.hip -- the two 1x1 convolutions of the RPN head (objectness logits + anchor deltas) over ALL FPN levels, for gfx950.
//
// Reference: detectron2 StandardRPNHead.forward (configs/Base.yaml:49 RPN.HEAD_NAME "StandardRPNHead"), called by
// RPNWithIgnore.forward at /root/reference/cubercnn/modeling/proposal_generator/rpn.py:129-135 through detectron2's RPN.forward:
//     t = relu(conv(x));  objectness_logits(t) : nn.Conv2d(C, A, 1)   anchor_deltas(t) : nn.Conv2d(C, 4A, 1)        (A = 3 anchors)
// per level of p2..p6.  Here the two heads are ONE 16-wide product per pixel: Y[p, 0:3] = logits, Y[p, 3:15] = deltas, Y[p, 15] = 0
// (the layout the RPN loss / decode kernels of rpn_roi.hip read), for the pixels of every level in one launch.
//
// Why own kernels: an (87296 x 256) x (256 x 16) product is 0.7 GFLOP against 95 MB of traffic -- HBM-bound by two orders of
// magnitude.  The implicit-GEMM tile kernels (csrc/conv_gemm.hip) spend their time on tile machinery instead: the data gradient of
// the p2 level (a 65536 x 16 x 256 product whose "reduction" is half an MFMA slab) ran 197 us for 67 MB written = 0.34 TB/s, the
// forward 57 us = 1.2 TB/s, one launch per level and direction, plus ReLU-backward, bias-gradient and parameter-gradient-add
// launches per level.  Here:
//   forward   MFMA 16x16x4 f32: a wave takes 16 pixels; every lane loads float4s of its pixel straight from HBM in the MFMA A
//             layout (no LDS), the 16 x 256 weights stay in 64 VGPRs per lane for the wave's lifetime.
//   dgrad     VALU: a wave owns one pixel row (64 lanes x float4 = 256 channels, fully coalesced 1 KB loads / stores), the 16
//             output gradients of the pixel are loaded once (64 bytes) and broadcast by v_readlane, the weights sit in 60 VGPRs; the ReLU mask of the
//             shared 3x3 convolution's output is applied on the way out (no separate ReLU-backward pass).
//   wgrad     VALU, same ownership: 15 float4 accumulators per lane over a grid-stride pixel loop, LDS reduction over the 4 waves
//             of a workgroup, one partial row per workgroup, then a finalize kernel that sums the rows and writes / accumulates the
//             two weight gradients and the two bias gradients directly (no concatenated gradient, no per-parameter add kernels).
//             Deterministic: no atomics.
// C (input channels) is fixed at 256 (MODEL.FPN.OUT_CHANNELS of every reference config); other widths stay on the generic kernels.
#include <device_rt.h>

namespace {

constexpr int HC = 256;          // input channels
constexpr int HN = 16;           // output columns: 3 logits | 12 deltas | 1 pad
constexpr int MAXLV = 8;
constexpr int WG_PARTIAL = 15 * HC + HN;      // floats per workgroup partial row of the weight gradient: [15][256] dW | [16] db

struct HeadLevels {
    const float* t[MAXLV];       // (P_l, 256) NHWC activations (ReLU output of the shared 3x3 convolution)
    float* y[MAXLV];             // forward: (P_l, 16) outputs | dgrad: (P_l, 256) input gradients
    const float* dy[MAXLV];      // dgrad / wgrad: (P_l, 16) output gradients
    long start[MAXLV + 1];       // forward: prefix of ceil(P_l / 16) pixel groups | dgrad, wgrad: prefix of P_l
    long pix[MAXLV];             // P_l
    int nlev;
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 z4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// row n of the 16 x 256 weight matrix [objectness (3) | deltas (12) | zero]
__device__ __forceinline__ const float* wrow(const float* __restrict__ w_obj, const float* __restrict__ w_del, int n) {
    return n < 3 ? w_obj + n * HC : w_del + (n - 3) * HC;
}

// ---- forward ------------------------------------------------------------------------------------------------------------------
// lane (r = lane & 15, q = lane >> 4): A[i = r (pixel)][k = q], B[k = q][j = r (output)]; the K order inside a 16-channel chunk is
// (q, i) -> channel 16 j + 4 q + i for BOTH operands, so a lane's float4 feeds 4 consecutive MFMAs.
__global__ void __launch_bounds__(256) head16_fwd_kernel(HeadLevels lv, const float* __restrict__ w_obj, const float* __restrict__ b_obj,
                                                         const float* __restrict__ w_del, const float* __restrict__ b_del) {
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    const long groups = lv.start[lv.nlev];
    if (wave >= groups) return;
    float4 B[16];
    if (r < 15) {
        const float* wr = wrow(w_obj, w_del, r);
#pragma unroll
        for (int j = 0; j < 16; ++j) B[j] = ld4(wr + 16 * j + 4 * q);
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) B[j] = z4();
    }
    const float bias = r < 3 ? b_obj[r] : (r < 15 ? b_del[r - 3] : 0.f);
    for (long g = wave; g < groups; g += nwaves) {
        int l = 0;
        while (l + 1 < lv.nlev && g >= lv.start[l + 1]) ++l;
        const long p0 = (g - lv.start[l]) * 16;
        const long P = lv.pix[l];
        const bool ok = p0 + r < P;
        const float* a = lv.t[l] + (p0 + (ok ? r : 0)) * HC + 4 * q;
        float4 av[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) av[j] = ok ? ld4(a + 16 * j) : z4();
        f32x4 acc = {bias, bias, bias, bias};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            acc = mfma_16x16x4(av[j].x, B[j].x, acc);
            acc = mfma_16x16x4(av[j].y, B[j].y, acc);
            acc = mfma_16x16x4(av[j].z, B[j].z, acc);
            acc = mfma_16x16x4(av[j].w, B[j].w, acc);
        }
        float* o = lv.y[l] + (p0 + 4 * q) * HN + r;        // D: column r, rows 4 q + i
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (p0 + 4 * q + i < P) o[i * HN] = acc[i];
    }
}

// ---- data gradient: dt[p, c] = relu'(t[p, c]) * sum_n dy[p, n] w[n, c] ---------------------------------------------------------
__global__ void __launch_bounds__(256) head16_dgrad_kernel(HeadLevels lv, const float* __restrict__ w_obj, const float* __restrict__ w_del,
                                                           int relu_mask) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    const long total = lv.start[lv.nlev];
    float4 W[15];
#pragma unroll
    for (int n = 0; n < 15; ++n) W[n] = ld4(wrow(w_obj, w_del, n) + 4 * lane);
    constexpr int U = 4;            // pixels per trip: U independent 1 KB loads in flight per wave
    for (long base = wave * U; base < total; base += nwaves * U) {
        float4 tv[U], out[U];
        float dyv[U];               // lane n (mod 16) holds dy[p, n]: one coalesced 64-byte load per pixel, broadcast by v_readlane
        float* op[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long p = base + u;
            ok[u] = p < total;
            int l = 0;
            while (l + 1 < lv.nlev && p >= lv.start[l + 1]) ++l;
            const long pl = ok[u] ? p - lv.start[l] : 0;
            dyv[u] = ok[u] ? lv.dy[l][pl * HN + (lane & 15)] : 0.f;
            op[u] = lv.y[l] + pl * HC + 4 * lane;
            tv[u] = (ok[u] && relu_mask) ? ld4(lv.t[l] + pl * HC + 4 * lane) : make_float4(1.f, 1.f, 1.f, 1.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float4 a = z4();
#pragma unroll
            for (int n = 0; n < 15; ++n) {
                const float g = omni_readlane(dyv[u], n);       // wave-uniform (an SGPR operand of the four FMAs)
                a.x = fmaf(g, W[n].x, a.x); a.y = fmaf(g, W[n].y, a.y); a.z = fmaf(g, W[n].z, a.z); a.w = fmaf(g, W[n].w, a.w);
            }
            out[u] = make_float4(tv[u].x > 0.f ? a.x : 0.f, tv[u].y > 0.f ? a.y : 0.f, tv[u].z > 0.f ? a.z : 0.f, tv[u].w > 0.f ? a.w : 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (ok[u]) st4(op[u], out[u]);
    }
}

// ---- weight gradient: dW[n, c] = sum_p dy[p, n] t[p, c], db[n] = sum_p dy[p, n] -------------------------------------------------
__global__ void __launch_bounds__(256) head16_wgrad_kernel(HeadLevels lv, float* __restrict__ partial) {
    __shared__ float4 red[3][15][64];
    __shared__ float redb[4][HN];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const long wave = (long)blockIdx.x * 4 + wid, nwaves = (long)gridDim.x * 4;
    const long total = lv.start[lv.nlev];
    float4 acc[15];
#pragma unroll
    for (int n = 0; n < 15; ++n) acc[n] = z4();
    float bsum = 0.f;
    constexpr int U = 4;
    for (long base = wave * U; base < total; base += nwaves * U) {
        float4 tv[U];
        float dyv[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long p = base + u;
            ok[u] = p < total;
            int l = 0;
            while (l + 1 < lv.nlev && p >= lv.start[l + 1]) ++l;
            const long pl = ok[u] ? p - lv.start[l] : 0;
            dyv[u] = ok[u] ? lv.dy[l][pl * HN + (lane & 15)] : 0.f;
            tv[u] = ok[u] ? ld4(lv.t[l] + pl * HC + 4 * lane) : z4();      // (a zero activation row contributes nothing)
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bsum += dyv[u];
#pragma unroll
            for (int n = 0; n < 15; ++n) {
                const float g = omni_readlane(dyv[u], n);
                acc[n].x = fmaf(g, tv[u].x, acc[n].x); acc[n].y = fmaf(g, tv[u].y, acc[n].y);
                acc[n].z = fmaf(g, tv[u].z, acc[n].z); acc[n].w = fmaf(g, tv[u].w, acc[n].w);
            }
        }
    }
    // workgroup reduction in a fixed order (wave 0 + 1 + 2 + 3), one partial row per workgroup
    if (wid > 0) {
#pragma unroll
        for (int n = 0; n < 15; ++n) red[wid - 1][n][lane] = acc[n];
    }
    if (lane < HN) redb[wid][lane] = bsum;
    __syncthreads();
    float* row = partial + (long)blockIdx.x * WG_PARTIAL;
    if (wid == 0) {
#pragma unroll
        for (int n = 0; n < 15; ++n) {
            float4 v = acc[n];
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                const float4 o = red[w][n][lane];
                v = make_float4(v.x + o.x, v.y + o.y, v.z + o.z, v.w + o.w);
            }
            st4(row + n * HC + 4 * lane, v);
        }
        if (lane < HN) row[15 * HC + lane] = ((redb[0][lane] + redb[1][lane]) + redb[2][lane]) + redb[3][lane];
    }
}

// sums the workgroup rows in a fixed order and writes (accumulate == 0) or adds (accumulate != 0) the four parameter gradients.
// Round 4: 4 elements per workgroup x 64 row groups (every thread adds rows rg, rg + 64, ... in ascending order, the group sums
// meet in an LDS tree) instead of one thread walking all ~680 rows of its element: 66 -> ~8 us on the weight-gradient stream.
__global__ void __launch_bounds__(256) head16_wgrad_finalize_kernel(const float* __restrict__ partial, int rows, float* __restrict__ dw_obj,
                                                                    float* __restrict__ db_obj, float* __restrict__ dw_del,
                                                                    float* __restrict__ db_del, int accumulate) {
    __shared__ float sm[256];
    const int t = threadIdx.x, el = t & 3, rg = t >> 2;
    const int e = blockIdx.x * 4 + el;
    float s = 0.f;
    if (e < WG_PARTIAL)
        for (int rr = rg; rr < rows; rr += 64) s += partial[(long)rr * WG_PARTIAL + e];
    sm[t] = s;
    __syncthreads();
    for (int h = 128; h >= 4; h >>= 1) {
        if (t < h) sm[t] += sm[t + h];
        __syncthreads();
    }
    if (t >= 4 || e >= WG_PARTIAL) return;
    s = sm[t];
    float* dst;
    if (e < 3 * HC) dst = dw_obj + e;
    else if (e < 15 * HC) dst = dw_del + (e - 3 * HC);
    else {
        const int n = e - 15 * HC;
        if (n >= 15) return;
        dst = n < 3 ? db_obj + n : db_del + (n - 3);
    }
    if (dst == nullptr) return;
    *dst = accumulate ? *dst + s : s;
}

int fill_levels(HeadLevels& lv, const void* const* t, const void* const* y, const void* const* dy, const long long* pix, int nlev,
                bool groups_of_16) {
    if (nlev <= 0 || nlev > MAXLV) return OMNI_ERR_ARG;
    lv.nlev = nlev;
    lv.start[0] = 0;
    for (int l = 0; l < nlev; ++l) {
        if (pix[l] < 0) return OMNI_ERR_ARG;
        lv.t[l] = t ? (const float*)t[l] : nullptr;
        lv.y[l] = y ? (float*)y[l] : nullptr;
        lv.dy[l] = dy ? (const float*)dy[l] : nullptr;
        lv.pix[l] = pix[l];
        lv.start[l + 1] = lv.start[l] + (groups_of_16 ? (pix[l] + 15) / 16 : pix[l]);
    }
    return OMNI_OK;
}

}  // namespace

extern "C" {

// y_l (P_l, 16) = t_l (P_l, 256) . [w_obj (3, 256) | w_del (12, 256) | 0]^T + [b_obj | b_del | 0] for every level in one launch.
// t / y: HOST arrays of nlev device pointers, pix: HOST array of pixel counts P_l = B * H_l * W_l.
int omni_rpn_head16_fwd(const void* const* t, const long long* pix, int nlev, const float* w_obj, const float* b_obj, const float* w_del,
                        const float* b_del, const void* const* y, void* stream) {
    HeadLevels lv;
    if (w_obj == nullptr || w_del == nullptr || b_obj == nullptr || b_del == nullptr || t == nullptr || y == nullptr) return OMNI_ERR_ARG;
    if (fill_levels(lv, t, y, nullptr, pix, nlev, true) != OMNI_OK) return OMNI_ERR_ARG;
    const long groups = lv.start[nlev];
    if (groups == 0) return OMNI_OK;
    // ~2-3 pixel groups per wave: the 16 KB of weights a wave pulls into registers are amortised without starving the 256 CUs
    long wgs = (groups + 11) / 12;
    if (wgs < 1) wgs = 1;
    if (wgs > 2048) wgs = 2048;
    hipLaunchKernelGGL(head16_fwd_kernel, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, lv, w_obj, b_obj, w_del, b_del);
    return omni_launch_status();
}

// dt_l (P_l, 256) = (dy_l (P_l, 16) . [w_obj | w_del | 0]) masked by t_l > 0 when relu_mask != 0 (t = the ReLU output the head read).
int omni_rpn_head16_dgrad(const void* const* dy, const void* const* t, const long long* pix, int nlev, const float* w_obj,
                          const float* w_del, int relu_mask, const void* const* dt, void* stream) {
    HeadLevels lv;
    if (w_obj == nullptr || w_del == nullptr || dy == nullptr || dt == nullptr || (relu_mask && t == nullptr)) return OMNI_ERR_ARG;
    if (fill_levels(lv, t, dt, dy, pix, nlev, false) != OMNI_OK) return OMNI_ERR_ARG;
    const long total = lv.start[nlev];
    if (total == 0) return OMNI_OK;
    long wgs = (total + 63) / 64;          // 16 pixels per wave
    if (wgs < 1) wgs = 1;
    if (wgs > 2048) wgs = 2048;
    hipLaunchKernelGGL(head16_dgrad_kernel, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, lv, w_obj, w_del, relu_mask);
    return omni_launch_status();
}

// dw_obj (3, 256), db_obj (3), dw_del (12, 256), db_del (12) [each nullable] = / += the sums over all pixels of all levels.
// partial: scratch of partial_rows * (15 * 256 + 16) floats (one row per workgroup; 512 rows fill the chip).
int omni_rpn_head16_wgrad(const void* const* dy, const void* const* t, const long long* pix, int nlev, float* partial, int partial_rows,
                          float* dw_obj, float* db_obj, float* dw_del, float* db_del, int accumulate, void* stream) {
    HeadLevels lv;
    if (dy == nullptr || t == nullptr || partial == nullptr || partial_rows < 1) return OMNI_ERR_ARG;
    if (fill_levels(lv, t, nullptr, dy, pix, nlev, false) != OMNI_OK) return OMNI_ERR_ARG;
    const long total = lv.start[nlev];
    long wgs = (total + 127) / 128;        // >= 32 pixels per wave
    if (wgs < 1) wgs = 1;
    if (wgs > partial_rows) wgs = partial_rows;
    hipLaunchKernelGGL(head16_wgrad_kernel, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, lv, partial);
    hipLaunchKernelGGL(head16_wgrad_finalize_kernel, dim3((WG_PARTIAL + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       (const float*)partial, (int)wgs, dw_obj, db_obj, dw_del, db_del, accumulate);
    return omni_launch_status();
}

}  // extern "C"
