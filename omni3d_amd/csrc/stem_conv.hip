// stem_conv.hip -- direct convolution for the full-resolution, few-channel layers of the DLA-34 stem
//   base_layer  7x7  3(4) -> 16 @512x512     /root/reference/cubercnn/modeling/backbone/dla.py:241-245
//   level0      3x3 16    -> 16 @512x512     /root/reference/cubercnn/modeling/backbone/dla.py:246-247,279-289
// (forward, and the data gradient of level0 = the same convolution with the rotated, channel-transposed filter).
//
// Why not the implicit GEMM of conv_gemm.hip: with C <= 16 a k-slab is one filter tap, so the im2col formulation
// re-stages every input pixel R*S times through registers -> LDS and pads N = 16 output channels to 32; these layers
// then run at 20-35 TFLOP/s while their HBM floor is ~30 us.  Here one workgroup owns a 4 x 64 output tile, stages its
// input halo ((4+R-1) x (64+R-1) pixels) into LDS ONCE, keeps the whole filter in LDS, and every wave walks the taps with
// v_mfma_f32_16x16x4_f32 (M = 16 pixels, N = 16 output channels exactly, k = 4):
//   C = 16: one tap = 4 k-steps; a lane's ds_read_b128 holds channels 4g..4g+3 of its pixel and feeds MFMA t with channel
//           4g+t (k index = lane group g), the filter fragment is read the same way -> 5 LDS reads per 16 MFMAs.
//   C = 4 : a group of 4 horizontally adjacent taps = 4 k-steps; lane group g reads the pixel shifted by g taps, MFMA t
//           takes channel t, i.e. its k dimension runs over the 4 taps (rows of 7 taps are zero-padded to 8 in LDS).
// Any k order gives the same sum up to fp32 rounding.
#include <device_rt.h>

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int C, int R>
__global__ void __launch_bounds__(256) stem_conv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            float* __restrict__ out, int N, int H, int W, int ldx, int ldo,
                                                            float* __restrict__ stats) {
    constexpr int TH = 4, TW = 64, PAD = R / 2;
    constexpr int HR = TH + R - 1, HC = TW + R - 1 + (C == 4 ? 1 : 0);   // C = 4: one spare column for the padded 8th tap
    constexpr int PP = (C == 16) ? 20 : 4;                               // LDS floats per pixel (20: conflict-free b128 reads)
    constexpr int SP = (C == 16) ? R : 8;                                // taps per filter row held in LDS
    constexpr int KD = R * SP * C;                                       // LDS filter row length
    constexpr int KP = KD + ((KD / 4) % 2 == 0 ? 4 : 8);                 // odd number of 16-byte chunks per row
    __shared__ __attribute__((aligned(16))) float s_in[HR * HC * PP];
    __shared__ __attribute__((aligned(16))) float s_w[16 * KP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int n = t / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;

    // ---- stage the input halo (zero outside the image) and the filter ----
    // All global loads of a thread are issued before the first LDS store, explicitly (round 3; hipcc had already hoisted them out
    // of the load -> store loop: 68 -> 66 us on the level0 shape, i.e. the staging phase is not what keeps this kernel at 0.47 of
    // the fp32-MFMA peak against a ridge-point floor of ~31 us).
    constexpr int C4 = C / 4;
    constexpr int NI = (HR * HC * C4 + 255) / 256, NW = (16 * R * SP * C4 + 255) / 256;
    float4 vi[NI], vw[NW];
#pragma unroll
    for (int u = 0; u < NI; ++u) {
        const int i = tid + 256 * u;
        const int c4 = i % C4, col = (i / C4) % HC, row = i / (C4 * HC);
        const int iy = oy0 - PAD + row, ix = ox0 - PAD + col;
        vi[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < HR * HC * C4 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
            vi[u] = ld4(x + (((long)n * H + iy) * W + ix) * ldx + 4 * c4);
    }
#pragma unroll
    for (int u = 0; u < NW; ++u) {
        const int i = tid + 256 * u;
        const int c4 = i % C4, s = (i / C4) % SP, r = (i / (C4 * SP)) % R, k = i / (C4 * SP * R);
        vw[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < 16 * R * SP * C4 && s < R) vw[u] = ld4(w + (((long)k * R + r) * R + s) * C + 4 * c4);
    }
#pragma unroll
    for (int u = 0; u < NI; ++u) {
        const int i = tid + 256 * u;
        const int c4 = i % C4, col = (i / C4) % HC, row = i / (C4 * HC);
        if (i < HR * HC * C4) *reinterpret_cast<float4*>(s_in + (row * HC + col) * PP + 4 * c4) = vi[u];
    }
#pragma unroll
    for (int u = 0; u < NW; ++u) {
        const int i = tid + 256 * u;
        const int c4 = i % C4, s = (i / C4) % SP, r = (i / (C4 * SP)) % R, k = i / (C4 * SP * R);
        if (i < 16 * R * SP * C4) *reinterpret_cast<float4*>(s_w + k * KP + (r * SP + s) * C + 4 * c4) = vw[u];
    }
    __syncthreads();

    const int px = lane & 15, g = lane >> 4;
    f32x4 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (C == 16) {
#pragma unroll 1
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int s = 0; s < R; ++s) {
                const float4 bw = *reinterpret_cast<const float4*>(s_w + px * KP + (r * SP + s) * C + 4 * g);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float4 a = *reinterpret_cast<const float4*>(s_in + ((wave + r) * HC + 16 * b + px + s) * PP + 4 * g);
                    acc[b] = mfma_16x16x4(a.x, bw.x, acc[b]);
                    acc[b] = mfma_16x16x4(a.y, bw.y, acc[b]);
                    acc[b] = mfma_16x16x4(a.z, bw.z, acc[b]);
                    acc[b] = mfma_16x16x4(a.w, bw.w, acc[b]);
                }
            }
    } else {
#pragma unroll 1
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int sg = 0; sg < SP / 4; ++sg) {
                const float4 bw = *reinterpret_cast<const float4*>(s_w + px * KP + (r * SP + 4 * sg + g) * C);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float4 a = *reinterpret_cast<const float4*>(s_in + ((wave + r) * HC + 16 * b + px + 4 * sg + g) * PP);
                    acc[b] = mfma_16x16x4(a.x, bw.x, acc[b]);
                    acc[b] = mfma_16x16x4(a.y, bw.y, acc[b]);
                    acc[b] = mfma_16x16x4(a.z, bw.z, acc[b]);
                    acc[b] = mfma_16x16x4(a.w, bw.w, acc[b]);
                }
            }
    }
    // D[row = 4g + i][col = px]: output pixel ox0 + 16b + 4g + i of row oy0 + wave, channel px
    const int oy = oy0 + wave;
    float sv = 0.f, sq = 0.f;
    if (oy < H) {
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ox = ox0 + 16 * b + 4 * g + i;
                if (ox < W) {
                    out[(((long)n * H + oy) * W + ox) * ldo + px] = acc[b][i];
                    sv += acc[b][i];
                    sq += acc[b][i] * acc[b][i];
                }
            }
    }
    if (stats != nullptr) {     // BatchNorm statistics of the 16 output channels over this 4 x 64 tile -> stats[blockIdx][2][16]
        sv += __shfl_xor(sv, 16, 64); sq += __shfl_xor(sq, 16, 64);
        sv += __shfl_xor(sv, 32, 64); sq += __shfl_xor(sq, 32, 64);
        __syncthreads();                                   // s_in is free now
        if (lane < 16) { s_in[wave * 32 + lane] = sv; s_in[wave * 32 + 16 + lane] = sq; }
        __syncthreads();
        if (tid < 32) stats[(long)blockIdx.x * 32 + tid] = s_in[tid] + s_in[32 + tid] + s_in[64 + tid] + s_in[96 + tid];
    }
}


// ---- weight gradient: dW[k][r][s][c] += sum over output pixels of dy[pix][k] * x[pix + (r, s) - pad][c] -----------------
// MFMA 16x16x4 with M = the 16 output channels, N = 16 columns of (tap, c) and the k dimension running over 4 consecutive
// output pixels of a row: A[k][j] = dy[pix0 + j][k], B[j][col] = x[pix0 + j + tap(col)][c(col)].
//   C = 16: one N block per tap (9 accumulator blocks);  C = 4: one block per 4 horizontally adjacent taps (rows of 7 taps
//   padded to 8: 14 blocks) -- in both cases the 16 lanes of a k group read 16 consecutive floats of the LDS halo.
// Persistent workgroups loop over 4 x 64 pixel tiles and keep the accumulators in registers; every wave adds its partial
// filter gradient to dW with fp32 atomics at the end (dW is the flat gradient bucket or a zeroed buffer).
template <int C, int R>
// part != nullptr (deterministic mode, round 4): every wave stores its partial filter gradient as row (4 * blockIdx.x + wave) of
// part[4 * gridDim.x][16 * R * R * C] instead; stem_wgrad_finalize_kernel adds the rows in a fixed order.
__global__ void __launch_bounds__(256) stem_conv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              float* __restrict__ dw, int N, int H, int W, int ldx, int lddy,
                                                              float* __restrict__ part) {
    constexpr int TH = 4, TW = 64, PAD = R / 2;
    constexpr int HR = TH + R - 1, HC = TW + R - 1 + (C == 4 ? 1 : 0);
    constexpr int NB = (C == 16) ? R * R : R * 2;            // accumulator blocks
    __shared__ __attribute__((aligned(16))) float s_in[HR * HC * C];
    __shared__ __attribute__((aligned(16))) float s_dy[TH * TW * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, g = lane >> 4;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int tiles = N * tiles_y * tiles_x;
    f32x4 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int C4 = C / 4;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        int t = tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        const int n = t / tiles_y;
        const int oy0 = ty * TH, ox0 = tx * TW;
        // all global loads of the tile before the first LDS store (see stem_conv_fwd_kernel); they are issued BEFORE the barrier
        // that frees the LDS tiles, so they fly while the other waves finish the previous tile's MFMAs
        constexpr int NI = (HR * HC * C4 + 255) / 256, ND = (TH * TW * 4 + 255) / 256;
        float4 vi[NI], vd[ND];
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int i = tid + 256 * u;
            const int c4 = i % C4, cc = (i / C4) % HC, row = i / (C4 * HC);
            const int iy = oy0 - PAD + row, ix = ox0 - PAD + cc;
            vi[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < HR * HC * C4 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                vi[u] = ld4(x + (((long)n * H + iy) * W + ix) * ldx + 4 * c4);
        }
#pragma unroll
        for (int u = 0; u < ND; ++u) {
            const int i = tid + 256 * u;
            const int k4 = i % 4, cc = (i / 4) % TW, row = i / (4 * TW);
            const int oy = oy0 + row, ox = ox0 + cc;
            vd[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < TH * TW * 4 && oy < H && ox < W) vd[u] = ld4(dy + (((long)n * H + oy) * W + ox) * lddy + 4 * k4);
        }
        __syncthreads();                                      // previous tile's LDS reads are done
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int i = tid + 256 * u;
            const int c4 = i % C4, cc = (i / C4) % HC, row = i / (C4 * HC);
            if (i < HR * HC * C4) *reinterpret_cast<float4*>(s_in + (row * HC + cc) * C + 4 * c4) = vi[u];
        }
#pragma unroll
        for (int u = 0; u < ND; ++u) {
            const int i = tid + 256 * u;
            const int k4 = i % 4, cc = (i / 4) % TW, row = i / (4 * TW);
            if (i < TH * TW * 4) *reinterpret_cast<float4*>(s_dy + (row * TW + cc) * 16 + 4 * k4) = vd[u];
        }
        __syncthreads();
#pragma unroll 1
        for (int p0 = 0; p0 < TW; p0 += 4) {                  // 4 output pixels of row `wave` per k group
            const float a = s_dy[(wave * TW + p0 + g) * 16 + col];
            if (C == 16) {
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int sx = 0; sx < R; ++sx)
                        acc[r * R + sx] = mfma_16x16x4(a, s_in[((wave + r) * HC + p0 + g + sx) * C + col], acc[r * R + sx]);
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int blk = 0; blk < 2; ++blk)
                        acc[r * 2 + blk] = mfma_16x16x4(a, s_in[((wave + r) * HC + p0 + g + 4 * blk) * C + col], acc[r * 2 + blk]);
            }
        }
    }
    // D[row = 4g + i][col]: k = 4g + i;  C = 16: (tap = block, c = col);  C = 4: (s = 4 blk + col / 4, c = col % 4)
    constexpr int KD = R * R * C;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        int off;
        bool ok = true;
        if (C == 16) {
            off = b * C + col;
        } else {
            const int r = b >> 1, sx = 4 * (b & 1) + (col >> 2);
            ok = sx < R;
            off = (r * R + sx) * C + (col & 3);
        }
        if (!ok) continue;
        if (part != nullptr) {
            float* row = part + ((long)blockIdx.x * 4 + wave) * (16 * KD);
#pragma unroll
            for (int i = 0; i < 4; ++i) row[(4 * g + i) * KD + off] = acc[b][i];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) atomicAdd(dw + (long)(4 * g + i) * KD + off, acc[b][i]);
        }
    }
}

// dw[e] (+)= sum over the `rows` partial rows of part[row][e], E elements: 4 elements per workgroup, 64 row groups per element,
// every thread adds its rows in ascending order (fp64), the 64 group sums meet in LDS in a fixed tree -- run-to-run identical.
__global__ void __launch_bounds__(256) stem_wgrad_finalize_kernel(const float* __restrict__ part, int rows, int E, float* __restrict__ dw,
                                                                  int accumulate) {
    __shared__ double sm[256];
    const int t = threadIdx.x, el = t & 3, rg = t >> 2;
    const int e = (int)blockIdx.x * 4 + el;
    double a = 0.0;
    if (e < E)
        for (int r = rg; r < rows; r += 64) a += (double)part[(long)r * E + e];
    sm[t] = a;
    __syncthreads();
    for (int s = 128; s >= 4; s >>= 1) {
        if (t < s) sm[t] += sm[t + s];
        __syncthreads();
    }
    if (t < 4 && e < E) dw[e] = accumulate ? dw[e] + (float)sm[t] : (float)sm[t];
}

}  // namespace

extern "C" {

// out (N,H,W,16) = conv(x (N,H,W,C), w (16,R,R,C)), stride 1, padding R/2.  (C, R) in {(4, 7), (16, 3)}.
static int stem_fwd_impl(const float* x, const float* w, float* out, int N, int H, int W, int C, int K, int R, int ldx, int ldo,
                         float* stats, int stats_rows, int* nblk_out, void* stream) {
    if (nblk_out) *nblk_out = 0;
    if (N < 0 || H <= 0 || W <= 0 || K != 16 || ldx < C || ldo < K || (ldx & 3)) return OMNI_ERR_ARG;
    if (!((C == 4 && R == 7) || (C == 16 && R == 3))) return OMNI_ERR_ARG;
    if (N == 0) return OMNI_OK;
    const long tiles = (long)N * ((H + 3) / 4) * ((W + 63) / 64);
    float* sp = (stats != nullptr && tiles <= stats_rows) ? stats : nullptr;
    if (sp && nblk_out) *nblk_out = (int)tiles;
    if (C == 4)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(stem_conv_fwd_kernel<4, 7>), dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, x, w,
                           out, N, H, W, ldx, ldo, sp);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(stem_conv_fwd_kernel<16, 3>), dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, x, w,
                           out, N, H, W, ldx, ldo, sp);
    return omni_launch_status();
}

int omni_stem_conv_fwd(const float* x, const float* w, float* out, int N, int H, int W, int C, int K, int R, int ldx, int ldo,
                       void* stream) {
    return stem_fwd_impl(x, w, out, N, H, W, C, K, R, ldx, ldo, nullptr, 0, nullptr, stream);
}

// the same with BatchNorm partial statistics stats[rows][2][16] of the output (one row per 4 x 64 tile); *nblk_out = rows (0 = none)
int omni_stem_conv_fwd_stats(const float* x, const float* w, float* out, int N, int H, int W, int C, int K, int R, int ldx, int ldo,
                             float* stats, int stats_rows, int* nblk_out, void* stream) {
    return stem_fwd_impl(x, w, out, N, H, W, C, K, R, ldx, ldo, stats, stats_rows, nblk_out, stream);
}

// dw (16,R,R,C) (+)= sum_pix dy (N,H,W,16) x (N,H,W,C); accumulate == 0 zeroes dw first (atomic accumulation either way).
static int stem_wgrad_impl(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int K, int R, int ldx, int lddy,
                           int accumulate, float* ws, long long ws_floats, long long* plan, bool det, void* stream) {
    if (N < 0 || H <= 0 || W <= 0 || K != 16 || ldx < C || lddy < K || (ldx & 3) || (lddy & 3)) return OMNI_ERR_ARG;
    if (!((C == 4 && R == 7) || (C == 16 && R == 3))) return OMNI_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    long tiles = (long)N * ((H + 3) / 4) * ((W + 63) / 64);
    const unsigned grid = (unsigned)(tiles < 768 ? tiles : 768);        // 3 resident workgroups per CU
    const int E = K * R * R * C;
    if (plan != nullptr) { plan[0] = plan[1] = plan[2] = 0; plan[3] = (long long)grid * 4 * E; return OMNI_OK; }
    if (det && N > 0 && (ws == nullptr || ws_floats < (long long)grid * 4 * E)) return OMNI_ERR_ARG;
    if (!accumulate && (!det || N == 0)) omni_memset_async(dw, 0, sizeof(float) * (size_t)E, st);
    if (N == 0) return OMNI_OK;
    float* part = det ? ws : nullptr;
    if (C == 4)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(stem_conv_wgrad_kernel<4, 7>), dim3(grid), dim3(256), 0, st, x, dy, dw, N, H, W, ldx, lddy, part);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(stem_conv_wgrad_kernel<16, 3>), dim3(grid), dim3(256), 0, st, x, dy, dw, N, H, W, ldx, lddy, part);
    if (det)
        hipLaunchKernelGGL(stem_wgrad_finalize_kernel, dim3((unsigned)((E + 3) / 4)), dim3(256), 0, st, (const float*)part, (int)grid * 4, E, dw,
                           accumulate);
    return omni_launch_status();
}

int omni_stem_conv_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int K, int R, int ldx, int lddy,
                         int accumulate, void* stream) {
    return stem_wgrad_impl(x, dy, dw, N, H, W, C, K, R, ldx, lddy, accumulate, nullptr, 0, nullptr, false, stream);
}

// deterministic form: per-wave partial filter gradients go to `ws` (ws_floats floats; plan != NULL: plan[3] = floats needed, no
// launch) and a second launch adds them in a fixed order into dw (overwritten, or added to when accumulate != 0); no atomics
int omni_stem_conv_wgrad_det(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int K, int R, int ldx, int lddy,
                             int accumulate, float* ws, long long ws_floats, long long* plan, void* stream) {
    return stem_wgrad_impl(x, dy, dw, N, H, W, C, K, R, ldx, lddy, accumulate, ws, ws_floats, plan, true, stream);
}

}  // extern "C"
