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

// ROT (C == 16 only): `w` is the FORWARD filter (16, R, R, 16) of a layer whose DATA GRADIENT this launch computes -- x is dy, out is
// dx, and the filter used is w rotated by 180 degrees with its channel roles swapped, wd[c][r][s][k] = w[k][R-1-r][R-1-s][c], built
// while the filter is staged (round 5: the flip + transpose + layout copy were three ATen launches, 23 us, for a 9 KB tensor).
template <int C, int R, bool ROT = false>
__global__ void __launch_bounds__(256) stem_conv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            float* __restrict__ out, int N, int H, int W, int ldx, int ldo,
                                                            float* __restrict__ stats, int cw) {
    static_assert(!ROT || C == 16, "the rotated form is the 16 -> 16 layer's data gradient");
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
        if (i < 16 * R * SP * C4 && s < R) {
            if (C != 4 || cw == C) {
                vw[u] = ld4(w + (((long)k * R + r) * R + s) * C + 4 * c4);     // (ROT too: w is read linearly)
            } else {      // round 6: the first layer's filter as the model holds it, (16, R, R, cw = 3); the image's 4th channel is zero padding
                const float* q = w + (((long)k * R + r) * R + s) * cw;
                vw[u].x = q[0];
                vw[u].y = cw > 1 ? q[1] : 0.f;
                vw[u].z = cw > 2 ? q[2] : 0.f;
            }
        }
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
        if (i < 16 * R * SP * C4) {
            if (ROT) {      // w[k][r][s][4 c4 + e] is element (row 4 c4 + e, tap (R-1-r, R-1-s), input channel k) of the rotated filter
                float* q = s_w + (4 * c4) * KP + ((R - 1 - r) * SP + (R - 1 - s)) * C + k;
                q[0] = vw[u].x; q[KP] = vw[u].y; q[2 * KP] = vw[u].z; q[3 * KP] = vw[u].w;
            } else {
                *reinterpret_cast<float4*>(s_w + k * KP + (r * SP + s) * C + 4 * c4) = vw[u];
            }
        }
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


// ---- data gradient of the stride-2 3x3 layer (DLA-34 level1: 16 -> 32 channels at 512 x 512, dla.py:248-249,291-295) ------------
//   dx[n, ih, iw, c] = sum over (r, s, k) with (ih + 1 - r), (iw + 1 - s) even of dy[n, (ih + 1 - r) / 2, (iw + 1 - s) / 2, k] * w[k, r, s, c]
// 2.4 GFLOP against 100 MB (33.5 MB of dy read, 67 MB of dx written): HBM floor 12.5 us, MFMA floor 15 us.  The implicit GEMM ran
// this as four parity-class launches-in-one on 256 x 32 tiles (N = 16 padded to 32, every dy pixel re-staged per tap): 118 us isolated,
// 177-217 us inside the step -- the longest single launch of the critical path after the fc1 GEMMs (gpurun_out/r05a_timeline.txt).
// Here a workgroup owns an 8 x 64 tile of dx, stages its 5 x 33 pixel halo of dy (all K channels) and the whole filter into LDS once
// and walks the taps of each PARITY CLASS with v_mfma_f32_16x16x4_f32 (M = 16 pixels of one class in one row, N = the 16 channels of
// dx exactly): even rows / columns meet one tap per axis (r = 1), odd ones two (r = 0, 2) -- no MFMA work on the structural zeros.
// A wave takes one even and one odd row of the tile (rows w and 7 - w), so the four waves carry the same number of MFMAs.
//   fragments: a lane's ds_read_b128 holds dy channels 16 h + 4 g .. + 3 of its pixel (A) / filter entries of the same channels for its
//   dx channel (B); MFMA t takes element t, i.e. the k index of the four MFMAs runs over the lane groups g (as in stem_conv_fwd_kernel).
template <int C, int K>
__global__ void __launch_bounds__(256) stem_dgrad_s2_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                            float* __restrict__ dx, int N, int H, int W, int OH, int OW, int lddy,
                                                            int lddx) {
    static_assert(C == 16 && K % 16 == 0, "N = 16 dx channels per MFMA block; dy channels in chunks of 16");
    constexpr int TH = 8, TW = 64, HR = TH / 2 + 1, HC = TW / 2 + 1, PP = K + 4, KP = K + 4, K4 = K / 4, KH = K / 16;
    __shared__ __attribute__((aligned(16))) float s_dy[HR * HC * PP];
    __shared__ __attribute__((aligned(16))) float s_w[9 * C * KP];          // [tap][c][k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int n = t / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;                                  // even
    const int hy0 = oy0 / 2, hx0 = ox0 / 2;
    // ---- stage: every global load before the first LDS store
    constexpr int NI = (HR * HC * K4 + 255) / 256, NW = (K * 9 * (C / 4) + 255) / 256;
    float4 vi[NI], vw[NW];
#pragma unroll
    for (int u = 0; u < NI; ++u) {
        const int i = tid + 256 * u;
        const int k4 = i % K4, col = (i / K4) % HC, row = i / (K4 * HC);
        const int oy = hy0 + row, ox = hx0 + col;
        vi[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < HR * HC * K4 && oy < OH && ox < OW) vi[u] = ld4(dy + (((long)n * OH + oy) * OW + ox) * lddy + 4 * k4);
    }
#pragma unroll
    for (int u = 0; u < NW; ++u) {
        const int i = tid + 256 * u;
        vw[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < K * 9 * (C / 4)) vw[u] = ld4(w + 4L * i);                   // w (K, 3, 3, C) read linearly
    }
#pragma unroll
    for (int u = 0; u < NI; ++u) {
        const int i = tid + 256 * u;
        const int k4 = i % K4, col = (i / K4) % HC, row = i / (K4 * HC);
        if (i < HR * HC * K4) *reinterpret_cast<float4*>(s_dy + (row * HC + col) * PP + 4 * k4) = vi[u];
    }
#pragma unroll
    for (int u = 0; u < NW; ++u) {
        const int i = tid + 256 * u;
        if (i < K * 9 * (C / 4)) {
            const int c4 = i % (C / 4), tap = (i / (C / 4)) % 9, k = i / (9 * (C / 4));
            float* q = s_w + (tap * C + 4 * c4) * KP + k;
            q[0] = vw[u].x; q[KP] = vw[u].y; q[2 * KP] = vw[u].z; q[3 * KP] = vw[u].w;
        }
    }
    __syncthreads();

    const int px = lane & 15, g = lane >> 4;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        const int y = half == 0 ? wave : TH - 1 - wave;                      // tile row of dx; parity = y & 1 (oy0 is even)
        const int ih = oy0 + y;
        const int nr = 1 + (y & 1);                                          // row taps: even: r = 1 | odd: r = 0, 2
        f32x4 acc[2][2];                                                     // [column parity][block of 16 class pixels]
#pragma unroll
        for (int pw = 0; pw < 2; ++pw)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[pw][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int jr = 0; jr < nr; ++jr) {
            const int r = (y & 1) ? 2 * jr : 1;
            const int lr = (y + 1 - r) >> 1;                                 // halo row of dy
#pragma unroll
            for (int pw = 0; pw < 2; ++pw)
#pragma unroll
                for (int js = 0; js <= pw; ++js) {                           // column taps: even: s = 1 | odd: s = 0 (dc 1), s = 2 (dc 0)
                    const int sx = pw ? 2 * js : 1, dc = pw ? 1 - js : 0;
#pragma unroll
                    for (int h2 = 0; h2 < KH; ++h2) {
                        const float4 bw = *reinterpret_cast<const float4*>(s_w + ((r * 3 + sx) * C + px) * KP + 16 * h2 + 4 * g);
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            const float4 a = *reinterpret_cast<const float4*>(s_dy + (lr * HC + 16 * b + px + dc) * PP + 16 * h2 + 4 * g);
                            acc[pw][b] = mfma_16x16x4(a.x, bw.x, acc[pw][b]);
                            acc[pw][b] = mfma_16x16x4(a.y, bw.y, acc[pw][b]);
                            acc[pw][b] = mfma_16x16x4(a.z, bw.z, acc[pw][b]);
                            acc[pw][b] = mfma_16x16x4(a.w, bw.w, acc[pw][b]);
                        }
                    }
                }
        }
        // D[row = 4g + i][col = px]: class pixel j = 16 b + 4 g + i of this row -> dx column ox0 + 2 j + pw, channel px
        if (ih < H) {
            float* orow = dx + ((long)n * H + ih) * W * lddx + px;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int pw = 0; pw < 2; ++pw) {
                        const int iw = ox0 + 2 * (16 * b + 4 * g + i) + pw;
                        if (iw < W) orow[(long)iw * lddx] = acc[pw][b][i];
                    }
        }
    }
}

// ---- weight gradient: dW[k][r][s][c] += sum over output pixels of dy[pix][k] * x[S * pix + (r, s) - pad][c] -------------
// MFMA 16x16x4 with M = 16 output channels, N = 16 columns of (tap, c) and the k dimension running over 4 consecutive
// output pixels of a row: A[k][j] = dy[pix0 + j][k], B[j][col] = x[S * (pix0 + j) + tap(col)][c(col)].
//   C = 16: one N block per tap (9 accumulator blocks per 16 output channels);  C = 4: one block per 4 horizontally adjacent
//   taps (rows of 7 taps padded to 8: 14 blocks) -- in both cases the 16 lanes of a k group read 16 consecutive floats of the
//   LDS halo.
// (C, R, S, K) = (4, 7, 1, 16) base_layer | (16, 3, 1, 16) level0 | (16, 3, 2, 32) level1 (round 4: the stride-2 layer ran on the
//   implicit GEMM's 32 x 128 tiles cut 512 ways over the pixels, 197 us for 2.4 GFLOP and 100 MB).
// Persistent workgroups loop over TH x TW output tiles and keep the accumulators in registers.  At the end the four waves add
// their partial filter gradients in LDS, wave 0 first, and the workgroup either stores ONE partial row (part != nullptr:
// deterministic mode, stem_wgrad_finalize_kernel adds the rows in a fixed order) or adds it to dw with fp32 atomics.
template <int C, int R, int S, int K>
__global__ void __launch_bounds__(256) stem_conv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              float* __restrict__ dw, int N, int H, int W, int OH, int OW, int ldx,
                                                              int lddy, float* __restrict__ part) {
    constexpr int TH = 4, TW = (S == 1) ? 64 : 32, PAD = R / 2, KB = K / 16, K4 = K / 4;
    constexpr int HR = S * (TH - 1) + R, HC = S * (TW - 1) + R + (C == 4 ? 1 : 0);
    constexpr int NB = (C == 16) ? R * R : R * 2;            // accumulator blocks per 16 output channels
    constexpr int KD = R * R * C, E = K * KD;
    constexpr int IN_FLOATS = HR * HC * C, DY_FLOATS = TH * TW * K;
    static_assert(E <= IN_FLOATS + DY_FLOATS, "the filter gradient is summed over the waves in the tile buffers");
    static_assert(C == 16 || S == 1, "the 4-channel form walks 4 adjacent taps per block");
    __shared__ __attribute__((aligned(16))) float smem[IN_FLOATS + DY_FLOATS];
    float* s_in = smem;
    float* s_dy = smem + IN_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, g = lane >> 4;
    const int tiles_x = (OW + TW - 1) / TW, tiles_y = (OH + TH - 1) / TH;
    const int tiles = N * tiles_y * tiles_x;
    f32x4 acc[KB][NB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[kb][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int C4 = C / 4;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        int t = tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        const int n = t / tiles_y;
        const int oy0 = ty * TH, ox0 = tx * TW;
        // all global loads of the tile before the first LDS store (see stem_conv_fwd_kernel); they are issued BEFORE the barrier
        // that frees the LDS tiles, so they fly while the other waves finish the previous tile's MFMAs
        constexpr int NI = (HR * HC * C4 + 255) / 256, ND = (TH * TW * K4 + 255) / 256;
        float4 vi[NI], vd[ND];
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int i = tid + 256 * u;
            const int c4 = i % C4, cc = (i / C4) % HC, row = i / (C4 * HC);
            const int iy = S * oy0 - PAD + row, ix = S * ox0 - PAD + cc;
            vi[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < HR * HC * C4 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                vi[u] = ld4(x + (((long)n * H + iy) * W + ix) * ldx + 4 * c4);
        }
#pragma unroll
        for (int u = 0; u < ND; ++u) {
            const int i = tid + 256 * u;
            const int k4 = i % K4, cc = (i / K4) % TW, row = i / (K4 * TW);
            const int oy = oy0 + row, ox = ox0 + cc;
            vd[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < TH * TW * K4 && oy < OH && ox < OW) vd[u] = ld4(dy + (((long)n * OH + oy) * OW + ox) * lddy + 4 * k4);
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
            const int k4 = i % K4, cc = (i / K4) % TW, row = i / (K4 * TW);
            if (i < TH * TW * K4) *reinterpret_cast<float4*>(s_dy + (row * TW + cc) * K + 4 * k4) = vd[u];
        }
        __syncthreads();
#pragma unroll 1
        for (int p0 = 0; p0 < TW; p0 += 4) {                  // 4 output pixels of row `wave` per k group
            float a[KB];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) a[kb] = s_dy[(wave * TW + p0 + g) * K + kb * 16 + col];
            if (C == 16) {
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int sx = 0; sx < R; ++sx) {
                        const float b = s_in[((S * wave + r) * HC + S * (p0 + g) + sx) * C + col];
#pragma unroll
                        for (int kb = 0; kb < KB; ++kb) acc[kb][r * R + sx] = mfma_16x16x4(a[kb], b, acc[kb][r * R + sx]);
                    }
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int blk = 0; blk < 2; ++blk) {
                        const float b = s_in[((wave + r) * HC + p0 + g + 4 * blk) * C + col];
#pragma unroll
                        for (int kb = 0; kb < KB; ++kb) acc[kb][r * 2 + blk] = mfma_16x16x4(a[kb], b, acc[kb][r * 2 + blk]);
                    }
            }
        }
    }
    // D[row = 4g + i][col]: k = 16 kb + 4g + i;  C = 16: (tap = block, c = col);  C = 4: (s = 4 blk + col / 4, c = col % 4).
    // The four waves add into smem[E] one after the other (every element belongs to the same lane in each wave).
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
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
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float* q = smem + (kb * 16 + 4 * g + i) * KD + off;
                        *q = (w == 0) ? acc[kb][b][i] : *q + acc[kb][b][i];
                    }
                }
        }
    }
    __syncthreads();
    if (part != nullptr) {
        float* row = part + (long)blockIdx.x * E;
        for (int e = tid; e < E; e += 256) row[e] = smem[e];
    } else {
        for (int e = tid; e < E; e += 256) atomicAdd(dw + e, smem[e]);
    }
}

// dw[e] (+)= sum over the `rows` partial rows of part[row][e], E elements: 4 elements per workgroup, 64 row groups per element,
// every thread adds its rows in ascending order (fp64), the 64 group sums meet in LDS in a fixed tree -- run-to-run identical.
// cw < C: dw holds cw channels per tap ((K, R, R, cw), the first layer's 3-channel filter gradient) -- the partial rows' padded channels
// (exact zeros: the image's padding channel) are dropped here instead of by a slice + add behind the kernel
__global__ void __launch_bounds__(256) stem_wgrad_finalize_kernel(const float* __restrict__ part, int rows, int E, float* __restrict__ dw,
                                                                  int accumulate, int C, int cw) {
    __shared__ double sm[256];
    const int t = threadIdx.x, el = t & 3, rg = t >> 2;
    const int e = (int)blockIdx.x * 4 + el;
    double a = 0.0;
    if (e < E) {
        int r = rg;
        for (; r + 192 < rows; r += 256) {       // four independent loads in flight: this loop is pure latency
            const float v0 = part[(long)r * E + e], v1 = part[(long)(r + 64) * E + e], v2 = part[(long)(r + 128) * E + e],
                        v3 = part[(long)(r + 192) * E + e];
            a += (double)v0;
            a += (double)v1;
            a += (double)v2;
            a += (double)v3;
        }
        for (; r < rows; r += 64) a += (double)part[(long)r * E + e];
    }
    sm[t] = a;
    __syncthreads();
    for (int s = 128; s >= 4; s >>= 1) {
        if (t < s) sm[t] += sm[t + s];
        __syncthreads();
    }
    if (t < 4 && e < E) {
        const int c = e % C;
        if (c < cw) {
            float* d = dw + (long)(e / C) * cw + c;
            *d = accumulate ? *d + (float)sm[t] : (float)sm[t];
        }
    }
}

}  // namespace

extern "C" {

// out (N,H,W,16) = conv(x (N,H,W,C), w (16,R,R,C)), stride 1, padding R/2.  (C, R) in {(4, 7), (16, 3)}.
static int stem_fwd_impl(const float* x, const float* w, float* out, int N, int H, int W, int C, int K, int R, int ldx, int ldo,
                         float* stats, int stats_rows, int* nblk_out, void* stream, bool rot = false, int cw = 0) {
    if (nblk_out) *nblk_out = 0;
    if (cw == 0) cw = C;
    if (N < 0 || H <= 0 || W <= 0 || K != 16 || ldx < C || ldo < K || (ldx & 3) || (rot && C != 16)) return OMNI_ERR_ARG;
    if (cw != C && (C != 4 || cw < 1 || cw > 4)) return OMNI_ERR_ARG;
    if (!((C == 4 && R == 7) || (C == 16 && R == 3))) return OMNI_ERR_ARG;
    if (N == 0) return OMNI_OK;
    const long tiles = (long)N * ((H + 3) / 4) * ((W + 63) / 64);
    float* sp = (stats != nullptr && tiles <= stats_rows) ? stats : nullptr;
    if (sp && nblk_out) *nblk_out = (int)tiles;
    if (C == 4)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(stem_conv_fwd_kernel<4, 7>), dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, x, w,
                           out, N, H, W, ldx, ldo, sp, cw);
    else if (rot)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(stem_conv_fwd_kernel<16, 3, true>), dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, x, w,
                           out, N, H, W, ldx, ldo, sp, cw);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(stem_conv_fwd_kernel<16, 3>), dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, x, w,
                           out, N, H, W, ldx, ldo, sp, cw);
    return omni_launch_status();
}

int omni_stem_conv_fwd(const float* x, const float* w, float* out, int N, int H, int W, int C, int K, int R, int ldx, int ldo,
                       void* stream) {
    return stem_fwd_impl(x, w, out, N, H, W, C, K, R, ldx, ldo, nullptr, 0, nullptr, stream);
}

// data gradient of the 3x3 16 -> 16 stride-1 layer: dx (N,H,W,16) from dy (N,H,W,16) and the layer's FORWARD filter w (16,3,3,16); the
// rotated, channel-transposed filter is formed while it is staged (no temporary)
int omni_stem_conv_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int K, int R, int lddy, int lddx,
                         void* stream) {
    if (C != 16 || K != 16 || R != 3) return OMNI_ERR_ARG;
    return stem_fwd_impl(dy, w, dx, N, H, W, K, C, R, lddy, lddx, nullptr, 0, nullptr, stream, true);
}

// data gradient of the 3x3 stride-2 pad-1 layer 16 -> 32 (DLA-34 level1): dx (N,H,W,16) from dy (N,OH,OW,32), OH = (H - 1) / 2 + 1,
// and w (32,3,3,16).  Every element of dx is written (no accumulation form: the layer's input has one consumer).
int omni_stem_conv_s2_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int K, int R, int lddy, int lddx,
                            void* stream) {
    if (N < 0 || H <= 0 || W <= 0 || C != 16 || K != 32 || R != 3 || lddy < K || lddx < C || (lddy & 3)) return OMNI_ERR_ARG;
    if (N == 0) return OMNI_OK;
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const long tiles = (long)N * ((H + 7) / 8) * ((W + 63) / 64);
    if (tiles > 0x7fffffff) return OMNI_ERR_ARG;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(stem_dgrad_s2_kernel<16, 32>), dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, dy, w, dx, N,
                       H, W, OH, OW, lddy, lddx);
    return omni_launch_status();
}

// the same with BatchNorm partial statistics stats[rows][2][16] of the output (one row per 4 x 64 tile); *nblk_out = rows (0 = none)
int omni_stem_conv_fwd_stats(const float* x, const float* w, float* out, int N, int H, int W, int C, int K, int R, int ldx, int ldo,
                             float* stats, int stats_rows, int* nblk_out, void* stream) {
    return stem_fwd_impl(x, w, out, N, H, W, C, K, R, ldx, ldo, stats, stats_rows, nblk_out, stream);
}

// dw (K,R,R,C) (+)= sum_pix dy (N,OH,OW,K) x (N,H,W,C); accumulate == 0 zeroes dw first (atomic accumulation either way).
// (C, R, stride, K) in {(4, 7, 1, 16), (16, 3, 1, 16), (16, 3, 2, 32)}; stride 2: H, W even, OH = H / 2.
static int stem_wgrad_impl(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int K, int R, int stride, int ldx,
                           int lddy, int accumulate, float* ws, long long ws_floats, long long* plan, bool det, void* stream, int cw = 0) {
    if (cw == 0) cw = C;
    if (N < 0 || H <= 0 || W <= 0 || ldx < C || lddy < K || (ldx & 3) || (lddy & 3)) return OMNI_ERR_ARG;
    if (cw != C && (!det || C != 4 || cw < 1 || cw > 4)) return OMNI_ERR_ARG;      // (the narrow filter gradient is written by the finalize launch)
    const int form = (C == 4 && R == 7 && stride == 1 && K == 16) ? 0 : (C == 16 && R == 3 && stride == 1 && K == 16) ? 1
                   : (C == 16 && R == 3 && stride == 2 && K == 32 && !(H & 1) && !(W & 1)) ? 2 : -1;
    if (form < 0) return OMNI_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int OH = H / stride, OW = W / stride, TW = stride == 1 ? 64 : 32;
    const long tiles = (long)N * ((OH + 3) / 4) * ((OW + TW - 1) / TW);
    // <= 3 resident workgroups per CU, every one with the same number of tiles (+- 1)
    const long per = (tiles + 767) / 768;
    const unsigned grid = (unsigned)(tiles <= 768 ? tiles : (tiles + per - 1) / per);
    const int E = K * R * R * C;
    if (plan != nullptr) { plan[0] = plan[1] = plan[2] = 0; plan[3] = (long long)grid * E; return OMNI_OK; }
    if (det && N > 0 && (ws == nullptr || ws_floats < (long long)grid * E)) return OMNI_ERR_ARG;
    if (!accumulate && (!det || N == 0)) omni_memset_async(dw, 0, sizeof(float) * (size_t)(E / C * cw), st);
    if (N == 0) return OMNI_OK;
    float* part = det ? ws : nullptr;
    if (form == 0)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(stem_conv_wgrad_kernel<4, 7, 1, 16>), dim3(grid), dim3(256), 0, st, x, dy, dw, N, H, W, OH, OW, ldx, lddy, part);
    else if (form == 1)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(stem_conv_wgrad_kernel<16, 3, 1, 16>), dim3(grid), dim3(256), 0, st, x, dy, dw, N, H, W, OH, OW, ldx, lddy, part);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(stem_conv_wgrad_kernel<16, 3, 2, 32>), dim3(grid), dim3(256), 0, st, x, dy, dw, N, H, W, OH, OW, ldx, lddy, part);
    if (det)
        hipLaunchKernelGGL(stem_wgrad_finalize_kernel, dim3((unsigned)((E + 3) / 4)), dim3(256), 0, st, (const float*)part, (int)grid, E, dw,
                           accumulate, C, cw);
    return omni_launch_status();
}

int omni_stem_conv_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int K, int R, int ldx, int lddy,
                         int accumulate, void* stream) {
    return stem_wgrad_impl(x, dy, dw, N, H, W, C, K, R, 1, ldx, lddy, accumulate, nullptr, 0, nullptr, false, stream);
}

// deterministic form: per-workgroup partial filter gradients go to `ws` (ws_floats floats; plan != NULL: plan[3] = floats needed, no
// launch) and a second launch adds them in a fixed order into dw (overwritten, or added to when accumulate != 0); no atomics
int omni_stem_conv_wgrad_det(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int K, int R, int ldx, int lddy,
                             int accumulate, float* ws, long long ws_floats, long long* plan, void* stream) {
    return stem_wgrad_impl(x, dy, dw, N, H, W, C, K, R, 1, ldx, lddy, accumulate, ws, ws_floats, plan, true, stream);
}

// Round 6: the first layer (dla.py:241-245: 7x7, 3 -> 16) against the 4-channel padded image with the filter AS THE MODEL HOLDS IT,
// w / dw (16, R, R, cw) with cw = 3: the kernels read / write the narrow layout themselves -- the padded copy of the filter in front
// of the forward and the slice + add of its gradient behind the backward (the last launch of the step's critical path) are gone.
int omni_stem_conv_fwd_cw(const float* x, const float* w, int cw, float* out, int N, int H, int W, int C, int K, int R, int ldx, int ldo,
                          float* stats, int stats_rows, int* nblk_out, void* stream) {
    return stem_fwd_impl(x, w, out, N, H, W, C, K, R, ldx, ldo, stats, stats_rows, nblk_out, stream, false, cw);
}
int omni_stem_conv_wgrad_det_cw(const float* x, const float* dy, float* dw, int cw, int N, int H, int W, int C, int K, int R, int ldx,
                                int lddy, int accumulate, float* ws, long long ws_floats, long long* plan, void* stream) {
    return stem_wgrad_impl(x, dy, dw, N, H, W, C, K, R, 1, ldx, lddy, accumulate, ws, ws_floats, plan, true, stream, cw);
}

// the stride-2 member of the family: dw (32,3,3,16) from x (N,H,W,16) and dy (N,H/2,W/2,32) (DLA-34 level1, dla.py:291-295).
// deterministic != 0: as omni_stem_conv_wgrad_det (ws / plan); 0: fp32 atomics into dw, ws / plan unused
int omni_stem_conv_s2_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int K, int R, int ldx, int lddy,
                            int accumulate, int deterministic, float* ws, long long ws_floats, long long* plan, void* stream) {
    return stem_wgrad_impl(x, dy, dw, N, H, W, C, K, R, 2, ldx, lddy, accumulate, ws, ws_floats, deterministic ? plan : nullptr,
                           deterministic != 0, stream);
}

}  // extern "C"
