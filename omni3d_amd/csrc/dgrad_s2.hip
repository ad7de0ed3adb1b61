// dgrad_s2.hip -- data gradient of the 3x3 / stride-2 / pad-1 convolutions (round 6).
//
//   dx[n, ih, iw, c] = sum over (r, s, k) with (ih + 1 - r), (iw + 1 - s) even of dy[n, (ih + 1 - r) / 2, (iw + 1 - s) / 2, k] * w[k, r, s, c]
//
// Reference call sites: autograd of the stride-2 nn.Conv2d that opens every DLA level (cubercnn/modeling/backbone/dla.py:43-51,
// 186-215: BasicBlock.conv1 of tree1) and every ResNet stage (torchvision BasicBlock / resnet.py:30-60).
//
// A strided gradient decomposes into stride^2 = 4 dense convolutions, one per parity class (a, b) = (ih & 1, iw & 1) of the dx pixel,
// with 1 / 2 / 2 / 4 of the nine filter taps.  The generic kernel (conv_gemm.hip conv_dgrad_kernel) runs the classes as gridDim.z
// slices of one launch: every class re-reads the whole of dy, far apart in time, and every slice stages its own share of the filter
// -- PMC (profiles/r05_pmc_families.csv): 79.9 MB moved for 25.5 MB of operands on the 64 -> 128 layer (3.1x), 0.22-0.31 of the
// fp32-MFMA peak inside the step.
//
// Here ONE workgroup owns an 8 x 8 tile of dy pixels (+ a one-pixel halo on the high side) and produces the 16 x 16 dx pixels of
// all four classes from it:
//   * per reduction slab of 32 dy channels the dy tile (81 pixels) and the nine tap slices of the filter for the workgroup's dx
//     channels are staged in LDS ONCE; the four classes read the same dy fragments (a tap only shifts the pixel by (dj, di) in
//     {0, 1}^2: four distinct A fragments serve all nine taps);
//   * wave (wm, wn) owns 32 pixels x 32 channels of EVERY class (four 32 x 32 accumulators, nine taps of work each: balanced);
//   * the MFMA reduction pairs are (k, k + 16) so that a lane reads 16 consecutive floats of its pixel / channel row: four
//     ds_read_b128 per fragment, 52 per 144 MFMAs;
//   * the next slab's global loads are issued before the current slab's MFMAs (register-staged prefetch, one LDS buffer);
//   * every dx element is written exactly once by its owner (plain store, or read-modify-write for a gradient fan-in target):
//     no atomics, no zero-fill, run-to-run identical.
// dy is read once (plus halo), the filter once per workgroup from L2.
#include <device_rt.h>

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

constexpr int TH = 8, TW = 8, HP = TH + 1, WP = TW + 1;      // dy tile and its halo
constexpr int KC = 32, KCP = KC + 4;                           // reduction slab and the padded LDS row (144 B: conflict-free b128)

// NT: dx channels per workgroup (64: waves 2 x 2 over pixels x channels; 32: waves 2 x 2 over pixels x class groups)
template <int NT>
__global__ void __launch_bounds__(256) dgrad_s2_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                                       int N, int H, int W, int C, int K, int OH, int OW, int lddy, int lddx,
                                                       int accumulate) {
    __shared__ __attribute__((aligned(16))) float s_a[HP * WP * KCP];
    __shared__ __attribute__((aligned(16))) float s_b[9 * NT * KCP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (OW + TW - 1) / TW, tiles_y = (OH + TH - 1) / TH;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int n = t / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int n0 = blockIdx.y * NT;                              // first dx channel of this workgroup

    // ---- global -> register staging of one slab ----
    constexpr int A4 = HP * WP * (KC / 4);                       // float4 loads of the dy tile: 648
    constexpr int NA = (A4 + 255) / 256;                         // 3
    constexpr int B4 = 9 * KC * (NT / 4);                        // float4 loads of the filter slices: 4608 (NT 64) / 2304 (NT 32)
    constexpr int NB = B4 / 256;                                 // 18 / 9
    static_assert(B4 % 256 == 0, "filter staging");
    float4 ra[NA], rb[NB];
    auto load_slab = [&](int k0) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int i = tid + 256 * u;
            const int k4 = i % (KC / 4), pix = i / (KC / 4);
            const int row = pix / WP, col = pix - row * WP;
            const int oh = oy0 + row, ow = ox0 + col;
            ra[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < A4 && oh < OH && ow < OW) ra[u] = ld4(dy + (((long)n * OH + oh) * OW + ow) * lddy + k0 + 4 * k4);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int i = tid + 256 * u;
            const int c4 = i % (NT / 4), kk = (i / (NT / 4)) % KC, tap = i / ((NT / 4) * KC);
            rb[u] = ld4(w + ((long)(k0 + kk) * 9 + tap) * C + n0 + 4 * c4);          // w (K, 3, 3, C)
        }
    };
    auto store_slab = [&]() {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int i = tid + 256 * u;
            const int k4 = i % (KC / 4), pix = i / (KC / 4);
            if (i < A4) *reinterpret_cast<float4*>(s_a + pix * KCP + 4 * k4) = ra[u];
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int i = tid + 256 * u;
            const int c4 = i % (NT / 4), kk = (i / (NT / 4)) % KC, tap = i / ((NT / 4) * KC);
            float* q = s_b + (tap * NT + 4 * c4) * KCP + kk;     // transposed: [tap][channel][k]
            q[0] = rb[u].x; q[KCP] = rb[u].y; q[2 * KCP] = rb[u].z; q[3 * KCP] = rb[u].w;
        }
    };

    // ---- this wave's share ----
    // NT == 64: wave = (wm, wn): pixels [32 wm, 32 wm + 32) x channels [32 wn, 32 wn + 32) of all four classes
    // NT == 32: wave = (wm, g):  pixels [32 wm, 32 wm + 32) x all 32 channels of the classes {(0,0), (1,1)} (g = 0) or {(0,1), (1,0)} (g = 1)
    const int wm = wave >> 1, wx = wave & 1;
    const int l31 = lane & 31, kh = lane >> 5;
    const int pj = 4 * wm + (l31 >> 3), pi = l31 & 7;             // this lane's dy-class pixel (j, i) inside the tile
    const int cb = (NT == 64 ? 32 * wx : 0) + l31;                // this lane's channel inside the workgroup's NT
    constexpr int NACC = NT == 64 ? 4 : 2;
    f32x16 acc[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    auto frag = [&](const float* base, float (&f)[16]) {           // 16 consecutive floats of one LDS row -> registers
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float4 x = *reinterpret_cast<const float4*>(base + 4 * v);
            f[4 * v] = x.x; f[4 * v + 1] = x.y; f[4 * v + 2] = x.z; f[4 * v + 3] = x.w;
        }
    };
    // one tap: class accumulator `q` += A(dj, di) x B(tap); A fragments are passed in (four of them serve the nine taps)
    auto tap_mma = [&](f32x16& c, const float (&a)[16], int tap) {
        float b[16];
        frag(s_b + (tap * NT + cb) * KCP + 16 * kh, b);
#pragma unroll
        for (int q = 0; q < 16; ++q) c = mfma_32x32x2(a[q], b[q], c);
    };

    const int nslab = K / KC;
    load_slab(0);
    for (int sl = 0; sl < nslab; ++sl) {
        store_slab();
        __syncthreads();
        if (sl + 1 < nslab) load_slab((sl + 1) * KC);            // in flight during the MFMAs below
        // A fragments: dy pixel (pj + dj, pi + di), k half kh
        float a00[16], a01[16], a10[16], a11[16];
        frag(s_a + ((pj) * WP + pi) * KCP + 16 * kh, a00);
        frag(s_a + ((pj) * WP + pi + 1) * KCP + 16 * kh, a01);
        frag(s_a + ((pj + 1) * WP + pi) * KCP + 16 * kh, a10);
        frag(s_a + ((pj + 1) * WP + pi + 1) * KCP + 16 * kh, a11);
        // taps: tap = 3 r + s.  row parity a = 0: r = 1 (dj 0); a = 1: r = 0 (dj 1), r = 2 (dj 0); columns likewise
        if (NT == 64 || wx == 0) {
            constexpr int Q11 = NT == 64 ? 3 : 1;
            tap_mma(acc[0], a00, 4);                              // class (0, 0): tap (1, 1)
            tap_mma(acc[Q11], a11, 0);                            // class (1, 1): (0, 0) -> (dj 1, di 1)
            tap_mma(acc[Q11], a10, 2);                            //               (0, 2) -> (1, 0)
            tap_mma(acc[Q11], a01, 6);                            //               (2, 0) -> (0, 1)
            tap_mma(acc[Q11], a00, 8);                            //               (2, 2) -> (0, 0)
        }
        if (NT == 64 || wx == 1) {
            constexpr int Q01 = NT == 64 ? 1 : 0, Q10 = NT == 64 ? 2 : 1;
            tap_mma(acc[Q01], a01, 3);                            // class (0, 1): (1, 0) -> (0, 1)
            tap_mma(acc[Q01], a00, 5);                            //               (1, 2) -> (0, 0)
            tap_mma(acc[Q10], a10, 1);                            // class (1, 0): (0, 1) -> (1, 0)
            tap_mma(acc[Q10], a00, 7);                            //               (2, 1) -> (0, 0)
        }
        __syncthreads();
    }

    // ---- epilogue: every element once.  acc register r of lane: pixel row (r & 3) + 8 (r >> 2) + 4 kh of the wave's 32, channel l31
    const int c_out = n0 + cb;
#pragma unroll
    for (int q = 0; q < NACC; ++q) {
        int ca, cbit;                                             // class (ca, cbit) of accumulator q
        if (NT == 64) { ca = q >> 1; cbit = q & 1; }
        else if (wx == 0) { ca = q; cbit = q; }                   // (0,0), (1,1)
        else { ca = q; cbit = 1 - q; }                            // (0,1), (1,0)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * kh;        // pixel within the wave's 32
            const int j = 4 * wm + (m >> 3), i = m & 7;
            const int ih = 2 * (oy0 + j) + ca, iw = 2 * (ox0 + i) + cbit;
            if (ih < H && iw < W && c_out < C) {
                float* o = dx + (((long)n * H + ih) * W + iw) * lddx + c_out;
                *o = accumulate ? *o + acc[q][r] : acc[q][r];
            }
        }
    }
}

}  // namespace

extern "C" {

// dx (N, H, W, C) [pixel pitch lddx] (=, or += when accumulate) the data gradient of conv2d(x, w (K, 3, 3, C), stride 2, padding 1)
// from dy (N, OH, OW, K) [pixel pitch lddy], OH = (H - 1) / 2 + 1.  C % 32 == 0, K % 32 == 0.  Every element of dx is written by
// exactly one workgroup: deterministic, no zero-fill.
int omni_conv2d_s2_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int K, int lddy, int lddx,
                         int accumulate, void* stream) {
    if (dy == nullptr || w == nullptr || dx == nullptr || N < 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || (C & 31) || (K & 31) ||
        lddy < K || lddx < C || (lddy & 3))
        return OMNI_ERR_ARG;
    if (N == 0) return OMNI_OK;
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const long tiles = (long)N * ((OH + TH - 1) / TH) * ((OW + TW - 1) / TW);
    if (tiles > 0x7fffffff) return OMNI_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if ((C & 63) == 0)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(dgrad_s2_kernel<64>), dim3((unsigned)tiles, (unsigned)(C / 64)), dim3(256), 0, st, dy, w, dx, N, H, W, C, K,
                           OH, OW, lddy, lddx, accumulate);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(dgrad_s2_kernel<32>), dim3((unsigned)tiles, (unsigned)(C / 32)), dim3(256), 0, st, dy, w, dx, N, H, W, C, K,
                           OH, OW, lddy, lddx, accumulate);
    return omni_launch_status();
}

}  // extern "C"
