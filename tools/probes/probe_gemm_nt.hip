// Probe (measurement tool, not part of the library): where do the 28 % of idle MFMA cycles of gemm_nt_pf_kernel<4> on the p2 point-GEMM
// shape 36 x [4096 x 256] x [256 x 256]^T go?  The kernel body is restated with switches that REMOVE one ingredient at a time
// (results are then wrong on purpose; only time is read):
//   bit 0 (1): no global loads inside the loop      bit 1 (2): no LDS writes inside the loop     bit 2 (4): no barrier inside the loop
//   bit 3 (8): two accumulators (even / odd k-steps) instead of one dependent chain           bit 4 (16): s_setprio 1 around the MFMAs
//   bit 5 (32): no epilogue stores            bit 6 (64): epilogue through a wave-private LDS transpose (4 x dwordx4 per lane)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I omni3d_amd/csrc -I include tools/probes/probe_gemm_nt.hip -o /tmp/probe_gemm_nt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "device_rt.h"

typedef unsigned omni_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 bufld4(omni_rsrc_t r, int voff) {
    const omni_u4 u = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
    return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
}
__device__ __forceinline__ int xcd_chunked(int id, int total) {
    if (total < 8) return id;
    const int q = total / 8, r = total % 8, xcd = id % 8, k = id / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}
struct P { const float* A; const float* B; float* out; int M, N, K; long ab, bb, ob; };

template <int MODE>
__global__ void __launch_bounds__(256) probe_kernel(P p) {
    constexpr int PF = 4, BM = 64, BN = 64, BKX = 32, BKP = BKX + 4, KQ = BKX / 4, RPP = 256 / KQ, AI = BM / RPP, BI = BN / RPP;
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * BKP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if ((MODE & 128) && blockIdx.x < 2048) {          // first-round workgroups of a CU start a quarter of a tile time apart
        const int k = (MODE & 256) ? (((int)blockIdx.x >> 3) & 3) : (((int)blockIdx.x >> 8) & 3);     // ids 0..255 land on the 256 CUs first (8 XCDs x 32 CUs), then the next 256, ...
        for (int i = 0; i < k; ++i) __builtin_amdgcn_s_sleep(127);
    }
    const int wm = wave >> 1, wn = wave & 1;
    const int kq = tid % KQ, lrow = tid / KQ;
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN, per_problem = tiles_m * tiles_n;
    const int item = xcd_chunked((int)blockIdx.x, (int)gridDim.x);
    const int prob = item / per_problem, tix = item - prob * per_problem;
    const omni_rsrc_t ra_ = omni_make_rsrc(p.A + (long)prob * p.ab, (unsigned)p.M * (unsigned)p.K * 4u);
    const omni_rsrc_t rb_ = omni_make_rsrc(p.B + (long)prob * p.bb, (unsigned)p.N * (unsigned)p.K * 4u);
    float* out = p.out + (long)prob * p.ob;
    const int tile_m = tix / tiles_n, tile_n = tix - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    int a_off[AI], b_off[BI];
#pragma unroll
    for (int i = 0; i < AI; ++i) { const int m = m0 + lrow + RPP * i; a_off[i] = m < p.M ? (m * p.K + kq * 4) * 4 : OMNI_OOB; }
#pragma unroll
    for (int j = 0; j < BI; ++j) { const int n = n0 + lrow + RPP * j; b_off[j] = n < p.N ? (n * p.K + kq * 4) * 4 : OMNI_OOB; }
    const int nk = p.K / BKX, k_end = p.K * 4;
    int koff = 0;
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
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
    for (int s = 0; s < PF; ++s) load_slab(s);
    store_slab(0, 0);
    load_slab(0);
    omni_barrier_lds();
    const int l31 = lane & 31, h = lane >> 5;
    for (int kt0 = 0; kt0 < nk; kt0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int buf = (kt0 + u) & 1;
            if (!(MODE & 2)) store_slab(buf ^ 1, (u + 1) % PF);
            if (!(MODE & 1)) load_slab((u + 1) % PF);
            const float* As = smem + buf * (BM + BN) * BKP;
            const float* Bs = As + BM * BKP;
            if (MODE & 16) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kc = 0; kc < BKX / 8; ++kc) {
                const float4 a = *reinterpret_cast<const float4*>(As + (wm * 32 + l31) * BKP + 8 * kc + 4 * h);
                const float4 b = *reinterpret_cast<const float4*>(Bs + (wn * 32 + l31) * BKP + 8 * kc + 4 * h);
                if (MODE & 8) {
                    acc0 = mfma_32x32x2(a.x, b.x, acc0); acc1 = mfma_32x32x2(a.y, b.y, acc1);
                    acc0 = mfma_32x32x2(a.z, b.z, acc0); acc1 = mfma_32x32x2(a.w, b.w, acc1);
                } else {
                    acc0 = mfma_32x32x2(a.x, b.x, acc0); acc0 = mfma_32x32x2(a.y, b.y, acc0);
                    acc0 = mfma_32x32x2(a.z, b.z, acc0); acc0 = mfma_32x32x2(a.w, b.w, acc0);
                }
            }
            if (MODE & 16) __builtin_amdgcn_s_setprio(0);
            if (!(MODE & 4)) omni_barrier_lds();
        }
    }
    if (MODE & 32) { if (acc0[0] == 123.456f) out[0] = acc0[1] + acc1[3]; return; }
    if (MODE & 64) {
        // wave-private 32 x 32 transpose through the (dead) slab buffers: 4 x 16-byte stores per lane, 8 full 128-byte rows each
        float* T = smem + wave * 1024;
#pragma unroll
        for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l31] = acc0[r] + ((MODE & 8) ? acc1[r] : 0.f);
        const int c4 = (lane & 7) * 4, r0 = lane >> 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = r0 + 8 * i;
            const float4 v = *reinterpret_cast<const float4*>(T + row * 32 + c4);
            const int m = m0 + wm * 32 + row, nn = n0 + wn * 32 + c4;
            if (m < p.M && nn < p.N) {
                if (MODE & 512) { typedef float f4v __attribute__((ext_vector_type(4))); f4v vv = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(vv, reinterpret_cast<f4v*>(out + (long)m * p.N + nn)); }
                else *reinterpret_cast<float4*>(out + (long)m * p.N + nn) = v;
            }
        }
        return;
    }
    const int n = n0 + wn * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < p.M && n < p.N) {
            if (MODE & 512) __builtin_nontemporal_store(acc0[r] + ((MODE & 8) ? acc1[r] : 0.f), out + (long)m * p.N + n);
            else out[(long)m * p.N + n] = acc0[r] + ((MODE & 8) ? acc1[r] : 0.f);
        }
    }
}

template <int MODE> float run(P p, int batch, int reps) {
    const long wgs = ((p.M + 63) / 64) * ((p.N + 63) / 64) * (long)batch;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(probe_kernel<MODE>, dim3((unsigned)wgs), dim3(256), 0, 0, p);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(probe_kernel<MODE>, dim3((unsigned)wgs), dim3(256), 0, 0, p);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
    const int batch = 36, M = argc > 1 ? atoi(argv[1]) : 4096, N = 256, K = argc > 2 ? atoi(argv[2]) : 256;
    const int zero = argc > 3 ? atoi(argv[3]) : 0;
    float *A, *B, *C;
    const size_t na = (size_t)batch * M * K, nb = (size_t)batch * N * K, nc = (size_t)batch * M * N;
    hipMalloc(&A, na * 4); hipMalloc(&B, nb * 4); hipMalloc(&C, nc * 4);
    std::vector<float> h(na > nb ? na : nb);
    srand(1);
    for (auto& v : h) v = zero ? 0.f : (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(A, h.data(), na * 4, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), nb * 4, hipMemcpyHostToDevice);
    P p{A, B, C, M, N, K, (long)M * K, (long)N * K, (long)M * N};
    const double gf = 2.0 * batch * M * N * K / 1e9;
    const int reps = 50;
    printf("shape 36 x [%d x %d] x [%d x %d]^T  %.2f GFLOP  inputs %s\n", M, K, N, K, gf, zero ? "zeros" : "random");
#define RUN(m, what) { float us = run<m>(p, batch, reps); printf("mode %2d  %-62s %7.1f us  %6.1f TF  %.3f of 157.3\n", m, what, us, gf / us * 1e3, gf / us * 1e3 / 157.3); }
    for (int pass = 0; pass < 2; ++pass) {
        RUN(0, "the kernel as shipped");
        RUN(1, "no global loads in the loop");
        RUN(3, "no global loads, no LDS writes");
        RUN(7, "no global loads, no LDS writes, no barrier (ds_read + MFMA)");
        RUN(4, "no barrier");
        RUN(8, "two accumulators (even / odd k-steps)");
        RUN(16, "s_setprio 1 around the MFMAs");
        RUN(24, "two accumulators + setprio");
        RUN(32, "no epilogue stores");
        RUN(33, "no epilogue stores, no loop loads");
        RUN(39, "ds_read + MFMA only, no stores");
        RUN(128, "first-round workgroups of a CU staggered by 0/1/2/3 x 3.4 us");
        RUN(384, "staggered, delay keyed on (id >> 3) & 3");
        RUN(392, "staggered (id >> 3) + two accumulators");
        RUN(512, "non-temporal epilogue stores");
        RUN(576, "LDS-transposed epilogue, non-temporal dwordx4 stores");
        RUN(584, "LDS-transposed + non-temporal + two accumulators");
        RUN(64, "epilogue through a wave-private LDS transpose, 4 x dwordx4");
        RUN(72, "LDS-transposed epilogue + two accumulators");
    }
    return 0;
}
