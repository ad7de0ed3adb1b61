// tools/exp/gemm_exp.hip -- standalone experiment (not product code): fp32-MFMA NT GEMM main-loop variants on gfx950.
//   C[b] (M x N) = A[b] (M x K) * B[b] (N x K)^T
// build: hipcc --offload-arch=gfx950 -O3 -o gemm_exp gemm_exp.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct GP { const float* A; const float* B; float* C; int batch, M, N, K, tiles_m, tiles_n; };

__device__ __forceinline__ void tile_coords(int tiles_m, int tiles_n, int& tm, int& tn) {
    const int nwg = tiles_m * tiles_n;
    int id = blockIdx.x;
    const int q = nwg / 8, r = nwg % 8;
    const int xcd = id % 8, k = id / 8;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    id = (nwg >= 8) ? swz : id;
    if (tiles_n <= 4) { tn = id % tiles_n; tm = id / tiles_n; } else { tm = id % tiles_m; tn = id / tiles_m; }
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA engine: 128x128x32 tile, 4 waves (2x2), unpadded XOR-swizzled LDS rows (32 floats = 8 chunks of 16 B;
// chunk c of row r is stored at position c ^ (r & 7)), buffer_load ... lds with out-of-range rows -> zeros.
// ---------------------------------------------------------------------------------------------
template <int STAGES, bool FRAG_DB, bool PRIO>
__global__ void __launch_bounds__(256) gemm_dma(GP p) {
    constexpr int BM = 128, BN = 128, BK = 32, STAGE = (BM + BN) * BK;
    // one __shared__ object PER STAGE: hipcc's waitcnt insertion then proves that a ds_read of stage s cannot alias the
    // LDS-DMA in flight into another stage (with one array it drains vmcnt(0) before every fragment read)
    __shared__ __attribute__((aligned(1024))) float st0[STAGE];
    __shared__ __attribute__((aligned(1024))) float st1[STAGE];
    __shared__ __attribute__((aligned(1024))) float st2[STAGES == 3 ? STAGE : 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    int tile_m, tile_n;
    tile_coords(p.tiles_m, p.tiles_n, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN, b = blockIdx.z;
    const float* Ab = p.A + (long)b * p.M * p.K;
    const float* Bb = p.B + (long)b * p.N * p.K;
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, p.M * p.K * 4, 0x00020000);
    __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, p.N * p.K * 4, 0x00020000);
    // DMA source offsets (bytes) of this lane's 4 A pieces and 4 B pieces of slab 0
    int voa[4], vob[4];
    const int dr = lane >> 3, dc = (lane & 7) ^ dr;       // row within the 8-row piece, global chunk fetched
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + dr;
        voa[i] = (m0 + row < p.M) ? ((m0 + row) * p.K + dc * 4) * 4 : (int)0x80000000;
        vob[i] = (n0 + row < p.N) ? ((n0 + row) * p.K + dc * 4) * 4 : (int)0x80000000;
    }
    auto issue = [&](int kt, float* st) {
        float* sa = st + wave * 4 * 256;
        float* sb = sa + BM * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDSP(sa + i * 256), 16, voa[i], kt * BK * 4, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDSP(sb + i * 256), 16, vob[i], kt * BK * 4, 0, 0);
    };
    // fragment read offsets (floats) within a stage
    int fa[2], fb[2], sw[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        fa[i] = (wm * 64 + 32 * i + l31) * BK;
        fb[i] = BM * BK + (wn * 64 + 32 * i + l31) * BK;
    }
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) sw[kc] = (((2 * kc + h) ^ (l31 & 7)) << 2);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nk = p.K / BK;
    auto compute = [&](const float* S) {
        float4 a[2][2], bf[2][2];
        auto ldfrag = [&](int buf, int kc) {
#pragma unroll
            for (int i = 0; i < 2; ++i) a[buf][i] = *reinterpret_cast<const float4*>(S + fa[i] + sw[kc]);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[buf][j] = *reinterpret_cast<const float4*>(S + fb[j] + sw[kc]);
        };
        if (FRAG_DB) ldfrag(0, 0);
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            const int cur = FRAG_DB ? (kc & 1) : 0;
            if (FRAG_DB) { if (kc + 1 < 4) ldfrag(cur ^ 1, kc + 1); } else ldfrag(0, kc);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float av = t == 0 ? a[cur][i].x : t == 1 ? a[cur][i].y : t == 2 ? a[cur][i].z : a[cur][i].w;
                        const float bv = t == 0 ? bf[cur][j].x : t == 1 ? bf[cur][j].y : t == 2 ? bf[cur][j].z : bf[cur][j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
    };
    issue(0, st0);
    if (STAGES == 2) {
        // slab kt is read from `rd` while the DMA of slab kt + 1 lands in `wr`; ONE barrier per slab
        auto step = [&](int kt, const float* rd, float* wr) {
            __syncthreads();           // vmcnt(0) + barrier: slab kt landed everywhere, everyone is done reading `wr`
            if (kt + 1 < nk) issue(kt + 1, wr);
            compute(rd);
        };
        for (int kt = 0; kt < nk; kt += 2) {
            step(kt, st0, st1);
            if (kt + 1 < nk) step(kt + 1, st1, st0);
        }
    } else {
        if (nk > 1) issue(1, st1);
        auto step = [&](int kt, const float* rd, float* wr) {
            if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + 2 < nk) issue(kt + 2, wr);
            compute(rd);
        };
        for (int kt = 0; kt < nk; kt += 3) {
            step(kt, st0, st2);
            if (kt + 1 < nk) step(kt + 1, st1, st0);
            if (kt + 2 < nk) step(kt + 2, st2, st1);
        }
    }
    float* o = p.C + (long)b * p.M * p.N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + (wn * 2 + j) * 32 + l31;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int m = m0 + (wm * 2 + i) * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
                if (m < p.M && n < p.N) o[(long)m * p.N + n] = acc[i][j][rr];
            }
        }
}

#include "dma3x.inc"
// ---------------------------------------------------------------------------------------------
// baseline: register-staged double buffer with padded rows (the round-1 engine, GEMM-only form)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gemm_regstage(GP p) {
    constexpr int BM = 128, BN = 128, BKX = 32, BKP = BKX + 4, KQ = BKX / 4, RPP = 256 / KQ, AI = BM / RPP, BI = BN / RPP;
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * BKP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    const int kq = tid % KQ, lrow = tid / KQ;
    int tile_m, tile_n;
    tile_coords(p.tiles_m, p.tiles_n, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN, b = blockIdx.z;
    const float* a_ptr[AI]; const float* b_ptr[BI]; bool a_ok[AI], b_ok[BI];
#pragma unroll
    for (int i = 0; i < AI; ++i) { const int m = m0 + lrow + RPP * i; a_ok[i] = m < p.M; a_ptr[i] = p.A + ((long)b * p.M + (a_ok[i] ? m : 0)) * p.K + kq * 4; }
#pragma unroll
    for (int j = 0; j < BI; ++j) { const int n = n0 + lrow + RPP * j; b_ok[j] = n < p.N; b_ptr[j] = p.B + ((long)b * p.N + (b_ok[j] ? n : 0)) * p.K + kq * 4; }
    float4 ra[AI], rb[BI];
    auto load_slab = [&]() {
#pragma unroll
        for (int i = 0; i < AI; ++i) { ra[i] = a_ok[i] ? *reinterpret_cast<const float4*>(a_ptr[i]) : make_float4(0, 0, 0, 0); a_ptr[i] += BKX; }
#pragma unroll
        for (int j = 0; j < BI; ++j) { rb[j] = b_ok[j] ? *reinterpret_cast<const float4*>(b_ptr[j]) : make_float4(0, 0, 0, 0); b_ptr[j] += BKX; }
    };
    auto store_slab = [&](int buf) {
        float* As = smem + buf * (BM + BN) * BKP; float* Bs = As + BM * BKP;
#pragma unroll
        for (int i = 0; i < AI; ++i) *reinterpret_cast<float4*>(As + (lrow + RPP * i) * BKP + kq * 4) = ra[i];
#pragma unroll
        for (int j = 0; j < BI; ++j) *reinterpret_cast<float4*>(Bs + (lrow + RPP * j) * BKP + kq * 4) = rb[j];
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nk = p.K / BKX;
    load_slab(); store_slab(0); __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_slab();
        const float* As = smem + buf * (BM + BN) * BKP; const float* Bs = As + BM * BKP;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            float4 a[2], bb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const float4*>(As + (wm * 64 + 32 * i + l31) * BKP + 8 * kc + 4 * h);
#pragma unroll
            for (int j = 0; j < 2; ++j) bb[j] = *reinterpret_cast<const float4*>(Bs + (wn * 64 + 32 * j + l31) * BKP + 8 * kc + 4 * h);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float av = t == 0 ? a[i].x : t == 1 ? a[i].y : t == 2 ? a[i].z : a[i].w;
                        const float bv = t == 0 ? bb[j].x : t == 1 ? bb[j].y : t == 2 ? bb[j].z : bb[j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
        }
        if (kt + 1 < nk) store_slab(buf ^ 1);
        __syncthreads();
    }
    float* o = p.C + (long)b * p.M * p.N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + (wn * 2 + j) * 32 + l31;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int m = m0 + (wm * 2 + i) * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
                if (m < p.M && n < p.N) o[(long)m * p.N + n] = acc[i][j][rr];
            }
        }
}

typedef void (*kern_t)(GP);
struct Var { const char* name; kern_t k; int bm = 128, bn = 128; };

__global__ void __launch_bounds__(256) mfma_only(float* o, int iters, float seed) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = seed * (threadIdx.x % 17) + 0.37f, b = seed * (threadIdx.x % 13) - 0.21f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
        }
        a = -a; 
    }
    float s = 0; for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    o[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const bool zero = argc > 1 && argv[1][0] == 'z';
    {
        float* o; CK(hipMalloc(&o, 2048 * 256 * 4));
        for (int wg : {256, 512, 1024}) {
            const int iters = 4096;
            hipLaunchKernelGGL(mfma_only, dim3(wg), dim3(256), 0, 0, o, iters, 0.001f);
            CK(hipDeviceSynchronize());
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(mfma_only, dim3(wg), dim3(256), 0, 0, o, iters, 0.001f);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
            const double gf = (double)wg * 4 * iters * 16 * 2.0 * 32 * 32 * 2 / 1e9;
            printf("mfma_only %d WGs: %.3f ms %.1f TFLOP/s\n", wg, ms, gf / ms);
        }
    }
    struct Shape { const char* name; int batch, M, N, K; } shapes[] = {
        {"wino p2 36x[4096x256x256]", 36, 4096, 256, 256},
        {"direct-like [65536x256x2304]", 1, 65536, 256, 2304},
        {"fc1 splitK4-like 4x[2048x1024x3136]", 4, 2048, 1024, 3136},
        {"wino l3 36x[1024x128x128]", 36, 1024, 128, 128},
        {"wino l4 36x[256x256x256]", 36, 256, 256, 256},
        {"ragged 3x[1000x200x96]", 3, 1000, 200, 96},
    };
    Var vars[] = {
        {"regstage(base)", gemm_regstage},
        {"dma2", gemm_dma<2, false, false>},
        {"dma2+fragdb", gemm_dma<2, true, false>},
        {"dma2+fragdb+prio", gemm_dma<2, true, true>},
        {"dma3+fragdb", gemm_dma<3, true, false>},
        {"dma3+fragdb+prio", gemm_dma<3, true, true>},
        {"dma3x 128x128", gemm_dma3x<128, 128, false, false>},
        {"dma3x+sgb 128x128", gemm_dma3x<128, 128, true, false>},
        {"dma3x 256x128", gemm_dma3x<256, 128, false, false>, 256, 128},
        {"dma3x+sgb 256x128", gemm_dma3x<256, 128, true, false>, 256, 128},
        {"dma3x+sgb+prio 256x128", gemm_dma3x<256, 128, true, true>, 256, 128},
    };
    for (auto& s : shapes) {
        const size_t na = (size_t)s.batch * s.M * s.K, nb = (size_t)s.batch * s.N * s.K, nc = (size_t)s.batch * s.M * s.N;
        std::vector<float> ha(na), hb(nb), hc(nc);
        unsigned seed = 12345;
        auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
        for (auto& v : ha) v = zero ? 0.f : rnd();
        for (auto& v : hb) v = rnd();
        float *da, *db, *dc;
        CK(hipMalloc(&da, na * 4 + 64)); CK(hipMalloc(&db, nb * 4 + 64)); CK(hipMalloc(&dc, nc * 4));
        CK(hipMemcpy(da, ha.data(), na * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), nb * 4, hipMemcpyHostToDevice));
        GP p{da, db, dc, s.batch, s.M, s.N, s.K, (s.M + 127) / 128, (s.N + 127) / 128};
        const double gflop = 2.0 * s.batch * s.M * (double)s.N * s.K / 1e9;
        printf("== %s  (%.2f GFLOP, %d workgroups)\n", s.name, gflop, p.tiles_m * p.tiles_n * s.batch);
        for (auto& v : vars) {
            CK(hipMemset(dc, 0xff, nc * 4));
            dim3 grid(((s.M + v.bm - 1) / v.bm) * ((s.N + v.bn - 1) / v.bn), 1, s.batch);
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(v.k, grid, dim3(256), 0, 0, p);
            CK(hipDeviceSynchronize());
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            const int iters = 20;
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(v.k, grid, dim3(256), 0, 0, p);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
            CK(hipMemcpy(hc.data(), dc, nc * 4, hipMemcpyDeviceToHost));
            // check 2000 sampled entries against an fp64 dot product
            double maxerr = 0; unsigned cs = 777;
            for (int q = 0; q < 2000; ++q) {
                cs = cs * 1664525u + 1013904223u; const int bb = (cs >> 8) % s.batch;
                cs = cs * 1664525u + 1013904223u; const int m = (q < 64) ? (s.M - 1 - q % s.M) : (cs >> 8) % s.M;
                cs = cs * 1664525u + 1013904223u; const int n = (q < 64) ? (s.N - 1 - (q * 7) % s.N) : (cs >> 8) % s.N;
                double ref = 0;
                for (int k = 0; k < s.K; ++k) ref += (double)ha[((size_t)bb * s.M + m) * s.K + k] * hb[((size_t)bb * s.N + n) * s.K + k];
                const double e = fabs(ref - hc[((size_t)bb * s.M + m) * s.N + n]);
                if (!(e <= maxerr)) maxerr = e;
            }
            printf("  %-22s %8.4f ms  %7.1f TFLOP/s  (%.3f of 157.3)  max|err| %.2e %s\n", v.name, ms, gflop / ms, gflop / ms / 157.3, maxerr,
                   maxerr < 1e-3 ? "ok" : "MISMATCH");
        }
        CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dc));
    }
    return 0;
}
