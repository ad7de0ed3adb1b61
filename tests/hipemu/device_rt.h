// tests/hipemu/device_rt.h -- HOST-SIDE STAND-IN for omni3d_amd/csrc/device_rt.h.
//
// TEST INFRASTRUCTURE ONLY.  The GPU-less CI container cannot execute gfx950 code, so the
// CPU test-suite compiles the *unmodified* kernel sources (omni3d_amd/csrc/*.hip) with the
// host clang++ against this header, which maps the HIP execution model onto fibers:
//   * one fiber per thread of a workgroup, workgroups executed one after another
//   * __syncthreads / wave shuffles / ballots / MFMA are rendezvous points
//   * wave = 64 lanes; MFMA fragment layouts follow the CDNA4 register maps
// Nothing here is shipped: the product library is built by hipcc from the real device_rt.h
// and omni3d_amd/lib.py refuses to run without it.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <math.h>
#include <cmath>
#include <algorithm>
#include <functional>

#define OMNI_HIPEMU 1
using std::min;
using std::max;
#define OMNI_WAVE 64
#define OMNI_OK 0
#define OMNI_ERR_ARG 1
#define OMNI_ERR_LAUNCH 2

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __constant__ static const
#define __launch_bounds__(...)
#define HIP_KERNEL_NAME(...) __VA_ARGS__
#define HIP_DYNAMIC_SHARED(type, var) type* var = (type*)hipemu::dyn_smem();

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
// element-wise operators HIP's vector types provide natively
static inline float4 operator+(float4 a, float4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
static inline float4 operator-(float4 a, float4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
static inline float4 operator*(float4 a, float4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
static inline float4 operator*(float4 a, float b) { return {a.x * b, a.y * b, a.z * b, a.w * b}; }
static inline float4 operator*(float a, float4 b) { return {a * b.x, a * b.y, a * b.z, a * b.w}; }
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float3 make_float3(float x, float y, float z) { return {x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return {x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline void omni_memset_async(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); }
enum hipMemcpyKind { hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return 0; }

namespace hipemu {
struct Ctx { dim3 tid, bid, bdim, gdim; int lane, wave, linear; };
extern Ctx g;                      // context of the fiber currently running
void* dyn_smem();
void block_sync();                 // __syncthreads
void wave_sync();                  // rendezvous of the live lanes of this wave
void* wave_scratch();              // 64 * 64 bytes per wave, valid between wave_sync()s
unsigned long long wave_live_mask();
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
}  // namespace hipemu

#define threadIdx (hipemu::g.tid)
#define blockIdx (hipemu::g.bid)
#define blockDim (hipemu::g.bdim)
#define gridDim (hipemu::g.gdim)
#define warpSize 64

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch(dim3(grid), dim3(block), (size_t)(shmem), [&]() { kernel(__VA_ARGS__); })

static inline int omni_launch_status() { return OMNI_OK; }

static inline void __syncthreads() { hipemu::block_sync(); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}

template <class T>
static inline T hipemu_exchange(T v, int src_lane) {
    static_assert(sizeof(T) <= 64, "exchange payload too large");
    char* s = (char*)hipemu::wave_scratch();
    int lane = hipemu::g.lane;
    memcpy(s + 64 * lane, &v, sizeof(T));
    hipemu::wave_sync();
    T r = v;
    if (src_lane >= 0 && src_lane < 64 && ((hipemu::wave_live_mask() >> src_lane) & 1ull))
        memcpy(&r, s + 64 * src_lane, sizeof(T));
    hipemu::wave_sync();
    return r;
}
template <class T> static inline T __shfl(T v, int src, int width = 64) {
    int lane = hipemu::g.lane;
    int base = lane & ~(width - 1);
    return hipemu_exchange(v, base + (src & (width - 1)));
}
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) {
    int lane = hipemu::g.lane;
    int src = lane ^ mask;
    if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
    return hipemu_exchange(v, src);
}
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
    int lane = hipemu::g.lane;
    int src = lane + (int)d;
    if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
    return hipemu_exchange(v, src);
}
template <class T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
    int lane = hipemu::g.lane;
    int src = lane - (int)d;
    if (src < 0 || (src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
    return hipemu_exchange(v, src);
}
// `live_out`: the live-lane mask the vote was taken over, sampled between the two rendezvous (a lane that has left the
// second one may run to the end of the kernel and retire before its neighbours compare their result)
static inline unsigned long long hipemu_ballot(int pred, unsigned long long* live_out) {
    unsigned long long* s = (unsigned long long*)hipemu::wave_scratch();
    int lane = hipemu::g.lane;
    s[lane * 8] = pred ? 1ull : 0ull;
    hipemu::wave_sync();
    unsigned long long m = 0, live = hipemu::wave_live_mask();
    for (int l = 0; l < 64; ++l)
        if (((live >> l) & 1ull) && s[l * 8]) m |= (1ull << l);
    hipemu::wave_sync();
    if (live_out) *live_out = live;
    return m;
}
static inline unsigned long long __ballot(int pred) { return hipemu_ballot(pred, nullptr); }
static inline int __any(int pred) { return __ballot(pred) != 0; }
static inline int __all(int pred) {
    unsigned long long live;
    return hipemu_ballot(pred, &live) == live;
}
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __clz(int v) { return v == 0 ? 32 : __builtin_clz((unsigned)v); }

// atomics: the emulator is single-threaded, plain read-modify-write is exact
template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> static inline T atomicMax(T* p, T v) { T o = *p; *p = o > v ? o : v; return o; }
template <class T> static inline T atomicMin(T* p, T v) { T o = *p; *p = o < v ? o : v; return o; }
template <class T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <class T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <class T> static inline T atomicCAS(T* p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }

static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
static inline float __int_as_float(int v) { float f; memcpy(&f, &v, 4); return f; }
static inline int __float_as_int(float f) { int v; memcpy(&v, &f, 4); return v; }
static inline unsigned __float_as_uint(float f) { unsigned v; memcpy(&v, &f, 4); return v; }
static inline float __uint_as_float(unsigned v) { float f; memcpy(&f, &v, 4); return f; }

static inline float wave_sum(float v) {
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
static inline float wave_max(float v) {
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}
static inline int wave_sum_i(int v) {
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

static inline float omni_readlane(float v, int lane) { return __shfl(v, lane, 64); }

// MFMA emulation.  Bitwise model of v_mfma_f32_32x32x2_f32 per the CDNA4 guide: a k-ordered
// fmaf chain, one rounding per product.
static inline f32x16 mfma_32x32x2(float a, float b, f32x16 c) {
    float* s = (float*)hipemu::wave_scratch();
    int lane = hipemu::g.lane;
    s[lane * 16] = a;
    s[lane * 16 + 1] = b;
    hipemu::wave_sync();
    int j = lane & 31, hi = lane >> 5;
    f32x16 d;
    for (int r = 0; r < 16; ++r) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int k = 0; k < 2; ++k) acc = fmaf(s[(i + 32 * k) * 16], s[(j + 32 * k) * 16 + 1], acc);
        d[r] = acc;
    }
    hipemu::wave_sync();
    return d;
}
static inline f32x4 mfma_16x16x4(float a, float b, f32x4 c) {
    float* s = (float*)hipemu::wave_scratch();
    int lane = hipemu::g.lane;
    s[lane * 16] = a;
    s[lane * 16 + 1] = b;
    hipemu::wave_sync();
    int j = lane & 15, q = lane >> 4;
    f32x4 d;
    for (int r = 0; r < 4; ++r) {
        int i = 4 * q + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(s[(i + 16 * k) * 16], s[(j + 16 * k) * 16 + 1], acc);
        d[r] = acc;
    }
    hipemu::wave_sync();
    return d;
}

// v_mfma_f32_32x32x16_bf16 (csrc/gemm_split.hip, the opt-in bf16-split experiment): lane l holds 8 bf16 of row / column l & 31 for
// k = 8 * (l >> 5) + [0, 8); products are exact in fp32, accumulated in fp32 in k order
static inline f32x16 mfma_bf16_32x32x16(const unsigned short* a, const unsigned short* b, f32x16 c) {
    unsigned short* s = (unsigned short*)hipemu::wave_scratch();
    int lane = hipemu::g.lane;
    for (int e = 0; e < 8; ++e) { s[lane * 32 + e] = a[e]; s[lane * 32 + 8 + e] = b[e]; }
    hipemu::wave_sync();
    int j = lane & 31, hi = lane >> 5;
    f32x16 d;
    for (int r = 0; r < 16; ++r) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int g = 0; g < 2; ++g)
            for (int e = 0; e < 8; ++e) {
                unsigned ua = (unsigned)s[(i + 32 * g) * 32 + e] << 16, ub = (unsigned)s[(j + 32 * g) * 32 + 8 + e] << 16;
                float fa, fb;
                memcpy(&fa, &ua, 4);
                memcpy(&fb, &ub, 4);
                acc = fmaf(fa, fb, acc);
            }
        d[r] = acc;
    }
    hipemu::wave_sync();
    return d;
}

// ---- LDS-DMA / scheduling primitives of csrc/gemm_engine.hip, emulated synchronously -----------------------------------
struct omni_rsrc_t { const char* base; unsigned bytes; };
static inline omni_rsrc_t omni_make_rsrc(const void* p, unsigned bytes) { return {(const char*)p, bytes}; }
#define OMNI_LDSP(p) (p)
static inline void omni_dma16(omni_rsrc_t r, float* lds_wave_base, int voffset, int soffset) {
    char* dst = (char*)lds_wave_base + 16 * hipemu::g.lane;
    const unsigned vo = (unsigned)voffset;                       // range check on the vector offset, like a raw buffer
    if ((unsigned long long)vo + 16ull <= (unsigned long long)r.bytes && (long long)vo + soffset + 16 <= (long long)r.bytes)
        memcpy(dst, r.base + vo + soffset, 16);
    else
        memset(dst, 0, 16);
}
static inline unsigned omni_bufld1(omni_rsrc_t r, int voffset) {
    unsigned v = 0u;
    if ((unsigned long long)(unsigned)voffset + 4ull <= (unsigned long long)r.bytes) memcpy(&v, r.base + (unsigned)voffset, 4);
    return v;
}
#define OMNI_OOB ((int)0x80000000)
#define OMNI_WAIT_VMCNT(n) do { } while (0)
static inline void omni_barrier() { hipemu::block_sync(); }
static inline void omni_barrier_lds() { hipemu::block_sync(); }
#define OMNI_OPAQUE_V(x) do { } while (0)
#define OMNI_OPAQUE_S(x) do { } while (0)
#define OMNI_SCHED_FENCE() do { } while (0)
#define OMNI_SCHED_GROUP(mask, n) do { } while (0)
#define OMNI_SETPRIO(n) do { } while (0)
#define OMNI_WAVES_PER_EU(n)
