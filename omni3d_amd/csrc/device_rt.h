// device_rt.h -- gfx950 device/runtime prelude shared by every kernel translation unit.
//
// All kernels in omni3d_amd/csrc are written against plain HIP for CDNA4 (wave64, MFMA,
// LDS).  This header only pulls in the HIP runtime and defines the small vector typedefs
// and helpers the kernels share.  (tests/hipemu/ carries a host-side stand-in with the
// same name that lets the *same kernel sources* run on the CPU of a GPU-less CI box; it
// is test infrastructure and is never linked into the product library.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define OMNI_WAVE 64

// status codes returned by every extern "C" entry point
#define OMNI_OK 0
#define OMNI_ERR_ARG 1
#define OMNI_ERR_LAUNCH 2

static inline int omni_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? OMNI_OK : OMNI_ERR_LAUNCH;
}

// Byte fill as a KERNEL (not hipMemsetAsync): a kernel node is replayed faithfully when the training step is captured
// into a hipGraph (cubercnn/solver/graphed.py); memset nodes were observed not to re-zero split-K / atomic
// accumulators on replay on ROCm 7.2.
static __global__ void __launch_bounds__(256) omni_fill_kernel(unsigned char* __restrict__ p, unsigned int pat4, size_t head,
                                                               size_t n16, size_t n) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (size_t)gridDim.x * blockDim.x;
    uint4* body = reinterpret_cast<uint4*>(p + head);
    const uint4 v = make_uint4(pat4, pat4, pat4, pat4);
    for (size_t i = gid; i < n16; i += nthr) body[i] = v;
    const unsigned char b = (unsigned char)(pat4 & 0xFFu);
    for (size_t i = gid; i < head; i += nthr) p[i] = b;
    for (size_t i = head + n16 * 16 + gid; i < n; i += nthr) p[i] = b;
}
static inline void omni_memset_async(void* ptr, int byte, size_t n, hipStream_t st) {
    if (n == 0) return;
    size_t head = (size_t)((16 - ((uintptr_t)ptr & 15)) & 15);
    if (head > n) head = n;
    const size_t n16 = (n - head) / 16;
    const unsigned int b = (unsigned int)(byte & 0xFF), pat4 = b | (b << 8) | (b << 16) | (b << 24);
    size_t g = (n16 + 255) / 256;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(omni_fill_kernel, dim3((unsigned)g), dim3(256), 0, st, (unsigned char*)ptr, pat4, head, n16, n);
}

// wave-level reductions (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// value of `v` in lane `lane` (a compile-time constant) as a wave-uniform scalar: v_readlane_b32
__device__ __forceinline__ float omni_readlane(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// exact-f32 matrix cores (CDNA4): D = A(32x2) * B(2x32) + C, one f32 per lane for A and B.
// lane l holds A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31];
// C/D: col j = l & 31, row i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5) for register r in [0,16).
__device__ __forceinline__ f32x16 mfma_32x32x2(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// D = A(16x4) * B(4x16) + C.  lane l: A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15];
// C/D: col j = l & 15, row i = 4 * (l >> 4) + r, r in [0,4).
__device__ __forceinline__ f32x4 mfma_16x16x4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- LDS-DMA / scheduling primitives used by csrc/gemm_engine.hip --------------------------------------------------------
// buffer resource over [p, p + bytes): loads through it return 0 for offsets outside the range (no branches for tile tails)
typedef __amdgpu_buffer_rsrc_t omni_rsrc_t;
__device__ __forceinline__ omni_rsrc_t omni_make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
// `buffer_load_dwordx4 ... lds`: every lane moves 16 bytes from base + voffset + soffset (bytes) straight into LDS at
// lds_wave_base + lane * 16 (wave-uniform base, lane-linear destination: 1 KiB per wave instruction); no VGPR round trip
#define OMNI_LDSP(p) ((__attribute__((address_space(3))) void*)(p))
__device__ __forceinline__ void omni_dma16(omni_rsrc_t r, float* lds_wave_base, int voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, OMNI_LDSP(lds_wave_base), 16, voffset, soffset, 0, 0);
}
// one dword through a buffer resource: out-of-range offsets come back as 0 without a branch
__device__ __forceinline__ unsigned omni_bufld1(omni_rsrc_t r, int voffset) { return __builtin_amdgcn_raw_buffer_load_b32(r, voffset, 0, 0); }
#define OMNI_OOB ((int)0x80000000)            /* voffset that is out of range for every resource (tensors are < 2 GiB) */
#define OMNI_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
__device__ __forceinline__ void omni_barrier() {          // s_barrier WITHOUT the vmcnt(0) drain __syncthreads() implies
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// workgroup barrier for LDS hand-over with global loads still in flight: waits for this wave's LDS traffic only (lgkmcnt), not
// for vmcnt -- __syncthreads() drains both, which serialises every register-staged prefetch deeper than one slab
__device__ __forceinline__ void omni_barrier_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// register budget hint: ask the compiler to fit n waves per SIMD (512 / n VGPRs)
#define OMNI_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n)))
#define OMNI_OPAQUE_V(x) asm volatile("" : "+v"(x))                                  /* the optimiser forgets what it knows about a VGPR value */
#define OMNI_OPAQUE_S(x) asm volatile("" : "+s"(x))                                  /* same for a wave-uniform (SGPR) value */
#define OMNI_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)                            /* nothing is scheduled across this point */
#define OMNI_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)      /* 0x8 MFMA, 0x20 VMEM read, 0x100 DS read */
#define OMNI_SETPRIO(n) __builtin_amdgcn_s_setprio(n)

