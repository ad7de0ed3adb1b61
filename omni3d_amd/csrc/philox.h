// philox.h -- counter-based Exp(1) variates for the two subsampling steps of the training path (round 6).
//
// The reference subsamples anchors and proposals with torch's generator (detectron2 `subsample_labels` -> torch.randperm;
// cubercnn/modeling/proposal_generator/rpn.py:318,322 torch.multinomial); this path formulates both as a top-k of w / Exp(1) keys.
// Rounds 1-5 filled the variates with `Tensor.exponential_()`: under hipGraph capture every such call is an ATen kernel PLUS the
// graph-safe generator bookkeeping torch adds around it (two int64 fills before each replay, an offset increment per draw) -- six
// dependent launches on the critical stream for two arrays of random numbers.  Here the consuming kernel draws its variate itself:
// Philox4x32-10 (Salmon et al., SC'11; the generator behind torch's CUDA RNG) keyed by a 64-bit seed, counter = (element index, row,
// draw counter); nothing is stored, nothing is launched.  The draw counter lives in device memory (`state[1]`, state[0] = seed) and is
// advanced by the LAST workgroup of the consuming launch to arrive at a ticket -- every workgroup reads the counter before it
// takes its ticket, so no workgroup can see the advanced value, and a captured graph replays a fresh draw every time.
#pragma once
#include <device_rt.h>

#ifdef OMNI_HIPEMU
#define OMNI_RNG_LD(p) (*(p))
#define OMNI_RNG_ST(p, v) (*(p) = (v))
#else
#define OMNI_RNG_LD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define OMNI_RNG_ST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif

static __host__ __device__ inline void omni_mulhilo32(unsigned a, unsigned b, unsigned& hi, unsigned& lo) {
    const unsigned long long p = (unsigned long long)a * (unsigned long long)b;
    hi = (unsigned)(p >> 32);
    lo = (unsigned)p;
}

// first word of Philox4x32-10(counter = (c0, c1, c2, c3), key = (k0, k1))
static __host__ __device__ inline unsigned omni_philox(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned h0, l0, h1, l1;
        omni_mulhilo32(0xD2511F53u, c0, h0, l0);
        omni_mulhilo32(0xCD9E8D57u, c2, h1, l1);
        const unsigned n0 = h1 ^ c1 ^ k0, n1 = l1, n2 = h0 ^ c3 ^ k1, n3 = l0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c0;
}

// Exp(1) variate of (element, row) in draw `ctr` of the stream `seed`: -log(u), u uniform on (0, 1) from the top 24 bits (never 0 or 1)
static __host__ __device__ inline float omni_exp1(unsigned long long seed, unsigned long long ctr, unsigned row, unsigned elem) {
    const unsigned x = omni_philox(elem, row, (unsigned)ctr, (unsigned)(ctr >> 32), (unsigned)seed, (unsigned)(seed >> 32));
    const float u = ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f);
    return -logf(u);
}

#if defined(__HIPCC__) || defined(OMNI_HIPEMU)
// Every workgroup: `ctr = omni_draw_begin(state)` before its first variate (one thread reads, the caller broadcasts it), and
// `omni_draw_end(state, ticket, ctr, nblocks)` from ONE thread after the read has been broadcast.  state: (2) int64 [seed, draw counter];
// ticket: (1) int32, zero between launches.
static __device__ inline unsigned long long omni_draw_begin(const long long* state) {
    return (unsigned long long)OMNI_RNG_LD(state + 1);
}
static __device__ inline void omni_draw_end(long long* state, int* ticket, unsigned long long ctr, int nblocks) {
    __threadfence();
    if (atomicAdd(ticket, 1) == nblocks - 1) {
        OMNI_RNG_ST(state + 1, (long long)(ctr + 1ull));
        OMNI_RNG_ST(ticket, 0);
    }
}
#endif
