// split_reduce.h -- run-to-run identical reduction of split-K partial tiles (round 4).
//
// The tile kernels cut a deep reduction over several workgroups when the tile count alone cannot fill 256 CUs (DLA levels 4 / 5,
// every weight gradient, the fc heads).  Rounds 1-3 met the partial tiles in the output with fp32 atomics: the order of the adds
// -- hence the rounding of the sum -- followed the order in which the workgroups happened to finish, so two runs of the same
// training step gave gradients that differ in the last bit, and a random-init network amplifies that to ~1 % of a bottom-up
// gradient tensor (profiles/r03_grad_run_to_run_spread.txt).  The reference's CPU path is run-to-run identical.
//
// Here every split writes its accumulators to a workspace slot, arrives at a per-tile counter, and the workgroup that arrives LAST
// adds the slots in split order 0, 1, 2, ... and runs the kernel's normal epilogue on the complete sum.  Which workgroup is last
// varies from run to run; what it computes does not.  Side effects: no zero-fill of the output in front of the launch, bias /
// ReLU / accumulate-into-carry are applied once on the full sum in the same kernel (no follow-up pass), no atomics on the output.
//
// Memory ordering on a chip with eight non-coherent L2s: the slots are written and read with agent-scope accesses (sc1: they go
// through to the memory side), a wave waits for its stores to be acknowledged before the workgroup barrier that precedes the
// arrival atomic, and the slot reads of the last workgroup are control-dependent on the value that atomic returned.  No
// whole-L2 write-back / invalidate (what a release / acquire fence pair costs on gfx950, measured in round 3: tools/bench_bn.py).
//
// Slot layout: NACC * 256 floats, element r * 256 + tid = accumulator register r of thread tid (coalesced across the 256 threads).
#pragma once
#include <device_rt.h>

#ifdef OMNI_HIPEMU
#define OMNI_LD_AGENT(p) (*(p))
#define OMNI_ST_AGENT(p, v) (*(p) = (v))
#else
#define OMNI_LD_AGENT(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define OMNI_ST_AGENT(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif

// All 256 threads of the workgroup call this with their accumulators.  Returns true in exactly one workgroup per tile -- the one
// that completes the sum -- with `acc` replaced by the sum of all `splits` partial tiles; false everywhere else.
//
// Two levels when there are more than 8 splits (weight gradients of the few-channel full-resolution layers run 256-512 pixel splits of
// two tiles): splits are grouped by omni_split_gsize(splits) in index order; the last arrival of a group adds the group's slots in split order,
// stores the group sum to a second-level slot and arrives at the tile's counter; the last group to arrive adds the group sums in
// group order.  The serial tail of one workgroup is G + splits / G tile reads instead of `splits`, and the order of every addition
// is still a function of the indices alone.
//   slots:    ws[(tile * (splits + groups) + s) * NACC * 256 ...]   s < splits: split partials, s >= splits: group sums
//   counters: ctr[tile * (1 + groups)] = groups arrived, ctr[tile * (1 + groups) + 1 + g] = splits of group g arrived
// with groups = 0 (and one counter per tile) when splits <= 8.  All counters zero on entry and on exit.
// group size: a power of two near sqrt(splits) (8 < splits <= 16: 4, <= 64: 8, <= 256: 16, else 32) -- a function of `splits` alone
static inline __host__ __device__ int omni_split_gsize(int splits) { return splits <= 8 ? 0 : splits <= 16 ? 4 : splits <= 64 ? 8 : splits <= 256 ? 16 : 32; }
static inline __host__ __device__ int omni_split_groups(int splits) {
    const int g = omni_split_gsize(splits);
    return g == 0 ? 0 : (splits + g - 1) / g;
}

template <int WM, int WN>
__device__ __forceinline__ void omni_split_store(float* __restrict__ slot, const f32x16 (&acc)[WM][WN]) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) OMNI_ST_AGENT(slot + ((i * WN + j) * 16 + r) * 256, acc[i][j][r]);
}
// acc = slot[0] + slot[1] + ... + slot[n - 1], added in that order; the loads of up to four slots (64 registers) are in flight together (the adds
// wait for them in order), because this is one workgroup's serial tail and every load is a trip to the memory side
template <int WM, int WN>
__device__ __forceinline__ void omni_split_sum(const float* __restrict__ slot, int n, f32x16 (&acc)[WM][WN]) {
    constexpr long NACC = WM * WN * 16;
    constexpr int B = NACC <= 16 ? 4 : 1;      // registers of loads in flight: chosen so that no kernel loses a resident workgroup to the tail
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int s = 0;
    for (; s + B <= n; s += B, slot += B * NACC * 256) {
        float t[B][WM * WN * 16];
#pragma unroll
        for (int b = 0; b < B; ++b)
#pragma unroll
            for (int e = 0; e < WM * WN * 16; ++e) t[b][e] = OMNI_LD_AGENT(slot + ((long)b * NACC + e) * 256);
#pragma unroll
        for (int b = 0; b < B; ++b)
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += t[b][(i * WN + j) * 16 + r];
    }
    for (; s < n; ++s, slot += NACC * 256) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += OMNI_LD_AGENT(slot + ((i * WN + j) * 16 + r) * 256);
    }
}
// publish this workgroup's slot stores and take a ticket; true for the arrival number `expect - 1` (which also resets the counter)
__device__ __forceinline__ bool omni_split_arrive(unsigned* __restrict__ counter, int expect) {
    __shared__ unsigned s_prev;
    OMNI_WAIT_VMCNT(0);                 // this wave's slot stores are acknowledged ...
    __syncthreads();                    // ... and so are the other three waves' before the arrival is published
    if (threadIdx.x == 0) {
        s_prev = atomicAdd(counter, 1u);
        if (s_prev == (unsigned)(expect - 1)) OMNI_ST_AGENT(counter, 0u);     // the next launch finds the counter at zero again
    }
    __syncthreads();
    return s_prev == (unsigned)(expect - 1);
}

template <int WM, int WN>
__device__ __forceinline__ bool omni_split_reduce(float* __restrict__ ws, unsigned* __restrict__ ctr, long tile, int split, int splits,
                                                  f32x16 (&acc)[WM][WN]) {
    constexpr long NACC = WM * WN * 16;
    const int tid = threadIdx.x;
    const int groups = omni_split_groups(splits);
    float* base = ws + tile * (long)(splits + groups) * (NACC * 256) + tid;
    omni_split_store<WM, WN>(base + (long)split * (NACC * 256), acc);
    if (groups == 0) {
        if (!omni_split_arrive(ctr + tile, splits)) return false;
        omni_split_sum<WM, WN>(base, splits, acc);
        return true;
    }
    unsigned* c = ctr + tile * (long)(1 + groups);
    const int gs = omni_split_gsize(splits);
    const int g = split / gs, g0 = g * gs;
    const int gn = min(gs, splits - g0);
    if (!omni_split_arrive(c + 1 + g, gn)) return false;
    omni_split_sum<WM, WN>(base + (long)g0 * (NACC * 256), gn, acc);
    omni_split_store<WM, WN>(base + (long)(splits + g) * (NACC * 256), acc);
    if (!omni_split_arrive(c, groups)) return false;
    omni_split_sum<WM, WN>(base + (long)splits * (NACC * 256), groups, acc);
    return true;
}

// host side: floats of workspace / counters a launch with `tiles` output tiles of NACC accumulators per thread and `splits` needs
static inline long omni_split_ws_floats(long tiles, long splits, long tile_elems) { return tiles * (splits + omni_split_groups((int)splits)) * tile_elems; }
static inline long omni_split_counters(long tiles, long splits) { return tiles * (1 + omni_split_groups((int)splits)); }
