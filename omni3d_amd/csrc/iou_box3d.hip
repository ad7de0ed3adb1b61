// iou_box3d.hip -- exact oriented-box IoU3D for gfx950 (CDNA4).
//
// Replaces pytorch3d._C.iou_box3d as called by the reference evaluator
//   /root/reference/cubercnn/evaluation/omni3d_evaluation.py:155  (box3d_overlap, :106-166)
// and the validity masks _check_coplanar (:65-86) / _check_nonzero (:89-104).
//
// MI355X mapping (NOT the upstream CUDA thread-per-pair + per-thread scratch design):
//   * a 32-lane half of a wavefront owns one (dt, gt) pair (two pairs per wave; 64 / 16 lanes per pair selectable);
//     workgroup = one wave, so the LDS region is private to the wave and __syncthreads() is a wave-local fence
//   * both clip directions (tris(box1) vs planes(box2) and tris(box2) vs planes(box1)) live in one
//     LDS triangle list, lanes = triangles; every plane pass is clip -> ballot prefix -> stable
//     compaction into the other LDS buffer, so triangle order equals the sequential algorithm
//   * the O(n1*n2) coplanar-duplicate removal is spread over the 64 lanes with per-triangle
//     normals/areas cached in LDS
//   * I/O is 192 B in + 4 B out per pair: the kernel is VALU/latency bound, HBM
//     traffic is negligible.
// fp contraction is disabled so the epsilon-threshold branches take exactly the decisions of the
// CPU oracle (oracle/iou_box3d_oracle.c), which the parity tests compare against.
#include <device_rt.h>
#pragma clang fp contract(off)

#if defined(OMNI_HIPEMU) && defined(IOU_DEBUG_HIST)
// (host-emulator instrumentation of tools/iou3d_list_sizes.py: sizes of the joint triangle list entering each plane pass / the dedupe
// phase, and the rounds of the clipping code all waves execute)
int g_iou_hist[7][128], g_iou_rounds[2];
long g_iou_phase[4][2];      // [rounds | with the normal | with the coplanarity test | with the intersection code][executions, active lanes]
extern "C" long* omni_debug_iou_phase() { return &g_iou_phase[0][0]; }
extern "C" int* omni_debug_iou_hist() { return &g_iou_hist[0][0]; }
extern "C" int* omni_debug_iou_rounds() { return g_iou_rounds; }
#endif

namespace {

constexpr float K_EPS = 1e-8f;
constexpr float D_EPS = 1e-3f;
constexpr float A_EPS = 1e-4f;
constexpr int CAP = 160;  // triangles per LDS list (both directions together)

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 vsub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 vadd(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 vscale(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 vdiv(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ float vdot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 vcross(V3 a, V3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float vnorm(V3 a) { return sqrtf(vdot(a, a)); }
__device__ __forceinline__ V3 get_normal(V3 e0, V3 e1) {
    V3 n = vcross(e0, e1);
    return vdiv(n, fmaxf(vnorm(n), K_EPS));
}

struct Tri { V3 v[3]; };

__device__ __forceinline__ V3 ldv(const float* p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ void stv(float* p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
__device__ __forceinline__ Tri ldtri(const float* p) {
    Tri t; t.v[0] = ldv(p); t.v[1] = ldv(p + 3); t.v[2] = ldv(p + 6); return t;
}
__device__ __forceinline__ void sttri(float* p, const Tri& t) { stv(p, t.v[0]); stv(p + 3, t.v[1]); stv(p + 6, t.v[2]); }

__device__ __forceinline__ V3 tri_normal(const Tri& t) {
    V3 ctr = vdiv(vadd(vadd(t.v[0], t.v[1]), t.v[2]), 3.0f);
    V3 a0 = vsub(t.v[0], ctr), a1 = vsub(t.v[1], ctr), a2 = vsub(t.v[2], ctr);
    // first maximum of the three cross-product norms wins (the comparison chain of the sequential form, NaNs included); the winner's
    // normal is then computed ONCE -- the sequential form normalised inside every taken branch, and in a wave all three branches are
    // taken by some lane (round 5: two get_normal bodies less per call, same operands into the one that remains)
    float best = -1.0f;
    int sel = -1;
    const float d01 = vnorm(vcross(a0, a1));
    if (d01 > best) { best = d01; sel = 0; }
    const float d02 = vnorm(vcross(a0, a2));
    if (d02 > best) { best = d02; sel = 1; }
    const float d12 = vnorm(vcross(a1, a2));
    if (d12 > best) { best = d12; sel = 2; }
    if (sel < 0) return mk(0.f, 0.f, 0.f);
    const V3 p = sel == 2 ? a1 : a0, q = sel == 0 ? a1 : a2;
    return get_normal(p, q);
}
__device__ __forceinline__ float tri_area(const Tri& t) {
    return vnorm(vcross(vsub(t.v[1], t.v[0]), vsub(t.v[2], t.v[0]))) / 2.0f;
}

__constant__ int c_box_tris[12][3] = {
    {0, 1, 2}, {0, 3, 2}, {4, 5, 6}, {4, 6, 7}, {1, 5, 6}, {1, 6, 2},
    {0, 4, 7}, {0, 7, 3}, {3, 2, 6}, {3, 6, 7}, {0, 1, 5}, {0, 4, 5}};
__constant__ int c_box_planes[6][4] = {
    {0, 1, 2, 3}, {3, 2, 6, 7}, {0, 1, 5, 4}, {0, 3, 7, 4}, {1, 2, 6, 5}, {4, 5, 6, 7}};

// farthest (triangle vertex, other vertex) pair, first maximum wins (iou_utils.h ArgMaxVerts)
template <int NO>
__device__ __forceinline__ V3 argmax_dir(const Tri& t, const V3* other) {
    float best = -1.0f;
    V3 a = mk(0.f, 0.f, 0.f), b = mk(0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < NO; ++j) {
            float d = vnorm(vsub(t.v[i], other[j]));
            if (d > best) { best = d; a = t.v[i]; b = other[j]; }
        }
    V3 d = vsub(a, b);
    return vdiv(d, fmaxf(vnorm(d), K_EPS));
}

__device__ __forceinline__ V3 plane_edge_intersection(V3 pc, V3 normal, V3 p0, V3 p1) {
    const V3 e = vsub(p1, p0);
    // The edge is cut unless it runs within asin(1e-3) of the plane: |dot(e / max(|e|, 1e-8), normal)| >= 1e-3.  Round 5: that test
    // needs a square root and three divisions and is true for almost every edge -- an edge with (e . n)^2 > (1.015e-3)^2 |e|^2 and
    // |e|^2 >= 1e-15 passes it whatever the rounding of the normalisation (1.5 % margin against ~1e-6), so the normalised form is only
    // evaluated for the rest, and a wave skips it when none of its lanes has such an edge.  Same decision, same intersection point.
    const float bot = vdot(e, normal);
    const float len2 = vdot(e, e);
    bool cut = bot * bot > 1.030225e-6f * len2 && len2 >= 1e-15f;
    if (!cut) {
        const V3 direc = vdiv(e, fmaxf(vnorm(e), K_EPS));
        cut = fabsf(vdot(direc, normal)) >= D_EPS;
    }
    V3 p = vdiv(vadd(p1, p0), 2.0f);
    if (cut) {
        float top = -1.0f * vdot(vsub(p0, pc), normal);
        float a = top / bot;
        p = vadd(p0, vscale(e, a));
    }
    return p;
}

// clip one triangle by one face plane; returns 0..2 triangles in o0/o1
#if defined(OMNI_HIPEMU) && defined(IOU_DEBUG_HIST)
static int g_clip_dbg[64];      // per lane, bits of its last clip_tri call: 1 normal needed | 2 coplanarity test | 4 clipped
#define IOU_DBG(b) (g_clip_dbg[threadIdx.x & 63] |= (b))
#else
#define IOU_DBG(b) ((void)0)
#endif
__device__ __forceinline__ int clip_tri(const V3* pv, V3 pc, V3 normal, const Tri& t, Tri& o0, Tri& o1) {
    V3 v0 = t.v[0], v1 = t.v[1], v2 = t.v[2];
    const float d0 = vdot(vsub(v0, pc), normal), d1 = vdot(vsub(v1, pc), normal), d2 = vdot(vsub(v2, pc), normal);
    bool in0 = d0 >= 0.0f;
    bool in1 = d1 >= 0.0f;
    bool in2 = d2 >= 0.0f;
    // coplanar triangle is kept as is.  The test starts with |nt . normal| > 1 - 1e-3, i.e. the triangle within 2.56 degrees of the plane:
    // then the plane distances of its vertices differ by at most sin(2.56 deg) = 0.0447 of its longest edge.  Round 5: a triangle whose
    // distances spread over MORE than 0.05 of the longest edge cannot pass that test, and its normal (three cross products, four
    // square roots, three divisions: a third of this function) is not computed at all -- most triangles in four of the six passes of
    // a yaw-rotated box pair; a wave skips the code when none of its lanes needs it.  Conservative (12 % margin, NaNs take the full
    // test), so every decision is the one the full test makes.
    const float spread = fmaxf(d0, fmaxf(d1, d2)) - fminf(d0, fminf(d1, d2));
    const V3 e01 = vsub(v1, v0), e02 = vsub(v2, v0), e12 = vsub(v2, v1);
    const float l2 = fmaxf(vdot(e01, e01), fmaxf(vdot(e02, e02), vdot(e12, e12)));
    const bool maybe_parallel = !(spread * spread > 0.0025f * l2);
    bool coplanar = false;
    if (maybe_parallel) {
        IOU_DBG(1);
        V3 nt = tri_normal(t);
        bool check1 = fabsf(vdot(nt, normal)) > 1.0f - D_EPS;
        if (check1) {
            IOU_DBG(2);
            V3 d = argmax_dir<4>(t, pv);
            coplanar = (fabsf(vdot(d, normal)) < D_EPS) || (fabsf(vdot(nt, d)) < D_EPS);
        }
    }
    if (coplanar || (in0 && in1 && in2)) { o0 = t; return 1; }
    if (!in0 && !in1 && !in2) return 0;
    const int nin = (int)in0 + (int)in1 + (int)in2;
    // Two vertices inside: the edges (vi1, vout) and (vi2, vout) are cut; one inside: (vin, vo1) and (vin, vo2).  Both cases are TWO
    // calls of plane_edge_intersection: the operands are selected first and the calls are shared (round 5) -- the lanes of a wave that
    // take different cases no longer execute the ~130-instruction intersection code twice with complementary halves masked off.
    // Same operands into the same function: the same floats.
    V3 a1, b1, a2, b2;
    if (nin == 2) {
        V3 vout, vi1, vi2;
        if (!in2) { vout = v2; vi1 = v0; vi2 = v1; }
        else if (!in1) { vout = v1; vi1 = v0; vi2 = v2; }
        else { vout = v0; vi1 = v1; vi2 = v2; }
        a1 = vi1; b1 = vout; a2 = vi2; b2 = vout;
    } else {
        V3 vin, vo1, vo2;
        if (in0) { vin = v0; vo1 = v1; vo2 = v2; }
        else if (in2) { vin = v2; vo1 = v0; vo2 = v1; }
        else { vin = v1; vo1 = v0; vo2 = v2; }
        a1 = vin; b1 = vo1; a2 = vin; b2 = vo2;
    }
    IOU_DBG(4);
    const V3 p1 = plane_edge_intersection(pc, normal, a1, b1);
    const V3 p2 = plane_edge_intersection(pc, normal, a2, b2);
    if (nin == 2) {
        o0.v[0] = a1; o0.v[1] = p1; o0.v[2] = a2;
        o1.v[0] = a2; o1.v[1] = p1; o1.v[2] = p2;
        return 2;
    }
    o0.v[0] = a1; o0.v[1] = p1; o0.v[2] = p2;
    return 1;
}

// LDS of ONE pair.  After the six plane passes the spare ping-pong buffer holds the dedupe phase's per-triangle unit normals
// [CAPT*3], areas [CAPT] and box2 keep flags [CAPT] (5 of its 9 floats per triangle).
template <int CAPT>
struct PairLds {
    float tri[2][CAPT * 9];  // ping-pong triangle lists
    float box[2][24];        // the two boxes' corners
    float pc[2][6][3];       // face-plane centres
    float pn[2][6][3];       // face-plane normals, pointing inside
    float vol[2];            // box volumes
};

constexpr float IOU_RETRY = -1.0f;   // first-pass marker: "this pair's triangle list outgrew the small LDS lists, recompute it"

// One pair per SUB-lane sub-group (G = 64 / SUB pairs per wave side by side).  A pair's joint triangle list starts with 24
// entries and rarely exceeds 40, so with one pair per wave at most ~40 of the 64 lanes ever had a triangle (PMC round 2: VALU
// active 44 % of the wave cycles); sub-groups of 32 lanes run two pairs through the same instruction stream.  Every ballot is
// taken over the wave and cut to the sub-group's bit range, loops that contain wave-level operations run to the maximum trip
// count over the sub-groups, and each sub-group keeps its own LDS lists, so the per-pair algorithm -- including the order of
// the triangles, which the epsilon rules depend on -- is unchanged.
// act: this sub-group has a pair (b1 / b2 = its boxes); -> vol, iou (sub-group uniform), over = a list hit CAPT.
template <int SUB, int CAPT>
__device__ __forceinline__ void iou_pair_body(PairLds<CAPT>& L, const bool act, const float* __restrict__ b1, const float* __restrict__ b2,
                                              const int lane, float& vol_r, float& iou_r, bool& over_r) {
    const int g = lane / SUB, sl = lane % SUB;
    const int shift = g * SUB;
    const unsigned long long sub_all = (SUB == 64) ? ~0ull : ((1ull << (SUB & 63)) - 1ull);
    const unsigned long long sub_lt = (sl == 0) ? 0ull : (~0ull >> (64 - sl));      // sub-group lanes below this one
    auto group_max = [&](int v) {                    // maximum over the wave's sub-groups (v is uniform inside a sub-group)
#pragma unroll
        for (int m = SUB; m < 64; m <<= 1) v = max(v, __shfl_xor(v, m, 64));
        return v;
    };
    auto group_sum = [&](float v) {                  // sum over the lanes of this sub-group
#pragma unroll
        for (int m = SUB / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        return v;
    };
    __syncthreads();  // previous pairs' LDS reads are done
    if (act)
        for (int k = sl; k < 48; k += SUB) {
            if (k < 24) L.box[0][k] = b1[k];
            else L.box[1][k - 24] = b2[k - 24];
        }
    __syncthreads();

    // ---- per-box prologue: face planes (sub-lanes 0..11), volumes (12, 13), initial triangles (all)
    if (act && sl < 12) {
        const int bx = sl / 6, f = sl % 6;
        const float* B = L.box[bx];
        V3 ctr = mk(0.f, 0.f, 0.f);
        for (int t = 0; t < 8; ++t) { ctr.x += B[3 * t]; ctr.y += B[3 * t + 1]; ctr.z += B[3 * t + 2]; }
        ctr = vdiv(ctr, 8.0f);
        V3 q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = ldv(B + 3 * c_box_planes[f][k]);
        V3 pc = vdiv(vadd(vadd(vadd(q[0], q[1]), q[2]), q[3]), 4.0f);
        float best = -1.0f;
        V3 ba = mk(0.f, 0.f, 0.f), bb = mk(0.f, 0.f, 0.f);
        bool any = false;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = i + 1; j < 4; ++j) {
                V3 a = vsub(q[i], pc), b = vsub(q[j], pc);
                float d = vnorm(vcross(a, b));
                if (d > best) { best = d; ba = a; bb = b; any = true; }      // (the winner's normal is computed once, below)
            }
        V3 n = any ? get_normal(ba, bb) : mk(0.f, 0.f, 0.f);
        if (vdot(vsub(ctr, pc), n) < 0.0f) n = vscale(n, -1.0f);
        stv(L.pc[bx][f], pc);
        stv(L.pn[bx][f], n);
    }
    if (SUB >= 32) {
        // box volumes (round 5): one tetrahedron per sub-lane (24 of them) instead of two lanes walking twelve each -- 360 VALU
        // instructions with 2 of 32 lanes active were 3 % lane utilisation (profiles/r05_pmc_iou3d.csv: 0.32 over the whole kernel).
        // The twelve terms of a box are then added in the ORACLE's order (t = 0, 1, ..., 11) by one lane, so the volume is the same float.
        float term = 0.f;
        if (act && sl < 24) {
            const int bx = sl / 12, t = sl % 12;
            const float* B = L.box[bx];
            V3 ctr = mk(0.f, 0.f, 0.f);
            for (int q = 0; q < 8; ++q) { ctr.x += B[3 * q]; ctr.y += B[3 * q + 1]; ctr.z += B[3 * q + 2]; }
            ctr = vdiv(ctr, 8.0f);
            V3 a = vsub(ldv(B + 3 * c_box_tris[t][0]), ctr);
            V3 b = vsub(ldv(B + 3 * c_box_tris[t][1]), ctr);
            V3 c = vsub(ldv(B + 3 * c_box_tris[t][2]), ctr);
            term = fabsf(vdot(a, vcross(b, c))) / 6.0f;
        }
        float v0 = 0.f, v1 = 0.f;
#pragma unroll
        for (int t = 0; t < 12; ++t) {
            v0 += __shfl(term, shift + t, 64);
            v1 += __shfl(term, shift + 12 + t, 64);
        }
        if (act && sl == 0) { L.vol[0] = v0; L.vol[1] = v1; }
    } else if (act && sl >= 12 && sl < 14) {
        const int bx = sl - 12;
        const float* B = L.box[bx];
        V3 ctr = mk(0.f, 0.f, 0.f);
        for (int t = 0; t < 8; ++t) { ctr.x += B[3 * t]; ctr.y += B[3 * t + 1]; ctr.z += B[3 * t + 2]; }
        ctr = vdiv(ctr, 8.0f);
        float vol = 0.f;
        for (int t = 0; t < 12; ++t) {
            V3 a = vsub(ldv(B + 3 * c_box_tris[t][0]), ctr);
            V3 b = vsub(ldv(B + 3 * c_box_tris[t][1]), ctr);
            V3 c = vsub(ldv(B + 3 * c_box_tris[t][2]), ctr);
            vol += fabsf(vdot(a, vcross(b, c))) / 6.0f;
        }
        L.vol[bx] = vol;
    }
    if (act)
        for (int t = sl; t < 24; t += SUB) {
            const int bx = t / 12, tt = t % 12;
            const float* B = L.box[bx];
            float* dst = L.tri[0] + t * 9;
#pragma unroll
            for (int k = 0; k < 3; ++k) stv(dst + 3 * k, ldv(B + 3 * c_box_tris[tt][k]));
        }
    __syncthreads();

    // ---- six plane passes over the joint list: entries [0,nA) are box1 triangles clipped by
    //      box2's planes, entries [nA,n) box2 triangles clipped by box1's planes
    int n = act ? 24 : 0, nA = act ? 12 : 0, cur = 0;
    bool over = false;
    for (int f = 0; f < 6; ++f) {
        const float* src = L.tri[cur];
        float* dst = L.tri[cur ^ 1];
        int base = 0, newA = 0;
        const int nmax = group_max(n);
#if defined(OMNI_HIPEMU) && defined(IOU_DEBUG_HIST)
        if (act && sl == 0) ++g_iou_hist[f][n < 127 ? n : 127];
#endif
        for (int i0 = 0; i0 < nmax; i0 += SUB) {
#if defined(OMNI_HIPEMU) && defined(IOU_DEBUG_HIST)
            if (lane == 0) ++g_iou_rounds[0];
#endif
            const int i = i0 + sl;
            int cnt = 0;
            Tri o0, o1;
#if defined(OMNI_HIPEMU) && defined(IOU_DEBUG_HIST)
            g_clip_dbg[lane] = 0;
#endif
            if (i < n) {
                const int other = (i < nA) ? 1 : 0;  // plane set of the other box
                V3 pv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) pv[k] = ldv(L.box[other] + 3 * c_box_planes[f][k]);
                Tri t = ldtri(src + i * 9);
                cnt = clip_tri(pv, ldv(L.pc[other][f]), ldv(L.pn[other][f]), t, o0, o1);
            }
#if defined(OMNI_HIPEMU) && defined(IOU_DEBUG_HIST)
            {   // how often does a ROUND (the whole wave) execute the normal / the coplanarity test / the intersection code, with how many lanes
                const unsigned long long q1 = __ballot(g_clip_dbg[lane] & 1), q2 = __ballot(g_clip_dbg[lane] & 2), q4 = __ballot(g_clip_dbg[lane] & 4), qa = __ballot(i < n);
                if (lane == 0) {
                    g_iou_phase[0][0] += 1; g_iou_phase[0][1] += __popcll(qa);
                    if (q1) { g_iou_phase[1][0] += 1; g_iou_phase[1][1] += __popcll(q1); }
                    if (q2) { g_iou_phase[2][0] += 1; g_iou_phase[2][1] += __popcll(q2); }
                    if (q4) { g_iou_phase[3][0] += 1; g_iou_phase[3][1] += __popcll(q4); }
                }
            }
#endif
            const unsigned long long b1m = (__ballot(cnt >= 1) >> shift) & sub_all, b2m = (__ballot(cnt == 2) >> shift) & sub_all;
            const int off = base + __popcll(b1m & sub_lt) + __popcll(b2m & sub_lt);
            if (cnt >= 1) { if (off < CAPT) sttri(dst + off * 9, o0); else over = true; }
            if (cnt == 2) { if (off + 1 < CAPT) sttri(dst + (off + 1) * 9, o1); else over = true; }
            // outputs produced by box1-side entries of this round
            int nAround = nA - i0; nAround = nAround < 0 ? 0 : (nAround > SUB ? SUB : nAround);
            const unsigned long long amask = (nAround >= 64) ? ~0ull : ((1ull << nAround) - 1ull);
            newA += __popcll(b1m & amask) + __popcll(b2m & amask);
            base += __popcll(b1m) + __popcll(b2m);
        }
        n = base < CAPT ? base : CAPT;
        nA = newA < n ? newA : n;
        cur ^= 1;
        __syncthreads();
    }
#if defined(OMNI_HIPEMU) && defined(IOU_DEBUG_HIST)
    if (act && sl == 0) ++g_iou_hist[6][n < 127 ? n : 127];
#endif
    const float* T = L.tri[cur];
    float* aux = L.tri[cur ^ 1];                 // spare list: normals | areas | keep flags of the dedupe phase
    float* nrm = aux;
    float* area = aux + 3 * CAPT;
    int* keep = reinterpret_cast<int*>(aux + 4 * CAPT);
    const int n1 = nA, n2 = n - nA;

    // ---- coplanar duplicate removal: box2 triangle q is dropped if coplanar with some box1
    //      triangle r whose area exceeds aEpsilon
    for (int i = sl; i < n; i += SUB) {
        Tri t = ldtri(T + i * 9);
        stv(nrm + 3 * i, tri_normal(t));
        area[i] = tri_area(t);
        keep[i] = 1;
    }
    __syncthreads();
    const int npair = n1 * n2;
    for (int w = sl; w < npair; w += SUB) {
        const int r = w / n2, q = n1 + (w % n2);
        if (area[r] > A_EPS) {
            V3 na = ldv(nrm + 3 * r), nb = ldv(nrm + 3 * q);
            if (fabsf(vdot(na, nb)) > 1.0f - D_EPS) {
                Tri ta = ldtri(T + r * 9);
                Tri tb = ldtri(T + q * 9);
                V3 d = argmax_dir<3>(ta, tb.v);
                if ((fabsf(vdot(d, na)) < D_EPS) || (fabsf(vdot(d, nb)) < D_EPS)) keep[q] = 0;
            }
        }
    }
    __syncthreads();

    // ---- polyhedron centre and tetrahedron-sum volume over the surviving triangles (box1's list + kept box2 entries),
    //      sub-group reductions.  The survivors are not compacted: sums do not care about the order.
    float cx = 0.f, cy = 0.f, cz = 0.f;
    int mine = 0;
    for (int i = sl; i < n; i += SUB) {
        if (i < n1 || keep[i] != 0) {
            Tri t = ldtri(T + i * 9);
            cx += (t.v[0].x + t.v[1].x + t.v[2].x) / 3.0f;
            cy += (t.v[0].y + t.v[1].y + t.v[2].y) / 3.0f;
            cz += (t.v[0].z + t.v[1].z + t.v[2].z) / 3.0f;
            ++mine;
        }
    }
    cx = group_sum(cx); cy = group_sum(cy); cz = group_sum(cz);
    const int m = (int)(group_sum((float)mine) + 0.5f);
    float v = 0.f;
    if (m > 0) {
        V3 ctr = vdiv(mk(cx, cy, cz), (float)m);
        for (int i = sl; i < n; i += SUB) {
            if (i < n1 || keep[i] != 0) {
                Tri t = ldtri(T + i * 9);
                V3 a = vsub(t.v[0], ctr), b = vsub(t.v[1], ctr), c = vsub(t.v[2], ctr);
                v += fabsf(vdot(a, vcross(b, c))) / 6.0f;
            }
        }
    }
    const float vol = group_sum(v);
    vol_r = (m > 0) ? vol : 0.f;
    iou_r = (m > 0 && act) ? vol / (L.vol[0] + L.vol[1] - vol) : 0.f;
    over_r = ((__ballot(over) >> shift) & sub_all) != 0ull;
}

// Bounding-sphere rejection.  Two PROPER boxes whose bounding spheres (centre = vertex mean, radius = farthest vertex) are
// disjoint cannot intersect, and the clipping algorithm then ends with empty triangle lists: a triangle of one box survives a
// plane pass of the other only if it is inside that face's half-space or lies IN the face's plane (the coplanarity rule keeps
// it "as is"), and to survive all six passes it would have to sit within the other box's extent along every face normal, i.e.
// inside its convex hull (the intersection of the six half-spaces of a parallelepiped IS its hull), which the sphere contains.
// The result is exactly vol = iou = 0, which is written without running the passes.
// "Proper" matters: for a degenerate operand (a zero-thickness box has zero face normals, so EVERYTHING counts as inside; a
// skewed vertex makes the half-space intersection larger than the hull) the reference algorithm returns garbage that does not
// vanish with distance, and parity means reproducing that garbage.  So the shortcut is only taken when both vertex sets are
// parallelepipeds in the documented corner order (omni3d_evaluation.py:117-142) -- the twelve edges equal e1 / e2 / e3 up to
// 1e-3 of the shortest edge, shortest edge > 1e-3, |det(e1, e2, e3)| >= 1e-2 |e1||e2||e3| -- and the measured deviation is
// added to the separation margin (1e-4 relative + 1e-4 absolute + 4 x deviation).  Everything else, NaN / Inf coordinates
// included, takes the full algorithm.  In an evaluation most (detection, ground truth) pairs of an image are far apart; in the
// bench workload ~50 %.
__device__ __forceinline__ bool box_sphere(const float* __restrict__ B, float (&c)[3], float& radius, float& dev) {
    float v[24];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float4 q = *reinterpret_cast<const float4*>(B + 4 * k);      // 96-byte rows: 16-byte aligned
        v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
    }
    float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) { sx += v[3 * t]; sy += v[3 * t + 1]; sz += v[3 * t + 2]; }
    c[0] = sx / 8.0f; c[1] = sy / 8.0f; c[2] = sz / 8.0f;
    float m = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const float dx = v[3 * t] - c[0], dy = v[3 * t + 1] - c[1], dz = v[3 * t + 2] - c[2];
        m = fmaxf(m, dx * dx + dy * dy + dz * dz);
    }
    radius = sqrtf(m);
    // corner order: 0-1-2-3 and 4-5-6-7 are opposite quads, i and i + 4 are joined.  e1 = v1 - v0, e2 = v3 - v0, e3 = v4 - v0
    float e[3][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { e[0][k] = v[3 + k] - v[k]; e[1][k] = v[9 + k] - v[k]; e[2][k] = v[12 + k] - v[k]; }
    constexpr int EDGES[9][3] = {{3, 2, 0}, {4, 5, 0}, {7, 6, 0}, {1, 2, 1}, {4, 7, 1}, {5, 6, 1}, {1, 5, 2}, {2, 6, 2}, {3, 7, 2}};
    float d = 0.f;
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int k = 0; k < 3; ++k)
            d = fmaxf(d, fabsf((v[3 * EDGES[q][1] + k] - v[3 * EDGES[q][0] + k]) - e[EDGES[q][2]][k]));
    dev = d;
    float len[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) len[a] = sqrtf(e[a][0] * e[a][0] + e[a][1] * e[a][1] + e[a][2] * e[a][2]);
    const float lmin = fminf(len[0], fminf(len[1], len[2]));
    const float det = e[0][0] * (e[1][1] * e[2][2] - e[1][2] * e[2][1]) - e[0][1] * (e[1][0] * e[2][2] - e[1][2] * e[2][0]) +
                      e[0][2] * (e[1][0] * e[2][1] - e[1][1] * e[2][0]);
    return lmin > 1e-3f && d <= 1e-3f * lmin && fabsf(det) >= 1e-2f * len[0] * len[1] * len[2];      // false for NaN
}

__device__ __forceinline__ bool spheres_disjoint(const float* __restrict__ b1, const float* __restrict__ b2) {
    float c1[3], c2[3], r1, r2, d1, d2;
    const bool ok1 = box_sphere(b1, c1, r1, d1), ok2 = box_sphere(b2, c2, r2, d2);
    const float dx = c1[0] - c2[0], dy = c1[1] - c2[1], dz = c1[2] - c2[2];
    const float d = sqrtf(dx * dx + dy * dy + dz * dz), rs = r1 + r2;
    return ok1 && ok2 && d > rs * 1.0001f + 1e-4f + 4.0f * (d1 + d2);
}

// MODE 0: matrix (pair p -> a = p / M, b = p % M); MODE 1: indexed pairs.
// First pass.  A wave takes 64 consecutive pairs: every lane screens one pair (validity mask, bounding spheres) and writes the
// zeros of the rejected ones; the survivors are then worked off G = 64 / SUB at a time, SUB lanes per pair, lists of CAPT
// triangles.  RETRY (CAPT below the full capacity): a pair whose list outgrows CAPT is marked with iou = IOU_RETRY for
// iou_box3d_retry_kernel instead of being counted as an overflow.
template <int MODE, int SUB, int CAPT, bool RETRY>
__global__ void __launch_bounds__(64) OMNI_WAVES_PER_EU(4) iou_box3d_kernel(
    const float* __restrict__ boxes1, const float* __restrict__ boxes2, const int* __restrict__ idx1, const int* __restrict__ idx2,
    const int* __restrict__ valid1, long long npairs, int M, float* __restrict__ vol_out, float* __restrict__ iou_out,
    int* __restrict__ overflow, int chunk) {
    constexpr int G = 64 / SUB;
    __shared__ PairLds<CAPT> Lall[G];
    const int lane = threadIdx.x, g = lane / SUB, sl = lane % SUB;
    // `chunk` (16 | 32 | 64) pairs are screened per round by the first `chunk` lanes: smaller chunks = more waves for the same
    // problem (the launcher keeps >= ~4096 of them when the problem allows), the screening itself is cheap
    for (long long c0 = (long long)blockIdx.x * chunk; c0 < npairs; c0 += (long long)gridDim.x * chunk) {
        const long long q = c0 + lane;
        int ia = 0, ib = 0;
        bool live = false;
        if (lane < chunk && q < npairs) {
            if (MODE == 0) { ia = (int)(q / M); ib = (int)(q % M); }
            else { ia = idx1[q]; ib = idx2[q]; }
            live = !(valid1 != nullptr && valid1[ia] == 0) && !spheres_disjoint(boxes1 + (size_t)ia * 24, boxes2 + (size_t)ib * 24);
            if (!live) { if (vol_out) vol_out[q] = 0.f; iou_out[q] = 0.f; }
        }
        unsigned long long todo = __ballot(live);
        while (todo != 0ull) {
            // sub-group g takes the (g+1)-th lowest survivor of this round
            int j = -1;
#pragma unroll
            for (int k = 0; k < G; ++k) {
                const int f = todo != 0ull ? __ffsll(todo) - 1 : -1;
                if (todo != 0ull) todo &= todo - 1ull;
                if (k == g) j = f;
            }
            const bool act = j >= 0;
            const int src = act ? j : 0;
            const int pa = __shfl(ia, src, 64), pb = __shfl(ib, src, 64);
            float vol, iou;
            bool over;
            iou_pair_body<SUB, CAPT>(Lall[g], act, boxes1 + (size_t)pa * 24, boxes2 + (size_t)pb * 24, lane, vol, iou, over);
            if (act && sl == 0) {
                const long long p = c0 + j;
                if (RETRY && over) { vol = 0.f; iou = IOU_RETRY; }
                if (vol_out) vol_out[p] = vol;
                iou_out[p] = iou;
                if (!RETRY && over && overflow) atomicAdd(overflow, 1);
            }
        }
    }
}

// Second pass of a RETRY launch: every wave scans 64 results at a time for the marker and recomputes the marked pairs, one per
// wave, with the full-capacity lists.  With no marked pair (the usual case: a joint list longer than 96 needs near-identical
// boxes) this is one read of the result vector.
template <int MODE>
__global__ void __launch_bounds__(64) iou_box3d_retry_kernel(const float* __restrict__ boxes1, const float* __restrict__ boxes2,
                                                             const int* __restrict__ idx1, const int* __restrict__ idx2, long long npairs,
                                                             int M, float* __restrict__ vol_out, float* __restrict__ iou_out,
                                                             int* __restrict__ overflow) {
    __shared__ PairLds<CAP> L;
    const int lane = threadIdx.x;
    for (long long c0 = (long long)blockIdx.x * 64; c0 < npairs; c0 += (long long)gridDim.x * 64) {
        const long long q = c0 + lane;
        unsigned long long todo = __ballot(q < npairs && iou_out[q] == IOU_RETRY);
        while (todo != 0ull) {
            const int j = __ffsll(todo) - 1;
            todo &= todo - 1ull;
            const long long p = c0 + j;
            int ia, ib;
            if (MODE == 0) { ia = (int)(p / M); ib = (int)(p % M); }
            else { ia = idx1[p]; ib = idx2[p]; }
            float vol, iou;
            bool over;
            iou_pair_body<64, CAP>(L, true, boxes1 + (size_t)ia * 24, boxes2 + (size_t)ib * 24, lane, vol, iou, over);
            if (lane == 0) {
                if (vol_out) vol_out[p] = vol;
                iou_out[p] = iou;
                if (over && overflow) atomicAdd(overflow, 1);
            }
        }
    }
}

// _check_coplanar & _check_nonzero (omni3d_evaluation.py:65-104): one lane per dt box.
// valid[i] = 1 iff both pass; counts[0] += #non-coplanar, counts[1] += #zero-area.
__global__ void box3d_validity_kernel(const float* __restrict__ boxes, int N, float eps_coplanar, float eps_nonzero,
                                      int* __restrict__ valid, int* __restrict__ counts) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float* B = boxes + (size_t)i * 24;
    float acc = 0.f;
    for (int p = 0; p < 6; ++p) {
        V3 q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = ldv(B + 3 * c_box_planes[p][k]);
        V3 e0 = vsub(q[1], q[0]), e1 = vsub(q[2], q[0]);
        e0 = vdiv(e0, fmaxf(vnorm(e0), 1e-12f));
        e1 = vdiv(e1, fmaxf(vnorm(e1), 1e-12f));
        V3 n = vcross(e0, e1);
        n = vdiv(n, fmaxf(vnorm(n), 1e-12f));
        acc += vdot(vsub(q[3], q[0]), n);
    }
    const bool coplanar = fabsf(acc) < eps_coplanar;
    bool nonzero = true;
    for (int t = 0; t < 12; ++t) {
        V3 a = ldv(B + 3 * c_box_tris[t][0]), b = ldv(B + 3 * c_box_tris[t][1]), c = ldv(B + 3 * c_box_tris[t][2]);
        float area = vnorm(vcross(vsub(b, a), vsub(c, a))) / 2.0f;
        if (!(area > eps_nonzero)) nonzero = false;
    }
    valid[i] = (coplanar && nonzero) ? 1 : 0;
    if (counts) {
        if (!coplanar) atomicAdd(&counts[0], 1);
        if (!nonzero) atomicAdd(&counts[1], 1);
    }
}

// production launch: 32 lanes per pair over 96-triangle lists (7.4 KB of LDS per pair: 10 two-pair waves per CU instead of 6
// with the full 160-triangle lists), marked pairs redone at full capacity by the retry pass
constexpr int IOU_VARIANT = 1032;      // (set from tools/bench_iou3d.py: profiles/r03_iou3d_variants*.log)

// (round 5 re-measured on the 100 k-pair workload: 4 / 8 / 16 / 32 / 64 pairs per wave -> 0.640 / 0.585 / 0.544 / 0.659 / 0.746 ms)
inline int iou_chunk(long long npairs) { return npairs >= 262144 ? 64 : npairs >= 131072 ? 32 : 16; }

inline int iou_grid(long long npairs) {
    // 256 CUs x up to 16 single-wave workgroups per CU (LDS-limited); grid-stride beyond that
    const int chunk = iou_chunk(npairs);
    const long long waves = (npairs + chunk - 1) / chunk;
    long long g = waves < 256 * 16 ? waves : 256 * 16;
    return (int)(g < 1 ? 1 : g);
}

// variant = lanes_per_pair (64 | 32 | 16) + 1000 when the first pass uses the small lists + retry pass; 0 = production
template <int MODE>
inline int iou_launch(int variant, const float* boxes1, const float* boxes2, const int* idx1, const int* idx2, const int* valid1,
                      long long np, int M, float* vol, float* iou, int* overflow, void* stream) {
    if (variant == 0) variant = IOU_VARIANT;
    const bool small = variant >= 1000;
    const int cap_code = variant / 1000;                 // 0: full lists | 1: 96 | 2: 64 | 3: 48 triangles in the first pass
    const int sub = variant % 1000;
    hipStream_t st = (hipStream_t)stream;
#define OMNI_IOU(SUB_, CAP_, RETRY_)                                                                                          \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(iou_box3d_kernel<MODE, SUB_, CAP_, RETRY_>), dim3(iou_grid(np)), dim3(64), 0, st,       \
                       boxes1, boxes2, idx1, idx2, valid1, np, M, vol, iou, overflow, iou_chunk(np))
    if (!small) {
        if (sub == 64) OMNI_IOU(64, CAP, false);
        else if (sub == 32) OMNI_IOU(32, CAP, false);
        else if (sub == 16) OMNI_IOU(16, CAP, false);
        else return OMNI_ERR_ARG;
    } else {
        if (cap_code == 1 && sub == 64) OMNI_IOU(64, 96, true);
        else if (cap_code == 1 && sub == 32) OMNI_IOU(32, 96, true);
        else if (cap_code == 1 && sub == 16) OMNI_IOU(16, 96, true);
        else if (cap_code == 2 && sub == 32) OMNI_IOU(32, 64, true);
        else if (cap_code == 2 && sub == 64) OMNI_IOU(64, 64, true);
        else if (cap_code == 3 && sub == 32) OMNI_IOU(32, 48, true);
        else return OMNI_ERR_ARG;
        const long long chunks = (np + 63) / 64;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(iou_box3d_retry_kernel<MODE>), dim3((unsigned)(chunks < 2048 ? chunks : 2048)), dim3(64), 0, st,
                           boxes1, boxes2, idx1, idx2, np, M, vol, iou, overflow);
    }
#undef OMNI_IOU
    return omni_launch_status();
}

}  // namespace

extern "C" {

int omni_iou_box3d(const float* boxes1, int N, const float* boxes2, int M, const int* valid1, float* vol, float* iou,
                   int* overflow, void* stream) {
    if (N < 0 || M < 0) return OMNI_ERR_ARG;
    const long long np = (long long)N * M;
    if (np == 0) return OMNI_OK;
    return iou_launch<0>(0, boxes1, boxes2, nullptr, nullptr, valid1, np, M, vol, iou, overflow, stream);
}

int omni_iou_box3d_pairs(const float* boxes1, const float* boxes2, const int* idx1, const int* idx2, long long npairs,
                         const int* valid1, float* vol, float* iou, int* overflow, void* stream) {
    if (npairs < 0) return OMNI_ERR_ARG;
    if (npairs == 0) return OMNI_OK;
    return iou_launch<1>(0, boxes1, boxes2, idx1, idx2, valid1, npairs, 1, vol, iou, overflow, stream);
}

// lanes_per_pair in {64, 32, 16}: one launch with the full-capacity triangle lists; 1000 / 2000 / 3000 + lanes: first pass over
// 96- / 64- / 48-triangle lists + retry pass (0 = the production choice).  A/B entry point of tools/bench_iou3d.py and of the parity tests, which
// run every variant against the oracle
int omni_iou_box3d_pairs_algo(const float* boxes1, const float* boxes2, const int* idx1, const int* idx2, long long npairs,
                              const int* valid1, float* vol, float* iou, int* overflow, int lanes_per_pair, void* stream) {
    if (npairs < 0) return OMNI_ERR_ARG;
    if (npairs == 0) return OMNI_OK;
    return iou_launch<1>(lanes_per_pair, boxes1, boxes2, idx1, idx2, valid1, npairs, 1, vol, iou, overflow, stream);
}

int omni_box3d_validity(const float* boxes, int N, float eps_coplanar, float eps_nonzero, int* valid, int* counts,
                        void* stream) {
    if (N < 0) return OMNI_ERR_ARG;
    if (N == 0) return OMNI_OK;
    hipLaunchKernelGGL(box3d_validity_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, boxes, N,
                       eps_coplanar, eps_nonzero, valid, counts);
    return omni_launch_status();
}

}  // extern "C"
