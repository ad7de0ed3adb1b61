// roi_align.hip -- FPN level assignment + ROIAlign (aligned=True, adaptive sampling grid) forward and
// backward over NHWC feature maps, for gfx950.
//
// Replaces detectron2 ROIPooler / ROIAlign -> torchvision.ops.roi_align, constructed at
//   box pooler   detectron2 StandardROIHeads._init_box_head   (used /root/reference/cubercnn/modeling/roi_heads/roi_heads.py:267)
//   cube pooler  /root/reference/cubercnn/modeling/roi_heads/roi_heads.py:166-171, used :362
// Semantics: level = clamp(floor(4 + log2(sqrt(area)/224 + 1e-8)), 2, 6);
// start = x1*scale - 0.5; bin = roi/P; grid = ceil(roi/P) samples per bin and axis; bilinear taps
// with the (y < -1 || y > H) -> 0 rule, clamp at 0 and the top-edge snap; mean over max(gh*gw, 1).
//
// MI355X mapping: upstream uses one thread per OUTPUT ELEMENT of an NCHW tensor (strided taps,
// scalar atomics in backward).  Here one 64-lane wave owns one (roi, bin) and the lanes walk the
// channel dimension 16 B each, so every tap is a single fully coalesced 1 KiB read (C = 256) and the
// backward scatter is a coalesced run of fp32 atomics.  Output is (R, P, P, C): with KRSC weights the
// following fc1 is a PxP "valid" convolution whose reduction index is contiguous on both operands.
#include <device_rt.h>
#pragma clang fp contract(off)

namespace {

constexpr int MAXL = 8;
struct FeatLevels {
    float* f[MAXL];   // (B, H, W, C) NHWC
    int H[MAXL], W[MAXL];
    float scale[MAXL];
    int nlev;
};

__global__ void roi_levels_kernel(const float* __restrict__ rois, int R, int min_level, int max_level,
                                  float canonical_size, int canonical_level, int* __restrict__ levels) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float w = rois[4 * r + 2] - rois[4 * r + 0], h = rois[4 * r + 3] - rois[4 * r + 1];
    const float sz = sqrtf(w * h);
    float lv = floorf((float)canonical_level + log2f(sz / canonical_size + 1e-8f));
    lv = fminf(fmaxf(lv, (float)min_level), (float)max_level);
    levels[r] = (int)lv - min_level;
}

struct Tap { int y0, y1, x0, x1; float w00, w01, w10, w11; bool ok; };

__device__ __forceinline__ Tap make_tap(float y, float x, int H, int W) {
    Tap t;
    t.ok = !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W);
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else { y_high = y_low + 1; }
    if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else { x_high = x_low + 1; }
    const float ly = y - (float)y_low, lx = x - (float)x_low, hy = 1.f - ly, hx = 1.f - lx;
    t.y0 = y_low; t.y1 = y_high; t.x0 = x_low; t.x1 = x_high;
    t.w00 = hy * hx; t.w01 = hy * lx; t.w10 = ly * hx; t.w11 = ly * lx;
    return t;
}

// DIR 0: forward (out (R,P,P,C));  DIR 1: backward (dout -> atomics into dfeat levels)
template <int DIR>
__global__ void __launch_bounds__(256) roi_align_kernel(FeatLevels fl, const float* __restrict__ rois,
                                                        const int* __restrict__ batch_idx, const int* __restrict__ levels,
                                                        int R, int P, int C, float* __restrict__ out,
                                                        float* __restrict__ out2 = nullptr, int per_image = 0, int first = 0, int aligned = 1) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long job = (long)blockIdx.x * 4 + wave;   // (roi, bin)
    if (job >= (long)R * P * P) return;
    const int r = (int)(job / (P * P)), bin = (int)(job % (P * P));
    const int ph = bin / P, pw = bin % P;
    const int l = levels[r];
    const int H = fl.H[l], W = fl.W[l];
    const float sc = fl.scale[l];
    // aligned (detectron2 "ROIAlignV2"): pixel-centre shift of half a pixel; otherwise ("ROIAlign", round 6) torchvision's legacy
    // form: no shift and a ROI of at least one pixel per side
    const float off = aligned ? 0.5f : 0.f;
    const float sw = rois[4 * r + 0] * sc - off, sh = rois[4 * r + 1] * sc - off;
    const float ew = rois[4 * r + 2] * sc - off, eh = rois[4 * r + 3] * sc - off;
    float rw = ew - sw, rh = eh - sh;
    if (!aligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
    const float bin_h = rh / (float)P, bin_w = rw / (float)P;
    const int gh = (int)ceilf(rh / (float)P), gw = (int)ceilf(rw / (float)P);
    const float count = (float)max(gh * gw, 1);
    float* feat = fl.f[l] + (long)batch_idx[r] * H * W * C;
    float* o = out + job * C;
    if (DIR == 1) {
        // backward: lane l owns channels l, l+64, l+128, ... so that every atomic instruction of the wave
        // covers 256 contiguous bytes (4 full cache lines) instead of a quarter of 16 lines
        for (int c = lane; c < C; c += 64) {
            const float g = o[c] / count;
            for (int iy = 0; iy < gh; ++iy) {
                const float y = sh + (float)ph * bin_h + ((float)iy + 0.5f) * bin_h / (float)gh;
                for (int ix = 0; ix < gw; ++ix) {
                    const float x = sw + (float)pw * bin_w + ((float)ix + 0.5f) * bin_w / (float)gw;
                    const Tap t = make_tap(y, x, H, W);
                    if (!t.ok) continue;
                    atomicAdd(feat + ((long)t.y0 * W + t.x0) * C + c, g * t.w00);
                    atomicAdd(feat + ((long)t.y0 * W + t.x1) * C + c, g * t.w01);
                    atomicAdd(feat + ((long)t.y1 * W + t.x0) * C + c, g * t.w10);
                    atomicAdd(feat + ((long)t.y1 * W + t.x1) * C + c, g * t.w11);
                }
            }
        }
        return;
    }
    const int C4 = C >> 2;
    for (int c4 = lane; c4 < C4; c4 += 64) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int iy = 0; iy < gh; ++iy) {
            const float y = sh + (float)ph * bin_h + ((float)iy + 0.5f) * bin_h / (float)gh;
            for (int ix = 0; ix < gw; ++ix) {
                const float x = sw + (float)pw * bin_w + ((float)ix + 0.5f) * bin_w / (float)gw;
                const Tap t = make_tap(y, x, H, W);
                if (!t.ok) continue;
                const float* p00 = feat + ((long)t.y0 * W + t.x0) * C + 4 * c4;
                const float* p01 = feat + ((long)t.y0 * W + t.x1) * C + 4 * c4;
                const float* p10 = feat + ((long)t.y1 * W + t.x0) * C + 4 * c4;
                const float* p11 = feat + ((long)t.y1 * W + t.x1) * C + 4 * c4;
                const float4 a = *reinterpret_cast<const float4*>(p00), b = *reinterpret_cast<const float4*>(p01);
                const float4 c = *reinterpret_cast<const float4*>(p10), d = *reinterpret_cast<const float4*>(p11);
                acc.x += t.w00 * a.x + t.w01 * b.x + t.w10 * c.x + t.w11 * d.x;
                acc.y += t.w00 * a.y + t.w01 * b.y + t.w10 * c.y + t.w11 * d.y;
                acc.z += t.w00 * a.z + t.w01 * b.z + t.w10 * c.z + t.w11 * d.z;
                acc.w += t.w00 * a.w + t.w01 * b.w + t.w10 * c.w + t.w11 * d.w;
            }
        }
        acc.x /= count; acc.y /= count; acc.z /= count; acc.w /= count;
        *reinterpret_cast<float4*>(o + 4 * c4) = acc;
        // round 6: the first `first` ROIs of every block of `per_image` also land in a second, contiguous tensor (the cube head
        // pools the same boxes as the box head with an identical pooler: its features were a 25.7 MB slice copy of `out` before)
        if (DIR == 0 && out2 != nullptr && (r % per_image) < first)
            *reinterpret_cast<float4*>(out2 + (((long)(r / per_image) * first + (r % per_image)) * P * P + bin) * C + 4 * c4) = acc;
    }
}

// ---- separable backward ---------------------------------------------------------------------------
// The bilinear weight of sample (iy, ix) at pixel (y, x) factors into wy(iy, y) * wx(ix, x) -- also through
// the clamp / edge-snap / out-of-range rules of make_tap, which act per axis -- and the samples of a bin form
// a grid, so the bin's footprint weights are the outer product WY_ph(y) * WX_pw(x) of two 1-D sums.  Hence
//     dfeat[y][x] += sum_ph WY_ph(y) * ( sum_pw WX_pw(x) * g[ph][pw] )
// and the whole ROI needs ONE atomic per footprint pixel and channel instead of 4 per sample
// (~(rh+2)(rw+2) vs 4*P*P*gh*gw: 3-3.5x fewer atomics, which are what bounds this kernel).
// One wave = one (roi, 64-channel group); the lanes own channels (256 contiguous bytes per atomic instruction);
// WY / WX are wave-uniform tables in LDS, built by 2*P lanes.
constexpr int SEP_MAXF = 144;     // footprint rows / columns held in the LDS tables (else: per-sample fallback)

struct Tap1 { int lo, hi; float wlo, whi; bool ok; };
__device__ __forceinline__ Tap1 make_tap1(float v, int L) {
    Tap1 t;
    t.ok = !(v < -1.0f || v > (float)L);
    if (v <= 0.f) v = 0.f;
    int lo = (int)v, hi;
    if (lo >= L - 1) { hi = lo = L - 1; v = (float)lo; } else { hi = lo + 1; }
    const float l = v - (float)lo;
    t.lo = lo; t.hi = hi; t.whi = l; t.wlo = 1.f - l;
    return t;
}

template <int PP>
__global__ void __launch_bounds__(256) roi_align_bwd_sep_kernel(FeatLevels fl, const float* __restrict__ rois,
                                                                const int* __restrict__ batch_idx,
                                                                const int* __restrict__ levels, int R, int C,
                                                                const float* __restrict__ dout, const float* __restrict__ dout2,
                                                                int per_image, int first) {
    // dout (R, PP, PP, C) [nullable]; dout2 [nullable]: a second gradient for the first `first` ROIs of every block of `per_image`
    // (the cube head pools a prefix of the box head's ROIs from the same ROIAlign pass): added on the fly instead of being merged
    // into a copy of dout first
    __shared__ __attribute__((aligned(16))) float s_w[4][2][SEP_MAXF][8];    // [wave][y|x][pixel][bin]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int CG = (C + 63) / 64;
    const long job = (long)blockIdx.x * 4 + wave;
    const bool live = job < (long)R * CG;
    const int r = live ? (int)(job / CG) : 0, cg = live ? (int)(job % CG) : 0;
    const int c = cg * 64 + lane;
    const int l = levels[r];
    const int H = fl.H[l], W = fl.W[l];
    const float sc = fl.scale[l];
    const float sw = rois[4 * r + 0] * sc - 0.5f, sh = rois[4 * r + 1] * sc - 0.5f;
    const float ew = rois[4 * r + 2] * sc - 0.5f, eh = rois[4 * r + 3] * sc - 0.5f;
    const float rw = ew - sw, rh = eh - sh;
    const float bin_h = rh / (float)PP, bin_w = rw / (float)PP;
    const int gh = (int)ceilf(rh / (float)PP), gw = (int)ceilf(rw / (float)PP);
    const bool any = live && gh > 0 && gw > 0;
    float* feat = fl.f[l] + (long)batch_idx[r] * H * W * C;
    int Y0 = 0, X0 = 0, fh = 0, fw = 0;
    if (any) {
        Y0 = make_tap1(sh + 0.5f * bin_h / (float)gh, H).lo;
        X0 = make_tap1(sw + 0.5f * bin_w / (float)gw, W).lo;
        fh = make_tap1(sh + (float)(PP - 1) * bin_h + ((float)(gh - 1) + 0.5f) * bin_h / (float)gh, H).hi - Y0 + 1;
        fw = make_tap1(sw + (float)(PP - 1) * bin_w + ((float)(gw - 1) + 0.5f) * bin_w / (float)gw, W).hi - X0 + 1;
    }
    const bool sep = any && fh <= SEP_MAXF && fw <= SEP_MAXF;
    float (*wy)[8] = s_w[wave][0];
    float (*wx)[8] = s_w[wave][1];
    if (sep) {
        for (int i = lane; i < fh * 8; i += 64) wy[i >> 3][i & 7] = 0.f;
        for (int i = lane; i < fw * 8; i += 64) wx[i >> 3][i & 7] = 0.f;
    }
    __syncthreads();
    if (sep) {
        if (lane < PP) {
            for (int ix = 0; ix < gw; ++ix) {
                const Tap1 t = make_tap1(sw + (float)lane * bin_w + ((float)ix + 0.5f) * bin_w / (float)gw, W);
                if (!t.ok) continue;
                wx[t.lo - X0][lane] += t.wlo;
                wx[t.hi - X0][lane] += t.whi;
            }
        } else if (lane >= 32 && lane < 32 + PP) {
            const int ph = lane - 32;
            for (int iy = 0; iy < gh; ++iy) {
                const Tap1 t = make_tap1(sh + (float)ph * bin_h + ((float)iy + 0.5f) * bin_h / (float)gh, H);
                if (!t.ok) continue;
                wy[t.lo - Y0][ph] += t.wlo;
                wy[t.hi - Y0][ph] += t.whi;
            }
        }
    }
    __syncthreads();
    if (!any || c >= C) return;
    const float inv_count = 1.f / (float)(gh * gw);
    const float* o = dout != nullptr ? dout + (long)r * PP * PP * C + c : nullptr;
    const float* o2 = nullptr;
    if (dout2 != nullptr) {
        const int img = r / per_image, k = r - img * per_image;
        if (k < first) o2 = dout2 + ((long)img * first + k) * PP * PP * C + c;
    }
    auto gval = [&](int bin) -> float {
        float v = o != nullptr ? o[(long)bin * C] : 0.f;
        if (o2 != nullptr) v += o2[(long)bin * C];
        return v * inv_count;
    };
    if (sep) {
        float g[PP][PP];
#pragma unroll
        for (int ph = 0; ph < PP; ++ph)
#pragma unroll
            for (int pw = 0; pw < PP; ++pw) g[ph][pw] = gval(ph * PP + pw);
        for (int yy = 0; yy < fh; ++yy) {
            float rowT[PP];
#pragma unroll
            for (int pw = 0; pw < PP; ++pw) rowT[pw] = 0.f;
            bool row_any = false;
#pragma unroll
            for (int ph = 0; ph < PP; ++ph) {
                const float w = wy[yy][ph];            // wave-uniform
                if (w != 0.f) {
                    row_any = true;
#pragma unroll
                    for (int pw = 0; pw < PP; ++pw) rowT[pw] += w * g[ph][pw];
                }
            }
            if (!row_any) continue;
            float* frow = feat + ((long)(Y0 + yy) * W + X0) * C + c;
            for (int xx = 0; xx < fw; ++xx) {
                float v = 0.f;
                bool px_any = false;
#pragma unroll
                for (int pw = 0; pw < PP; ++pw) {
                    const float w = wx[xx][pw];        // wave-uniform
                    if (w != 0.f) { px_any = true; v += w * rowT[pw]; }
                }
                if (px_any) atomicAdd(frow + (long)xx * C, v);
            }
        }
        return;
    }
    // footprint larger than the tables: per-sample scatter (same arithmetic as the forward pass)
    for (int bin = 0; bin < PP * PP; ++bin) {
        const int ph = bin / PP, pw = bin % PP;
        const float gv = gval(bin);
        for (int iy = 0; iy < gh; ++iy) {
            const float y = sh + (float)ph * bin_h + ((float)iy + 0.5f) * bin_h / (float)gh;
            for (int ix = 0; ix < gw; ++ix) {
                const float x = sw + (float)pw * bin_w + ((float)ix + 0.5f) * bin_w / (float)gw;
                const Tap t = make_tap(y, x, H, W);
                if (!t.ok) continue;
                atomicAdd(feat + ((long)t.y0 * W + t.x0) * C + c, gv * t.w00);
                atomicAdd(feat + ((long)t.y0 * W + t.x1) * C + c, gv * t.w01);
                atomicAdd(feat + ((long)t.y1 * W + t.x0) * C + c, gv * t.w10);
                atomicAdd(feat + ((long)t.y1 * W + t.x1) * C + c, gv * t.w11);
            }
        }
    }
}

// ---- deterministic backward (round 4): the OUTPUT owns the sum ------------------------------------------------------------
// The scatter above is bound by ~7e7 fp32 atomics per step and the order in which they land -- hence the rounding of every
// feature-gradient element that several ROIs touch -- changes from run to run.  Here the output owns the sum: a workgroup owns an
// 8 x 4 pixel tile of one (level, image) across ALL channels (lane = 4 channels, 32 float4 accumulators in registers) and lists the
// ROIs of that image and level whose footprint meets the tile, IN ROI ORDER.  Each element of dfeat is written exactly once (no
// zero-fill in front, no atomics) and its value depends on the inputs alone.
// Round 4 gave every wave of a workgroup its own tile and let it walk the whole list: the launch was bound by its LONGEST list (the
// benchmark's 2048 ROIs sit almost all on p2, ~10 x 6 pixels each; (8 x 4 tile, ROI) lists: mean 5.9, 90th percentile 16, longest
// 50, tools/probes/roi_bwd_probe.py) times ~5 us per entry -- up to seven rounds of dependent loads, one per bin row, more where a
// ROI carries the cube head's gradient as well and every bin's second load met its first in an add.  Round 5: the four waves of a
// workgroup SHARE one tile and deal its list out (wave s takes entries s, s + 4, ...: chains of <= 13), then meet through LDS in
// wave order -- the sum of an element is ((S0 + S1) + S2) + S3 with S_s the ascending-ROI sum of wave s, a function of the ROI set
// alone; a bin row's loads (7 from each gradient tensor) are issued as one block before anything waits.  Alone on the device: 272
// -> 136 us (tools/probes/roi_bwd_probe.py); in the two-stream step 10.81 -> 10.77 ms only -- the chain-bound form left most of the
// machine to the weight-gradient stream beside it (profiles/r05_ab_roi_gather.log).
struct GatherTiles {
    int off[MAXL + 1];        // first patch of each level (a patch = GT_PW x GT_PH tiles = 32 x 32 pixels of one image)
    int px[MAXL], py[MAXL];   // patches per image along x / y
    int tx[MAXL], ty[MAXL];   // tiles per image along x / y
    int B;
};
constexpr int GT_TW = 8, GT_TH = 4, GT_SLICES = 4, GT_MAXR = 4096;
constexpr int GT_PW = 4, GT_PH = 8, GT_PT = GT_PW * GT_PH;     // tiles per patch

// per ROI, once: (image << 8 | level, or -1 for an empty sampling grid), footprint rows Y0 | Y1 << 16, columns X0 | X1 << 16, sampling
// grid gh | gw << 16 -- what every region's list building compares against -- and the ROI's start / bin size on its level
template <int PP>
__global__ void roi_footprint_kernel(FeatLevels fl, const float* __restrict__ rois, const int* __restrict__ batch_idx,
                                     const int* __restrict__ levels, int R, int4* __restrict__ fp, float4* __restrict__ par) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int l = levels[r];
    const int H = fl.H[l], W = fl.W[l];
    const float sc = fl.scale[l];
    const float sw = rois[4 * r + 0] * sc - 0.5f, sh = rois[4 * r + 1] * sc - 0.5f;
    const float rw = rois[4 * r + 2] * sc - 0.5f - sw, rh = rois[4 * r + 3] * sc - 0.5f - sh;
    const float bin_h = rh / (float)PP, bin_w = rw / (float)PP;
    const int gh = (int)ceilf(rh / (float)PP), gw = (int)ceilf(rw / (float)PP);
    int4 o = make_int4(-1, 0, 0, 0);
    if (gh > 0 && gw > 0) {
        const int Y0 = make_tap1(sh + 0.5f * bin_h / (float)gh, H).lo, X0 = make_tap1(sw + 0.5f * bin_w / (float)gw, W).lo;
        const int Y1 = make_tap1(sh + (float)(PP - 1) * bin_h + ((float)(gh - 1) + 0.5f) * bin_h / (float)gh, H).hi;
        const int X1 = make_tap1(sw + (float)(PP - 1) * bin_w + ((float)(gw - 1) + 0.5f) * bin_w / (float)gw, W).hi;
        o = make_int4((batch_idx[r] << 8) | l, Y0 | (Y1 << 16), X0 | (X1 << 16), gh | (gw << 16));
    }
    fp[r] = o;
    par[r] = make_float4(sw, sh, bin_w, bin_h);
}

template <int PP>
__global__ void __launch_bounds__(256) roi_align_bwd_gather_kernel(FeatLevels fl, GatherTiles gt, const float* __restrict__ rois,
                                                                   int R, int C, const float* __restrict__ dout,
                                                                   const float* __restrict__ dout2, int per_image, int first,
                                                                   const int4* __restrict__ fp, const float4* __restrict__ par) {
    __shared__ unsigned char s_hit[GT_MAXR];
    __shared__ unsigned short s_list[GT_MAXR];
    __shared__ int s_count;
    __shared__ float4 s_acc[GT_TH * GT_TW][64];         // one wave's accumulators on their way to wave 0
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // workgroup -> tile: the tiles that meet one ROI are neighbours and read the same bins, and an XCD's L2 only helps the workgroups
    // of that XCD (blockIdx % 8) -- so a 32 x 32 pixel patch of tiles goes to ONE XCD, as 32 consecutive workgroups of it, and the
    // patches go round the XCDs (ROI density varies over the image; neighbouring patches on different XCDs level that out)
    const int xcd = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
    const int patch = (j / GT_PT) * 8 + xcd, tin = j % GT_PT;
    if (patch >= gt.off[MAXL]) return;
    int l = 0;
    while (l + 1 < fl.nlev && patch >= gt.off[l + 1]) ++l;
    int t = patch - gt.off[l];
    const int pix = t % gt.px[l]; t /= gt.px[l];
    const int piy = t % gt.py[l];
    const int n = t / gt.py[l];
    const int tix = pix * GT_PW + tin % GT_PW, tiy = piy * GT_PH + tin / GT_PW;
    if (tix >= gt.tx[l] || tiy >= gt.ty[l]) return;
    const int H = fl.H[l], W = fl.W[l];
    const int ty0 = tiy * GT_TH, tx0 = tix * GT_TW;
    // -- 1. which ROIs touch this tile (every thread tests R / 256 precomputed footprints), compacted in ascending ROI index
    const int key = (n << 8) | l;
    for (int r = tid; r < R; r += 256) {
        const int4 f = fp[r];
        const int Y0 = f.y & 0xffff, Y1 = f.y >> 16, X0 = f.z & 0xffff, X1 = f.z >> 16;
        s_hit[r] = (f.x == key && Y0 < ty0 + GT_TH && Y1 >= ty0 && X0 < tx0 + GT_TW && X1 >= tx0) ? 1 : 0;
    }
    __syncthreads();
    if (wave == 0) {
        int count = 0;
        for (int base = 0; base < R; base += 64) {
            const int r = base + lane;
            const bool h = r < R && s_hit[r] != 0;
            const unsigned long long m = __ballot(h);
            if (h) s_list[count + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)r;
            count += __popcll(m);
        }
        if (lane == 0) s_count = count;
    }
    __syncthreads();
    const int count = s_count;
    // -- 2. the four waves deal the list out: wave s adds entries s, s + GT_SLICES, ... (ascending) into its own accumulators; lane =
    // 4 channels.  The tile-local weights of a ROI live in two registers -- lane 8 * yy + ph holds WY[yy][ph], the summed y-weights
    // of bin ph's samples on tile row yy; lane 8 * xx + pw the same for x -- and reach the FMAs as wave-uniform scalars through
    // v_readlane.
    const int tr = lane >> 3, tb = lane & 7;
    const int c = 4 * lane;
    const bool cok = c < C;
    float4 acc[GT_TH * GT_TW];
#pragma unroll
    for (int i = 0; i < GT_TH * GT_TW; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q = wave; q < count; q += GT_SLICES) {
        const int r = s_list[q];
        const int4 f = fp[r];
        const float4 pa = par[r];
        const float sw = pa.x, sh = pa.y, bin_w = pa.z, bin_h = pa.w;
        const int gh = f.w & 0xffff, gw = f.w >> 16;
        float wyv = 0.f, wxv = 0.f;
        if (tb < PP) {
            if (tr < GT_TH)
                for (int iy = 0; iy < gh; ++iy) {             // samples in ascending order, like the scatter kernel's table build
                    const Tap1 tp = make_tap1(sh + (float)tb * bin_h + ((float)iy + 0.5f) * bin_h / (float)gh, H);
                    if (!tp.ok) continue;
                    if (tp.lo - ty0 == tr) wyv += tp.wlo;
                    if (tp.hi - ty0 == tr) wyv += tp.whi;
                }
            for (int ix = 0; ix < gw; ++ix) {
                const Tap1 tp = make_tap1(sw + (float)tb * bin_w + ((float)ix + 0.5f) * bin_w / (float)gw, W);
                if (!tp.ok) continue;
                if (tp.lo - tx0 == tr) wxv += tp.wlo;
                if (tp.hi - tx0 == tr) wxv += tp.whi;
            }
        }
        const unsigned long long ymask = __ballot(wyv != 0.f), xmask = __ballot(wxv != 0.f);
        if (ymask == 0 || xmask == 0) continue;
        // bin rows ph of the 7 x 7 grid that carry weight on this tile
        unsigned phm = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) phm |= (unsigned)(ymask >> (8 * k)) & 0x7fu;
        const float inv_count = 1.f / (float)(gh * gw);
        const float* o = dout != nullptr ? dout + (long)r * PP * PP * C + c : nullptr;           // (wave-uniform tests: the lanes
        const float* o2 = nullptr;                                                               //  past C are masked at the loads)
        if (dout2 != nullptr) {
            const int img = r / per_image, k = r - img * per_image;
            if (k < first) o2 = dout2 + ((long)img * first + k) * PP * PP * C + c;
        }
        {
#pragma clang fp contract(fast)
#pragma unroll
            for (int ph = 0; ph < PP; ++ph) {
                if (!((phm >> ph) & 1u)) continue;
                // the whole bin row of both gradient tensors as one block of loads (a bin without weight on this tile costs a load,
                // not a round trip: its weights are zero)
                float4 g[PP], g2[PP];
#pragma unroll
                for (int pw = 0; pw < PP; ++pw) g[pw] = g2[pw] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (o != nullptr && cok) {
#pragma unroll
                    for (int pw = 0; pw < PP; ++pw) g[pw] = *reinterpret_cast<const float4*>(o + (long)(ph * PP + pw) * C);
                }
                if (o2 != nullptr && cok) {
#pragma unroll
                    for (int pw = 0; pw < PP; ++pw) g2[pw] = *reinterpret_cast<const float4*>(o2 + (long)(ph * PP + pw) * C);
                }
                if (o2 != nullptr) {
#pragma unroll
                    for (int pw = 0; pw < PP; ++pw) { g[pw].x += g2[pw].x; g[pw].y += g2[pw].y; g[pw].z += g2[pw].z; g[pw].w += g2[pw].w; }
                }
                // T[xx] = sum_pw WX[xx][pw] * g[ph][pw], then acc[yy][xx] += WY[yy][ph] / count * T[xx]
#pragma unroll
                for (int xx = 0; xx < GT_TW; ++xx) {
                    if (((xmask >> (8 * xx)) & 0x7fULL) == 0) continue;
                    float4 T = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int pw = 0; pw < PP; ++pw) {
                        const float w = omni_readlane(wxv, xx * 8 + pw);
                        T.x += w * g[pw].x; T.y += w * g[pw].y; T.z += w * g[pw].z; T.w += w * g[pw].w;
                    }
#pragma unroll
                    for (int yy = 0; yy < GT_TH; ++yy) {
                        const float w = omni_readlane(wyv, yy * 8 + ph) * inv_count;
                        float4& a = acc[yy * GT_TW + xx];
                        a.x += w * T.x; a.y += w * T.y; a.z += w * T.z; a.w += w * T.w;
                    }
                }
            }
        }
    }
    // -- 3. the waves meet in wave order: ((S0 + S1) + S2) + S3
    for (int s = 1; s < GT_SLICES; ++s) {
        if (count <= s) break;                                  // (workgroup-uniform: wave s had no entry, nor has any later one)
        if (s > 1) __syncthreads();                             // wave 0 has read the previous wave's values
        if (wave == s) {
#pragma unroll
            for (int i = 0; i < GT_TH * GT_TW; ++i) s_acc[i][lane] = acc[i];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int i = 0; i < GT_TH * GT_TW; ++i) {
                const float4 v = s_acc[i][lane];
                acc[i].x += v.x; acc[i].y += v.y; acc[i].z += v.z; acc[i].w += v.w;
            }
        }
    }
    if (wave == 0 && cok) {
        float* feat = fl.f[l] + (long)n * H * W * C + c;
#pragma unroll
        for (int yy = 0; yy < GT_TH; ++yy)
#pragma unroll
            for (int xx = 0; xx < GT_TW; ++xx)
                if (ty0 + yy < H && tx0 + xx < W) *reinterpret_cast<float4*>(feat + ((long)(ty0 + yy) * W + tx0 + xx) * C) = acc[yy * GT_TW + xx];
    }
}

FeatLevels make_feat(const void* const* ptrs, const int* hw, const float* scales, int nlev) {
    FeatLevels fl;
    for (int l = 0; l < MAXL; ++l) { fl.f[l] = nullptr; fl.H[l] = fl.W[l] = 0; fl.scale[l] = 0.f; }
    for (int l = 0; l < nlev; ++l) { fl.f[l] = (float*)ptrs[l]; fl.H[l] = hw[2 * l]; fl.W[l] = hw[2 * l + 1]; fl.scale[l] = scales[l]; }
    fl.nlev = nlev;
    return fl;
}

// ---- POOLER_TYPE "ROIPool" (round 6) ------------------------------------------------------------------------------------------------
// torchvision.ops.roi_pool, the operator detectron2's ROIPooler builds for POOLER_TYPE "ROIPool" (cubercnn/modeling/roi_heads/
// roi_heads.py:166-171 passes MODEL.ROI_BOX_HEAD.POOLER_TYPE / MODEL.ROI_CUBE_HEAD.POOLER_TYPE through): ROI corners rounded to whole
// feature pixels (round half away from zero), a ROI of at least 1 x 1, bin (ph, pw) = rows [floor(ph * bh), ceil((ph + 1) * bh)) and
// the same for columns (bh = roi_height / P in fp32), shifted by the ROI start and clipped to the map; the maximum over the bin
// (strictly-greater scan in row-major order, so the FIRST maximum wins), an empty bin gives 0 with argmax -1.  The backward hands a
// bin's gradient to its argmax pixel (fp32 atomics into zeroed maps: overlapping ROIs share pixels).
// One wave per (roi, bin); lane l owns channels 4l .. 4l+3 (+256 per round): a wave instruction reads one pixel's 1 KB run.
template <int DIR>
__global__ void __launch_bounds__(256) roi_pool_kernel(FeatLevels fl, const float* __restrict__ rois, const int* __restrict__ batch_idx,
                                                       const int* __restrict__ levels, int R, int P, int C, float* __restrict__ out,
                                                       int* __restrict__ argmax) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long job = (long)blockIdx.x * 4 + wave;   // (roi, bin)
    if (job >= (long)R * P * P) return;
    const int r = (int)(job / (P * P)), bin = (int)(job % (P * P));
    const int ph = bin / P, pw = bin % P;
    const int l = levels[r];
    const int H = fl.H[l], W = fl.W[l];
    float* feat = fl.f[l] + (long)batch_idx[r] * H * W * C;
    const int C4 = C >> 2;
    if (DIR == 1) {
        for (int c4 = lane; c4 < C4; c4 += 64) {
            const float4 g = *reinterpret_cast<const float4*>(out + job * C + 4 * c4);
            const int4 a = *reinterpret_cast<const int4*>(argmax + job * C + 4 * c4);
            if (a.x >= 0) atomicAdd(feat + (long)a.x * C + 4 * c4 + 0, g.x);
            if (a.y >= 0) atomicAdd(feat + (long)a.y * C + 4 * c4 + 1, g.y);
            if (a.z >= 0) atomicAdd(feat + (long)a.z * C + 4 * c4 + 2, g.z);
            if (a.w >= 0) atomicAdd(feat + (long)a.w * C + 4 * c4 + 3, g.w);
        }
        return;
    }
    const float sc = fl.scale[l];
    const int rsw = (int)roundf(rois[4 * r + 0] * sc), rsh = (int)roundf(rois[4 * r + 1] * sc);
    const int rew = (int)roundf(rois[4 * r + 2] * sc), reh = (int)roundf(rois[4 * r + 3] * sc);
    const int rw = max(rew - rsw + 1, 1), rh = max(reh - rsh + 1, 1);     // malformed ROIs are 1 x 1
    const float bh = (float)rh / (float)P, bw = (float)rw / (float)P;
    int hs = (int)floorf((float)ph * bh), he = (int)ceilf((float)(ph + 1) * bh);
    int ws = (int)floorf((float)pw * bw), we = (int)ceilf((float)(pw + 1) * bw);
    hs = min(max(hs + rsh, 0), H); he = min(max(he + rsh, 0), H);
    ws = min(max(ws + rsw, 0), W); we = min(max(we + rsw, 0), W);
    const bool empty = he <= hs || we <= ws;
    for (int c4 = lane; c4 < C4; c4 += 64) {
        const float init = empty ? 0.f : -3.402823466e+38f;
        float4 m = make_float4(init, init, init, init);
        int4 a = make_int4(-1, -1, -1, -1);
        for (int h = hs; h < he; ++h)
            for (int w = ws; w < we; ++w) {
                const int idx = h * W + w;
                const float4 v = *reinterpret_cast<const float4*>(feat + (long)idx * C + 4 * c4);
                if (v.x > m.x) { m.x = v.x; a.x = idx; }
                if (v.y > m.y) { m.y = v.y; a.y = idx; }
                if (v.z > m.z) { m.z = v.z; a.z = idx; }
                if (v.w > m.w) { m.w = v.w; a.w = idx; }
            }
        *reinterpret_cast<float4*>(out + job * C + 4 * c4) = m;
        *reinterpret_cast<int4*>(argmax + job * C + 4 * c4) = a;
    }
}

}  // namespace

extern "C" {

// detectron2 assign_boxes_to_levels: levels[r] in [0, max_level - min_level].
int omni_roi_levels(const float* rois, int R, int min_level, int max_level, float canonical_size, int canonical_level,
                    int* levels, void* stream) {
    if (R < 0 || max_level < min_level) return OMNI_ERR_ARG;
    if (R == 0) return OMNI_OK;
    hipLaunchKernelGGL(roi_levels_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, rois, R, min_level,
                       max_level, canonical_size, canonical_level, levels);
    return omni_launch_status();
}

// level_ptrs[l]: (B, H_l, W_l, C) NHWC; level_hw: host ints (H_0, W_0, H_1, ...); level_scale: host floats
// (1/stride).  rois (R,4) XYXY, batch_idx (R), levels (R) from omni_roi_levels.  out (R, P, P, C).
int omni_roi_align_fwd(const void* const* level_ptrs, const int* level_hw, const float* level_scale, int nlev,
                       const float* rois, const int* batch_idx, const int* levels, int R, int P, int C, float* out,
                       void* stream) {
    if (nlev <= 0 || nlev > MAXL || (C & 3) || P <= 0) return OMNI_ERR_ARG;
    if (R == 0) return OMNI_OK;
    FeatLevels fl = make_feat(level_ptrs, level_hw, level_scale, nlev);
    const long jobs = (long)R * P * P;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_align_kernel<0>), dim3((unsigned)((jobs + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, fl, rois, batch_idx, levels, R, P, C, out);
    return omni_launch_status();
}

// The same, ALSO writing the first `first` ROIs of every block of `per_image` consecutive ROIs to out2 ((R / per_image) * first, P, P, C):
// one pass for the box head's and the cube head's pooled features (two ROIPoolers with the same resolution / sampling ratio on the
// same boxes, cubercnn/modeling/roi_heads/roi_heads.py:166-171, 267, 362).
int omni_roi_align_fwd2(const void* const* level_ptrs, const int* level_hw, const float* level_scale, int nlev,
                        const float* rois, const int* batch_idx, const int* levels, int R, int P, int C, float* out,
                        float* out2, int per_image, int first, void* stream) {
    if (nlev <= 0 || nlev > MAXL || (C & 3) || P <= 0) return OMNI_ERR_ARG;
    if (out2 != nullptr && (per_image <= 0 || first < 0 || first > per_image || R % per_image != 0)) return OMNI_ERR_ARG;
    if (R == 0) return OMNI_OK;
    FeatLevels fl = make_feat(level_ptrs, level_hw, level_scale, nlev);
    const long jobs = (long)R * P * P;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_align_kernel<0>), dim3((unsigned)((jobs + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, fl, rois, batch_idx, levels, R, P, C, out, out2, per_image, first);
    return omni_launch_status();
}

// Round 6: the two general-resolution kernels with the `aligned` switch of torchvision's roi_align exposed -- aligned = 0 is
// detectron2's POOLER_TYPE "ROIAlign" (cubercnn/modeling/roi_heads/roi_heads.py:166-171 passes the config key through).  The
// backward is the atomic form (caller zeroes dlevel_ptrs): the owner-computes kernels serve the aligned form only.
int omni_roi_align_fwd_mode(const void* const* level_ptrs, const int* level_hw, const float* level_scale, int nlev,
                            const float* rois, const int* batch_idx, const int* levels, int R, int P, int C, int aligned, float* out,
                            void* stream) {
    if (nlev <= 0 || nlev > MAXL || (C & 3) || P <= 0) return OMNI_ERR_ARG;
    if (R == 0) return OMNI_OK;
    FeatLevels fl = make_feat(level_ptrs, level_hw, level_scale, nlev);
    const long jobs = (long)R * P * P;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_align_kernel<0>), dim3((unsigned)((jobs + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, fl, rois, batch_idx, levels, R, P, C, out, (float*)nullptr, 0, 0, aligned != 0);
    return omni_launch_status();
}
int omni_roi_align_bwd_mode(const void* const* dlevel_ptrs, const int* level_hw, const float* level_scale, int nlev,
                            const float* rois, const int* batch_idx, const int* levels, int R, int P, int C, int aligned,
                            const float* dout, void* stream) {
    if (nlev <= 0 || nlev > MAXL || (C & 3) || P <= 0 || dout == nullptr) return OMNI_ERR_ARG;
    if (R == 0) return OMNI_OK;
    FeatLevels fl = make_feat(dlevel_ptrs, level_hw, level_scale, nlev);
    const long jobs = (long)R * P * P;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_align_kernel<1>), dim3((unsigned)((jobs + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, fl, rois, batch_idx, levels, R, P, C, const_cast<float*>(dout), (float*)nullptr, 0, 0, aligned != 0);
    return omni_launch_status();
}

// out (R, P, P, C) and argmax (R, P, P, C) int32 = h * W + w of the winning pixel in the ROI's image and level, -1 for an empty bin
int omni_roi_pool_fwd(const void* const* level_ptrs, const int* level_hw, const float* level_scale, int nlev, const float* rois,
                      const int* batch_idx, const int* levels, int R, int P, int C, float* out, int* argmax, void* stream) {
    if (nlev <= 0 || nlev > MAXL || (C & 3) || P <= 0 || out == nullptr || argmax == nullptr) return OMNI_ERR_ARG;
    if (R == 0) return OMNI_OK;
    FeatLevels fl = make_feat(level_ptrs, level_hw, level_scale, nlev);
    const long jobs = (long)R * P * P;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_pool_kernel<0>), dim3((unsigned)((jobs + 3) / 4)), dim3(256), 0, (hipStream_t)stream, fl, rois,
                       batch_idx, levels, R, P, C, out, argmax);
    return omni_launch_status();
}
// dlevel_ptrs[l] += the gradient of every bin at its argmax pixel (fp32 atomics; the caller zeroes the maps)
int omni_roi_pool_bwd(const void* const* dlevel_ptrs, const int* level_hw, const float* level_scale, int nlev, const int* batch_idx,
                      const int* levels, int R, int P, int C, const float* dout, const int* argmax, void* stream) {
    if (nlev <= 0 || nlev > MAXL || (C & 3) || P <= 0 || dout == nullptr || argmax == nullptr) return OMNI_ERR_ARG;
    if (R == 0) return OMNI_OK;
    FeatLevels fl = make_feat(dlevel_ptrs, level_hw, level_scale, nlev);
    const long jobs = (long)R * P * P;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_pool_kernel<1>), dim3((unsigned)((jobs + 3) / 4)), dim3(256), 0, (hipStream_t)stream, fl,
                       (const float*)nullptr, batch_idx, levels, R, P, C, const_cast<float*>(dout), const_cast<int*>(argmax));
    return omni_launch_status();
}

// dlevel_ptrs[l] (same shapes as the features) are ACCUMULATED into with fp32 atomics (caller zeroes).
static int roi_align_bwd_impl(const void* const* dlevel_ptrs, const int* level_hw, const float* level_scale, int nlev,
                              const float* rois, const int* batch_idx, const int* levels, int R, int P, int C, const float* dout,
                              const float* dout2, int per_image, int first, void* stream) {
    if (nlev <= 0 || nlev > MAXL || (C & 3) || P <= 0 || (dout == nullptr && dout2 == nullptr)) return OMNI_ERR_ARG;
    if (dout2 != nullptr && (per_image <= 0 || first < 0 || first > per_image || R % per_image != 0)) return OMNI_ERR_ARG;
    if (R == 0) return OMNI_OK;
    FeatLevels fl = make_feat(dlevel_ptrs, level_hw, level_scale, nlev);
    if (P == 7) {   // the pooler resolution of every Cube R-CNN config: separable, one atomic per footprint pixel
        const long jobs = (long)R * ((C + 63) / 64);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_align_bwd_sep_kernel<7>), dim3((unsigned)((jobs + 3) / 4)), dim3(256), 0,
                           (hipStream_t)stream, fl, rois, batch_idx, levels, R, C, dout, dout2, per_image, first);
        return omni_launch_status();
    }
    if (dout == nullptr || dout2 != nullptr) return OMNI_ERR_ARG;          // the general-P kernel takes one gradient
    const long jobs = (long)R * P * P;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_align_kernel<1>), dim3((unsigned)((jobs + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, fl, rois, batch_idx, levels, R, P, C, const_cast<float*>(dout));
    return omni_launch_status();
}

// Deterministic form of omni_roi_align_bwd2 (P == 7, R <= 4096, C <= 256, B <= 255 images): dlevel_ptrs[l] (B, H_l, W_l, C) are
// OVERWRITTEN -- every element exactly once, by the workgroup that owns its 8 x 4 tile (four waves deal out the tile's ROIs in
// ascending ROI index and meet in wave order).  No zero-fill by the caller, no atomics; two runs give bit-identical feature gradients.  ws: scratch for
// the per-ROI footprint records (plan != NULL: plan[3] = floats needed, nothing is launched); ctr / n_ctr are unused.
int omni_roi_align_bwd_det(const void* const* dlevel_ptrs, const int* level_hw, const float* level_scale, int nlev, int B,
                           const float* rois, const int* batch_idx, const int* levels, int R, int P, int C, const float* dout,
                           const float* dout2, int per_image, int first, float* ws, long long ws_floats, int* ctr, int n_ctr,
                           long long* plan, void* stream) {
    (void)ctr; (void)n_ctr;
    if (nlev <= 0 || nlev > MAXL || (C & 3) || C > 256 || P != 7 || B <= 0 || B > 255 || R < 0 || R > GT_MAXR || (dout == nullptr && dout2 == nullptr)) return OMNI_ERR_ARG;
    if (dout2 != nullptr && (per_image <= 0 || first < 0 || first > per_image || R % per_image != 0)) return OMNI_ERR_ARG;
    FeatLevels fl = make_feat(dlevel_ptrs, level_hw, level_scale, nlev);
    GatherTiles gt;
    int total = 0;                                            // patches
    for (int l = 0; l < MAXL; ++l) {
        gt.off[l] = total;
        gt.tx[l] = gt.ty[l] = gt.px[l] = gt.py[l] = 1;
        if (l < nlev) {
            if (fl.H[l] > 65535 || fl.W[l] > 65535) return OMNI_ERR_ARG;
            gt.tx[l] = (fl.W[l] + GT_TW - 1) / GT_TW;
            gt.ty[l] = (fl.H[l] + GT_TH - 1) / GT_TH;
            gt.px[l] = (gt.tx[l] + GT_PW - 1) / GT_PW;
            gt.py[l] = (gt.ty[l] + GT_PH - 1) / GT_PH;
            const long patches = (long)B * gt.px[l] * gt.py[l];
            if (total + patches > (1L << 22)) return OMNI_ERR_ARG;
            total += (int)patches;
        }
    }
    for (int l = nlev; l <= MAXL; ++l) gt.off[l] = total;
    gt.B = B;
    const long long need = 8LL * R + 4;                       // per-ROI records: int4 footprint + float4 parameters
    if (plan != nullptr) { plan[0] = plan[1] = plan[2] = 0; plan[3] = need; return OMNI_OK; }
    if (total == 0) return OMNI_OK;
    if (ws == nullptr || ws_floats < need) return OMNI_ERR_ARG;
    int4* fp = reinterpret_cast<int4*>(ws);
    float4* par = reinterpret_cast<float4*>(ws + 4LL * R);
    if (R > 0)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_footprint_kernel<7>), dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, fl, rois, batch_idx,
                           levels, R, fp, par);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_align_bwd_gather_kernel<7>), dim3((unsigned)((total + 7) / 8 * 8 * GT_PT)), dim3(256), 0, (hipStream_t)stream, fl, gt, rois, R,
                       C, dout, dout2, per_image, first, (const int4*)fp, (const float4*)par);
    return omni_launch_status();
}

int omni_roi_align_bwd(const void* const* dlevel_ptrs, const int* level_hw, const float* level_scale, int nlev,
                       const float* rois, const int* batch_idx, const int* levels, int R, int P, int C, const float* dout,
                       void* stream) {
    return roi_align_bwd_impl(dlevel_ptrs, level_hw, level_scale, nlev, rois, batch_idx, levels, R, P, C, dout, nullptr, 0, 0, stream);
}

// The same with two gradient tensors (P == 7): dout (R, 7, 7, C) [nullable] and dout2 ((R / per_image) * first, 7, 7, C) [nullable]
// for the first `first` ROIs of every block of `per_image` -- the box head's and the cube head's gradients of one shared ROIAlign
// pass (roi_heads.py:249-357 pools the cube head's ROIs separately; here they are a prefix of the box head's).
int omni_roi_align_bwd2(const void* const* dlevel_ptrs, const int* level_hw, const float* level_scale, int nlev,
                        const float* rois, const int* batch_idx, const int* levels, int R, int P, int C, const float* dout,
                        const float* dout2, int per_image, int first, void* stream) {
    return roi_align_bwd_impl(dlevel_ptrs, level_hw, level_scale, nlev, rois, batch_idx, levels, R, P, C, dout, dout2, per_image, first,
                              stream);
}

}  // extern "C"
