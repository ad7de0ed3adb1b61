"""Launchers for csrc/winograd.hip + the batched GEMM entry points of csrc/conv_gemm.hip: Winograd F(2x2, 3x3)
forward / data-gradient / weight-gradient of a 3x3, stride-1, pad-1 convolution on NHWC tensors."""
import torch

from .. import lib as _lib

CL = torch.channels_last


import os as _os

_F43 = _os.environ.get("OMNI_WINOGRAD_F43", "1") != "0"
_F43_MIN_TILES = int(_os.environ.get("OMNI_WINOGRAD_F43_MIN_TILES", "256"))     # measured: 1024 -> 14.18, 256 -> 14.02, 64 -> 14.05 ms / step


_MIN_TILES = int(_os.environ.get("OMNI_WINO_MIN_TILES", "256"))                 # 2x2 tiles a map needs for the Winograd path at all
_DGRAD_MIN_TILES = int(_os.environ.get("OMNI_WINO_DGRAD_MIN_TILES", "256"))     # ... and for the Winograd data gradient (round 3: 1024 -> 256,
#   the deep-prefetch point GEMMs made the 16x16 maps pay: 13.60 -> 13.55 ms / step, profiles/r03_ab_thresholds.log)


def eligible(x_shape, w_shape, stride, pad):
    """Wide (>= 128 channel) 3x3/s1/p1 layers on even maps with >= 256 tiles: FPN output / RPN convs on p2..p5 and the
    DLA level 3-5 blocks (measured per shape with tools/bench_kernels.py, profiles/README.md)."""
    N, C, H, W = x_shape
    K, _, R, S = w_shape
    if not (R == 3 and S == 3 and stride == 1 and pad == 1 and H % 2 == 0 and W % 2 == 0 and C % 32 == 0 and K % 32 == 0):
        return False
    if C >= 128 and K >= 128 and N * (H // 2) * (W // 2) >= _MIN_TILES:
        return True
    # 64-channel layers (DLA level 2, ResNet layer1) only pay off with the 36-point transform on large maps
    return C >= 64 and K >= 64 and tile_size(x_shape) == 4 and N * (H // 4) * (W // 4) >= 4096


_f22_depth = 0


class f22_only:
    """with wino.f22_only(): convolutions inside take the 16-point F(2x2,3x3) transform whatever the map size.  Its fp32 error is
    ~1/3 of F(4x4,3x3)'s (tools/winograd_error.py); used where a convolution's output feeds an ill-conditioned consumer and speed
    is not the point -- the FPN output convolutions in front of ROIAlign at inference (cubercnn/modeling/backbone/fpn.py)."""

    def __enter__(self):
        global _f22_depth
        _f22_depth += 1

    def __exit__(self, *a):
        global _f22_depth
        _f22_depth -= 1


def tile_size(x_shape):
    """4 = F(4x4,3x3) (36 points, 2.25 multiplies per output) when the map divides into >= 256 tiles of 4x4 (batch 4: maps of
    32x32 and larger), else 2 = F(2x2,3x3) (16 points, 4 multiplies per output)."""
    N, _, H, W = x_shape
    if _f22_depth > 0:
        return 2
    return 4 if (_F43 and H % 4 == 0 and W % 4 == 0 and N * (H // 4) * (W // 4) >= _F43_MIN_TILES) else 2


def dgrad_eligible(x_shape):
    """Below ~256 tiles the direct split-K data-gradient kernel is faster than transform + 16 GEMMs + transform."""
    N, _, H, W = x_shape
    return N * (H // 2) * (W // 2) >= _DGRAD_MIN_TILES


def _nhwc(t):
    assert t.is_contiguous(memory_format=CL), "expected a channels_last tensor"
    return t.permute(0, 2, 3, 1)


def _points(tile):
    return (tile + 2) ** 2


def transform_input(x, tile=2, affine=None, relu=False):
    """x (N,C,H,W) CL -> V (P, T, C), P = (tile + 2)^2 Winograd points, T = N * H/tile * W/tile.
    affine (2C) = [scale | shift]: the transform of relu?(x * scale + shift) -- the BatchNorm(+ReLU) between the convolution that wrote
    x and this one, applied on load (zero padding as for the normalised tensor, which is never stored)"""
    xv = _nhwc(x)
    N, H, W, C = xv.shape
    L = _lib.check_device(xv, affine)
    V = torch.empty((_points(tile), N * (H // tile) * (W // tile), C), dtype=torch.float32, device=x.device)
    if affine is not None:
        assert affine.numel() == 2 * C
        L.call("omni_wino_in_affine", _lib.ptr(xv), _lib.ptr(affine), int(bool(relu)), _lib.ptr(V), N, H, W, C, tile, _lib.stream_of(x))
    else:
        L.call("omni_wino_in", _lib.ptr(xv), _lib.ptr(V), N, H, W, C, tile, _lib.stream_of(x))
    return V


# ---- row-range forms: the tiles of several tensors side by side in one (P, rows_total, C) array, ONE batched GEMM for all of them ----
def levels_tile(shapes):
    """common tile size for the tensors of a shared convolution (the RPN's 3x3 over the FPN levels), or 0: the 36-point transform when
    every map divides into 4x4 tiles and there are enough of them in total, else the 16-point one on even maps"""
    if not shapes or any(H % 2 or W % 2 for _, _, H, W in shapes):
        return 0
    if _f22_depth == 0 and _F43 and all(H % 4 == 0 and W % 4 == 0 for _, _, H, W in shapes) \
            and sum(N * (H // 4) * (W // 4) for N, _, H, W in shapes) >= _F43_MIN_TILES:
        return 4
    return 2 if sum(N * (H // 2) * (W // 2) for N, _, H, W in shapes) >= _MIN_TILES else 0


def level_rows(shapes, tile):
    """-> (first row of every tensor, total rows)"""
    offs, tot = [], 0
    for N, _, H, W in shapes:
        offs.append(tot)
        tot += N * (H // tile) * (W // tile)
    return offs, tot


def _row_ptr(A, row0):
    return A.data_ptr() + 4 * row0 * A.shape[2]


def transform_input_rows(x, V_all, row0, tile):
    """x (N,C,H,W) CL -> rows [row0, row0 + T) of V_all (P, rows_total, C)"""
    xv = _nhwc(x)
    N, H, W, C = xv.shape
    L = _lib.check_device(xv, V_all)
    assert V_all.shape[0] == _points(tile) and V_all.shape[2] == C and row0 + N * (H // tile) * (W // tile) <= V_all.shape[1]
    L.call("omni_wino_in_rows", _lib.ptr(xv), _row_ptr(V_all, row0), N, H, W, C, tile, V_all.shape[1] * C, _lib.stream_of(x))


def transform_output_rows(Mt_all, row0, shape, bias=None, relu=False, carry=None):
    """rows [row0, ...) of Mt_all (P, rows_total, K) -> y (N,K,H,W) CL (+ bias, ReLU | + carry: see transform_output)"""
    tile = 2 if Mt_all.shape[0] == 16 else 4
    N, H, W = shape
    K = Mt_all.shape[2]
    L = _lib.check_device(Mt_all, bias)
    assert row0 + N * (H // tile) * (W // tile) <= Mt_all.shape[1]
    y = torch.empty((N, H, W, K), dtype=torch.float32, device=Mt_all.device)
    if carry is not None:
        assert bias is None and not relu and tuple(carry.shape) == (N, K, H, W)
    L.call("omni_wino_out_rows", _row_ptr(Mt_all, row0), _lib.ptr(bias), carry.data_ptr() if carry is not None else None,
           carry.stride(3) if carry is not None else 0, _lib.ptr(y), N, H, W, K, int(relu), tile, Mt_all.shape[1] * K, _lib.stream_of(Mt_all))
    return y.permute(0, 3, 1, 2)


def transform_dy_in_rows(dy, dM_all, Vd_all, row0, tile):
    """dy (N,K,H,W) CL -> rows [row0, ...) of dM_all and Vd_all (both (P, rows_total, K)): the weight gradient's and the data gradient's
    transforms of dy from one read"""
    dv = _nhwc(dy)
    N, H, W, K = dv.shape
    L = _lib.check_device(dv, dM_all, Vd_all)
    assert dM_all.shape == Vd_all.shape and dM_all.shape[0] == _points(tile) and dM_all.shape[2] == K
    assert row0 + N * (H // tile) * (W // tile) <= dM_all.shape[1]
    L.call("omni_wino_dy_in_rows", _lib.ptr(dv), _row_ptr(dM_all, row0), _row_ptr(Vd_all, row0), N, H, W, K, tile, dM_all.shape[1] * K,
           _lib.stream_of(dy))


def transform_weights(w, want_u=True, want_flip=False, tile=2):
    """w (K,C,3,3) CL (KRSC) -> (U (16,K,C) or None, U' (16,C,K) or None); U' = transform of the rotated, channel-transposed
    filter (the data gradient's weights), produced by the same launch."""
    K, C = w.shape[0], w.shape[1]
    wv = w.permute(0, 2, 3, 1)
    assert wv.is_contiguous()
    L = _lib.check_device(wv)
    U = torch.empty((_points(tile), K, C), dtype=torch.float32, device=w.device) if want_u else None
    Uf = torch.empty((_points(tile), C, K), dtype=torch.float32, device=w.device) if want_flip else None
    L.call("omni_wino_weights", _lib.ptr(wv), _lib.ptr(U), _lib.ptr(Uf), K, C, tile, _lib.stream_of(w))
    return U, Uf


WEIGHTS_MULTI_MAX = 48


def transform_weights_multi(items):
    """items: [(w (K,C,3,3) CL, want_u, want_flip, tile)] (<= 48) -> [(U or None, U' or None)], every filter in ONE launch"""
    import ctypes
    assert 0 < len(items) <= WEIGHTS_MULTI_MAX
    outs, gp, up, fp, Ks, Cs, Ts = [], [], [], [], [], [], []
    for w, want_u, want_flip, tile in items:
        K, C = w.shape[0], w.shape[1]
        wv = w.permute(0, 2, 3, 1)
        assert wv.is_contiguous() and (want_u or want_flip)
        U = torch.empty((_points(tile), K, C), dtype=torch.float32, device=w.device) if want_u else None
        Uf = torch.empty((_points(tile), C, K), dtype=torch.float32, device=w.device) if want_flip else None
        outs.append((U, Uf))
        gp.append(wv.data_ptr()); up.append(_lib.ptr(U)); fp.append(_lib.ptr(Uf)); Ks.append(K); Cs.append(C); Ts.append(tile)
    L = _lib.check_device(items[0][0].permute(0, 2, 3, 1))
    n = len(items)
    P = ctypes.c_void_p
    arr = lambda vals: ctypes.cast((P * n)(*vals), P)                     # noqa: E731
    ints = lambda vals: ctypes.cast((ctypes.c_int * n)(*vals), P)         # noqa: E731
    L.call("omni_wino_weights_multi", arr(gp), arr(up), arr(fp), ints(Ks), ints(Cs), ints(Ts), n, _lib.stream_of(items[0][0]))
    return outs


# OMNI_GEMM_SPLIT=3 | 6: the point GEMMs of the Winograd forward / data-gradient path on the bf16 matrix cores from an error-free
# operand split (an experiment with its own bench object and its own error report; 0 = the product's fp32-MFMA path)
GEMM_SPLIT = int(_os.environ.get("OMNI_GEMM_SPLIT", "0"))
# only problems with at least this many rows (Winograd tiles) take the split form: on the 64 x 64 and 128 x 128 maps the 6-term kernel is
# 1.17-1.5x faster than the fp32-MFMA one, on the small maps it is not (profiles/r06_bench_gemm_split.log); step: 10.62-10.64 ms with
# 1024, 10.73-10.74 with 0, 10.80-10.83 without the split (profiles/r06_ab_gemm_split_minm.log)
GEMM_SPLIT_MIN_M = int(_os.environ.get("OMNI_GEMM_SPLIT_MIN_M", "1024"))


def gemm_batched_split(V, U, terms):
    """V (B,M,C), U (B,K,C) -> (B,M,K) through the bf16 split with `terms` in (3, 6) products per fp32 product"""
    B, M, C = V.shape
    K = U.shape[1]
    L = _lib.check_device(V, U)
    out = torch.empty((B, M, K), dtype=torch.float32, device=V.device)
    L.call("omni_gemm_batched_split", _lib.ptr(V), _lib.ptr(U), _lib.ptr(out), B, M, K, C, int(terms), _lib.stream_of(V))
    return out


def gemm_batched(V, U, algo=0, workgroups=0):
    """V (B,M,C), U (B,K,C) -> (B,M,K).  algo / workgroups: see omni_gemm_batched_fwd_algo (0 = the launcher's choice)."""
    B, M, C = V.shape
    K = U.shape[1]
    L = _lib.check_device(V, U)
    out = torch.empty((B, M, K), dtype=torch.float32, device=V.device)
    if GEMM_SPLIT in (3, 6) and C % 32 == 0 and algo == 0 and M >= GEMM_SPLIT_MIN_M:      # OPT-IN experiment (csrc/gemm_split.hip): never the default, never the measured line
        L.call("omni_gemm_batched_split", _lib.ptr(V), _lib.ptr(U), _lib.ptr(out), B, M, K, C, GEMM_SPLIT, _lib.stream_of(V))
        return out
    L.call("omni_gemm_batched_fwd_algo", _lib.ptr(V), _lib.ptr(U), _lib.ptr(out), B, M, C, K, algo, workgroups, _lib.stream_of(V))
    return out


def gemm_batched_wgrad(V, dM, algo=0):
    """V (B,M,C), dM (B,M,K) -> dU (B,K,C) = dM^T V.  algo: see omni_gemm_batched_wgrad_algo (0 = the launcher's choice)."""
    B, M, C = V.shape
    K = dM.shape[2]
    L = _lib.check_device(V, dM)
    dU = torch.empty((B, K, C), dtype=torch.float32, device=V.device)
    from . import detmode as _det
    if not _det.on():
        L.call("omni_gemm_batched_wgrad_algo", _lib.ptr(V), _lib.ptr(dM), _lib.ptr(dU), B, M, C, K, algo, _lib.stream_of(V))
        return dU
    plan, addr = _det.new_plan()
    L.call("omni_gemm_batched_wgrad_det", _lib.ptr(V), _lib.ptr(dM), _lib.ptr(dU), B, M, C, K, algo, None, 0, None, 0, addr, _lib.stream_of(V))
    ws, wsf, ctr, nctr = _det.workspace(V, plan)
    L.call("omni_gemm_batched_wgrad_det", _lib.ptr(V), _lib.ptr(dM), _lib.ptr(dU), B, M, C, K, int(plan[0]), _lib.ptr(ws), wsf, _lib.ptr(ctr), nctr,
           None, _lib.stream_of(V))
    return dU


WGRAD_MULTI_MAX = max(1, min(16, int(_os.environ.get("OMNI_WGRAD_MULTI_MAX", "16"))))      # problems per launch (A/B knob; the kernel takes <= 16)
WGRAD_MULTI = _os.environ.get("OMNI_WGRAD_MULTI", "1") != "0"
_deferred = None        # [(V, dM, dU, accum_into)] while a batched_wgrads() context is open


def gemm_batched_wgrad_multi(problems):
    """problems: [(V (B,M,C), dM (B,M,K))], <= 16, any mix of shapes -> [dU (B,K,C)], all in ONE launch (omni_gemm_batched_wgrad_multi);
    bit-identical to gemm_batched_wgrad on each"""
    import ctypes
    from . import detmode as _det
    n = len(problems)
    assert 0 < n <= WGRAD_MULTI_MAX
    L = _lib.check_device(*[t for pr in problems for t in pr])
    outs = [torch.empty((V.shape[0], dM.shape[2], V.shape[2]), dtype=torch.float32, device=V.device) for V, dM in problems]
    P = ctypes.c_void_p
    arr = lambda vals: ctypes.cast((P * n)(*vals), P)                     # noqa: E731
    ints = lambda vals: ctypes.cast((ctypes.c_int * n)(*vals), P)         # noqa: E731
    head = (arr([V.data_ptr() for V, _ in problems]), arr([dM.data_ptr() for _, dM in problems]), arr([o.data_ptr() for o in outs]),
            ints([V.shape[0] for V, _ in problems]), ints([V.shape[1] for V, _ in problems]), ints([V.shape[2] for V, _ in problems]),
            ints([dM.shape[2] for _, dM in problems]), n)
    V0 = problems[0][0]
    if _os.environ.get("OMNI_WGRAD_MULTI_LOG") == "1":       # (tools: which problems a launch holds, for the in-step roofline figures)
        gf = sum(2.0 * V.shape[0] * V.shape[1] * V.shape[2] * dM.shape[2] for V, dM in problems) / 1e9
        print(f"gemm_tn_multi: {n} problems, {gf:.2f} GFLOP: " + " ".join(f"{V.shape[0]}x[{V.shape[1]}x{dM.shape[2]}x{V.shape[2]}]" for V, dM in problems),
              flush=True)
    if not _det.on():
        L.call("omni_gemm_batched_wgrad_multi", *head, None, 0, None, 0, None, _lib.stream_of(V0))
        return outs
    plan, addr = _det.new_plan()
    L.call("omni_gemm_batched_wgrad_multi", *head, None, 0, None, 0, addr, _lib.stream_of(V0))
    ws, wsf, ctr, nctr = _det.workspace(V0, plan)
    L.call("omni_gemm_batched_wgrad_multi", *head, _lib.ptr(ws), wsf, _lib.ptr(ctr), nctr, None, _lib.stream_of(V0))
    return outs


def _multi_fits(V, dM):
    return V.shape[1] > 0 and V.shape[1] * V.shape[2] * 4 < (1 << 31) and dM.shape[1] * dM.shape[2] * 4 < (1 << 31)


class batched_wgrads:
    """Context of the weight-gradient stream: the Winograd-domain weight gradients issued inside through wgrad_into() are
    collected, their GEMMs leave in launches of <= 16 problems when the context closes (csrc/conv_gemm.hip gemm_tn_multi_kernel),
    then the transforms back add into the gradient views in the order of issue."""

    def __enter__(self):
        global _deferred
        self.prev, _deferred = _deferred, []
        from . import conv as _conv
        self.prev_direct = _conv.open_direct_batch()       # round 6: the stage's DIRECT weight gradients leave in one launch too
        return self

    def __exit__(self, et, ev, tb):
        global _deferred
        items, _deferred = _deferred, self.prev
        from . import conv as _conv
        _conv.close_direct_batch(self.prev_direct, launch=et is None)
        if et is None:
            for i in range(0, len(items), WGRAD_MULTI_MAX):
                part = items[i:i + WGRAD_MULTI_MAX]
                if len(part) == 1:
                    dUs = [gemm_batched_wgrad(part[0][0], part[0][1])]
                else:
                    dUs = gemm_batched_wgrad_multi([(V, dM) for V, dM, _ in part])
                transform_dweights_multi([(dU, into) for dU, (_, _, into) in zip(dUs, part)])
        return False


def transform_dweights_multi(items):
    """items: [(dU (P,K,C), accum_into (K,C,3,3) CL gradient view)], <= 16: accum_into += the transform back of dU, all in one launch;
    views named twice receive their sources in list order"""
    import ctypes
    n = len(items)
    assert 0 < n <= WGRAD_MULTI_MAX
    if n == 1:
        return transform_dweights(*items[0])
    L = _lib.check_device(*[dU for dU, _ in items])
    gvs = [into.permute(0, 2, 3, 1) for _, into in items]
    assert all(g.is_contiguous() for g in gvs)
    P = ctypes.c_void_p
    arr = lambda vals: ctypes.cast((P * n)(*vals), P)                     # noqa: E731
    ints = lambda vals: ctypes.cast((ctypes.c_int * n)(*vals), P)         # noqa: E731
    L.call("omni_wino_dweights_multi", arr([dU.data_ptr() for dU, _ in items]), arr([g.data_ptr() for g in gvs]),
           ints([dU.shape[1] for dU, _ in items]), ints([dU.shape[2] for dU, _ in items]),
           ints([2 if dU.shape[0] == 16 else 4 for dU, _ in items]), n, _lib.stream_of(items[0][0]))
    return None


def wgrad_into(V, dM, accum_into):
    """accum_into (KRSC-contiguous gradient view) += the weight gradient of a Winograd layer: V its transformed input, dM its
    transformed output gradient.  Inside batched_wgrads() the GEMM joins the context's launch."""
    if _deferred is None or not WGRAD_MULTI or not _multi_fits(V, dM):
        return transform_dweights(gemm_batched_wgrad(V, dM), accum_into)
    _deferred.append((V, dM, accum_into))
    return None


def transform_output(Mt, shape, bias=None, relu=False, carry=None):
    """Mt (P,T,K) -> y (N,K,H,W) CL; shape = (N, H, W); the tile size follows from P.
    carry (data gradients): gradient fan-in, a logical (N,K,H,W) tensor in NHWC memory with any pixel pitch, added to the result"""
    tile = 2 if Mt.shape[0] == 16 else 4
    N, H, W = shape
    K = Mt.shape[2]
    L = _lib.check_device(Mt, bias)
    y = torch.empty((N, H, W, K), dtype=torch.float32, device=Mt.device)
    if carry is not None:
        assert bias is None and not relu and tuple(carry.shape) == (N, K, H, W)
        L.call("omni_wino_out_carry", _lib.ptr(Mt), carry.data_ptr(), carry.stride(3), _lib.ptr(y), N, H, W, K, tile, _lib.stream_of(Mt))
        return y.permute(0, 3, 1, 2)
    L.call("omni_wino_out", _lib.ptr(Mt), _lib.ptr(bias), _lib.ptr(y), N, H, W, K, int(relu), tile, _lib.stream_of(Mt))
    return y.permute(0, 3, 1, 2)


def transform_output_stats(Mt, shape):
    """transform_output (no bias / ReLU) + BatchNorm partial statistics -> (y, partial (nblk, 2K) or None)"""
    from .conv import STATS_ROWS, _nblk_cell, _stats_buf
    tile = 2 if Mt.shape[0] == 16 else 4
    N, H, W = shape
    K = Mt.shape[2]
    L = _lib.check_device(Mt)
    y = torch.empty((N, H, W, K), dtype=torch.float32, device=Mt.device)
    stats = _stats_buf(K, Mt.device)
    cell, addr = _nblk_cell()
    L.call("omni_wino_out_stats", _lib.ptr(Mt), _lib.ptr(y), N, H, W, K, tile, _lib.ptr(stats), STATS_ROWS, addr, _lib.stream_of(Mt))
    return y.permute(0, 3, 1, 2), (stats[:cell.value] if cell.value > 0 else None)


def transform_output_bn_bwd(Mt, shape, bn_x, mean_rstd, scale_shift):
    """data-gradient transform_output that also emits the backward partial statistics of the BatchNorm whose output gradient it
    writes (bn_x: that layer's input, (N,K,H,W) CL; scale_shift None = no ReLU) -> (dy, partial (nblk, 2K) or None)"""
    from .conv import STATS_ROWS, _nblk_cell, _stats_buf
    tile = 2 if Mt.shape[0] == 16 else 4
    N, H, W = shape
    K = Mt.shape[2]
    xv = bn_x.permute(0, 2, 3, 1)
    assert xv.is_contiguous() and tuple(xv.shape) == (N, H, W, K)
    L = _lib.check_device(Mt, xv, mean_rstd, scale_shift)
    y = torch.empty((N, H, W, K), dtype=torch.float32, device=Mt.device)
    stats = _stats_buf(K, Mt.device)
    cell, addr = _nblk_cell()
    L.call("omni_wino_out_bn_bwd_stats", _lib.ptr(Mt), _lib.ptr(y), N, H, W, K, tile, _lib.ptr(xv), _lib.ptr(mean_rstd), _lib.ptr(scale_shift),
           _lib.ptr(stats), STATS_ROWS, addr, _lib.stream_of(Mt))
    return y.permute(0, 3, 1, 2), (stats[:cell.value] if cell.value > 0 else None)


def transform_dy(dy, tile=2):
    """dy (N,K,H,W) CL -> dM (P,T,K)"""
    dv = _nhwc(dy)
    N, H, W, K = dv.shape
    L = _lib.check_device(dv)
    dM = torch.empty((_points(tile), N * (H // tile) * (W // tile), K), dtype=torch.float32, device=dy.device)
    L.call("omni_wino_dy", _lib.ptr(dv), _lib.ptr(dM), N, H, W, K, tile, _lib.stream_of(dy))
    return dM


# one-pass kernel writes 2 x P plane-strided streams per thread; from `_DY_SPLIT_PLANE_BYTES` per plane on, two passes (P streams
# each, dy read twice) are used instead (A/B knob, 0 = never)
_DY_SPLIT_PLANE_BYTES = int(_os.environ.get("OMNI_WINO_DY_SPLIT", "0"))


def transform_dy_both(dy, tile=2):
    """dy (N,K,H,W) CL -> (dM (P,T,K), V_dy (P,T,K)): the weight-gradient and data-gradient transforms of dy in one pass"""
    dv = _nhwc(dy)
    N, H, W, K = dv.shape
    if _DY_SPLIT_PLANE_BYTES and N * (H // tile) * (W // tile) * K * 4 >= _DY_SPLIT_PLANE_BYTES:
        return transform_dy(dy, tile), transform_input(dy, tile)
    L = _lib.check_device(dv)
    shape = (_points(tile), N * (H // tile) * (W // tile), K)
    dM = torch.empty(shape, dtype=torch.float32, device=dy.device)
    Vd = torch.empty(shape, dtype=torch.float32, device=dy.device)
    L.call("omni_wino_dy_in", _lib.ptr(dv), _lib.ptr(dM), _lib.ptr(Vd), N, H, W, K, tile, _lib.stream_of(dy))
    return dM, Vd


def transform_dweights(dU, accum_into=None):
    """dU (P,K,C) -> dw (K,C,3,3) CL; accum_into: KRSC-contiguous gradient view to ADD into (returns None)."""
    P, K, C = dU.shape
    tile = 2 if P == 16 else 4
    L = _lib.check_device(dU)
    if accum_into is not None:
        gv = accum_into.permute(0, 2, 3, 1)
        assert gv.is_contiguous()
        L.call("omni_wino_dweights", _lib.ptr(dU), _lib.ptr(gv), K, C, 1, tile, _lib.stream_of(dU))
        return None
    dw = torch.empty((K, 3, 3, C), dtype=torch.float32, device=dU.device)
    L.call("omni_wino_dweights", _lib.ptr(dU), _lib.ptr(dw), K, C, 0, tile, _lib.stream_of(dU))
    return dw.permute(0, 3, 1, 2)


def conv3x3_fwd(x, w, bias=None, relu=False, U=None, tile=2, want_stats=False, in_affine=None, in_relu=False):
    """-> (y, V): V is kept by the caller for the weight gradient.  U: precomputed transform_weights(w, tile=tile)[0].
    want_stats (no bias, no ReLU): -> (y, V, BatchNorm partial statistics or None).
    in_affine / in_relu: the convolution of relu?(x * scale + shift) (see transform_input)"""
    N, _, H, W = x.shape
    V = transform_input(x, tile, affine=in_affine, relu=in_relu)
    if U is None:
        U = transform_weights(w, tile=tile)[0]
    Mt = gemm_batched(V, U)
    if want_stats:
        y, parts = transform_output_stats(Mt, (N, H, W))
        return y, V, parts
    return transform_output(Mt, (N, H, W), bias, relu), V


def conv3x3_dgrad(dy, w, U_flip=None, tile=2, carry=None):
    """dx = the same Winograd convolution applied to dy with the rotated / transposed filter (+ carry: see transform_output)."""
    N, _, H, W = dy.shape
    if U_flip is None:
        U_flip = transform_weights(w, want_u=False, want_flip=True, tile=tile)[1]
    Mt = gemm_batched(transform_input(dy, tile), U_flip)
    return transform_output(Mt, (N, H, W), carry=carry)


def conv3x3_backward(V, dy, w, U_flip, accum_into=None, side_run=None, bn_below=None, carry=None):
    """data gradient + weight gradient through Winograd with ONE pass over dy -> (dx, dw or None when accumulated).
    side_run(fn, keepalive): runs the weight-gradient half (batched GEMM + transform back, accumulated in place) on the
    weight-gradient stream (functional._side_run)."""
    N, _, H, W = dy.shape
    tile = 2 if V.shape[0] == 16 else 4
    dM, Vd = transform_dy_both(dy, tile)
    if side_run is not None and accum_into is not None:
        dw = side_run(lambda: wgrad_into(V, dM, accum_into), (V, dM))
    else:
        dw = None
    if U_flip is None:
        U_flip = transform_weights(w, want_u=False, want_flip=True, tile=tile)[1]
    if carry is not None:       # gradient fan-in of the input tensor, read by the output transform (see transform_output)
        dx = transform_output(gemm_batched(Vd, U_flip), (N, H, W), carry=carry)
    elif bn_below is not None and tile == 4:
        # the input of this convolution is the output of a BatchNorm(+ReLU): dx is that layer's dy, and the transform that writes
        # it also leaves the partial sums the BatchNorm backward starts with (functional._BatchNorm.backward picks them up)
        dx, parts = transform_output_bn_bwd(gemm_batched(Vd, U_flip), (N, H, W), *bn_below[:3])
        if parts is not None:
            dx._omni_bn_bwd_parts = (parts, bn_below[1])
    else:
        dx = transform_output(gemm_batched(Vd, U_flip), (N, H, W))
    if side_run is None or accum_into is None:
        dw = transform_dweights(gemm_batched_wgrad(V, dM), accum_into)
    return dx, dw


def conv3x3_wgrad(V, dy, accum_into=None):
    dM = transform_dy(dy, 2 if V.shape[0] == 16 else 4)
    if accum_into is not None:
        return wgrad_into(V, dM, accum_into)
    return transform_dweights(gemm_batched_wgrad(V, dM), None)
