"""Launchers for csrc/bn_pool.hip.  All tensors are logical NCHW in channels_last memory
(physically NHWC), fp32, C % 4 == 0."""
import os

import torch

from .. import lib as _lib
from .conv import _nhwc


def _like_cl(shape_nhwc, ref):
    return torch.empty(shape_nhwc, dtype=torch.float32, device=ref.device)


# partial rows up to which the finalize is folded into the apply launch (omni_bn_fwd_algo / omni_bn_bwd_algo); 0 = the separate finalize
# launch of rounds 1-4.  Read once: an A/B knob for bench runs, the tests and the driver leave it unset.
FUSE_ROWS = int(os.environ.get("OMNI_BN_FUSE_ROWS", "512"))


def _pitched_out(out, N, H, W, C):
    """out: logical (N, C, H, W) view with channel stride 1 whose pixels are `ldy` floats apart (a channel slice of a wider NHWC tensor)
    -> (tensor to return, address, ldy)"""
    assert tuple(out.shape) == (N, C, H, W) and out.dtype == torch.float32 and out.stride(1) == 1, (out.shape, out.stride())
    ldy = out.stride(3)
    assert out.stride(2) == W * ldy and out.stride(0) == H * W * ldy and ldy >= C and ldy % 4 == 0 and out.data_ptr() % 16 == 0
    return out, out.data_ptr(), ldy


def bn_fwd(x, gamma, beta, running_mean, running_var, residual=None, relu=False, eps=1e-5, momentum=0.1, partials=None, out=None):
    """-> (y CL, mean_rstd (2C), scale_shift (2C)); running stats updated in place.  partials (nblk, 2C): per-block sums / sums
    of squares of x already produced by the kernel that wrote x (conv / Winograd / stem epilogue): skips the statistics pass.
    out: where y goes -- a channel slice of a wider NHWC tensor (the DLA Root's concatenated input) -- instead of a new tensor."""
    xv = _nhwc(x)
    rv = _nhwc(residual) if residual is not None else None
    N, H, W, C = xv.shape
    L = _lib.check_device(xv, rv, gamma, beta, running_mean, running_var)
    if out is not None:
        y, yaddr, ldy = _pitched_out(out, N, H, W, C)
    else:
        yv = _like_cl((N, H, W, C), x)
        y, yaddr, ldy = yv.permute(0, 3, 1, 2), _lib.ptr(yv), C
    mean_rstd = torch.empty(2 * C, dtype=torch.float32, device=x.device)
    scale_shift = torch.empty(2 * C, dtype=torch.float32, device=x.device)
    ws = None
    if partials is not None:
        assert partials.is_contiguous() and partials.shape[1] == 2 * C
    else:
        ws = torch.empty(2 * C * 258, dtype=torch.float64, device=x.device)
    L.call("omni_bn_fwd_algo", _lib.ptr(xv), _lib.ptr(partials), partials.shape[0] if partials is not None else 0, _lib.ptr(gamma),
           _lib.ptr(beta), _lib.ptr(rv), yaddr, ldy, _lib.ptr(running_mean), _lib.ptr(running_var), _lib.ptr(mean_rstd),
           _lib.ptr(scale_shift), _lib.ptr(ws), N * H * W, C, float(eps), float(momentum), int(relu), FUSE_ROWS, _lib.stream_of(x))
    return y, mean_rstd, scale_shift


def bn_finalize_fwd(x, gamma, beta, running_mean, running_var, partials, eps=1e-5, momentum=0.1):
    """the finalize half of bn_fwd alone (statistics from the producer's epilogue): -> (mean_rstd (2C), scale_shift (2C)); running
    statistics updated.  x only gives the geometry: the normalised tensor is applied on load by its consumer (wino.transform_input)."""
    xv = _nhwc(x)
    N, H, W, C = xv.shape
    assert partials is not None and partials.is_contiguous() and partials.shape[1] == 2 * C
    L = _lib.check_device(xv, gamma, beta, running_mean, running_var, partials)
    mean_rstd = torch.empty(2 * C, dtype=torch.float32, device=x.device)
    scale_shift = torch.empty(2 * C, dtype=torch.float32, device=x.device)
    L.call("omni_bn_finalize_fwd", _lib.ptr(partials), partials.shape[0], _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(running_mean),
           _lib.ptr(running_var), _lib.ptr(mean_rstd), _lib.ptr(scale_shift), N * H * W, C, float(eps), float(momentum), _lib.stream_of(x))
    return mean_rstd, scale_shift


def bn_apply(x, scale_shift, residual=None, relu=False):
    xv = _nhwc(x)
    rv = _nhwc(residual) if residual is not None else None
    N, H, W, C = xv.shape
    L = _lib.check_device(xv, rv, scale_shift)
    y = _like_cl((N, H, W, C), x)
    L.call("omni_bn_apply", _lib.ptr(xv), _lib.ptr(scale_shift), _lib.ptr(rv), _lib.ptr(y), N * H * W, C, int(relu),
           _lib.stream_of(x))
    return y.permute(0, 3, 1, 2)


def bn_bwd(x, dy, y, gamma, mean_rstd, relu=False, want_dres=False, accum_into=None, scale_shift=None, partials=None, res_carry=None):
    """-> (dx CL, dres CL or None, dgamma, dbeta).  accum_into = (dgamma_buf, dbeta_buf): the parameter
    gradients are added to those buffers instead (dgamma/dbeta returned as None).  scale_shift (2C, from bn_fwd) instead of y:
    the ReLU mask of a layer WITHOUT residual is recomputed from x (mode 2 of omni_bn_bwd), the output tensor is not read.
    partials (nblk, 2C): the reductions over dy already made by the kernel that produced dy (wino.transform_output_bn_bwd).
    res_carry: gradient fan-in of the residual tensor (logical NCHW, NHWC memory, any pixel pitch: functional._carry_pitch), added
    to dres where it is written.  dy may be such a pitched tensor too (a channel slice of the Root's concatenated gradient)."""
    xv = _nhwc(x)
    dy_pitched = not dy.is_contiguous(memory_format=torch.channels_last)
    dyv = None if dy_pitched else _nhwc(dy)
    mode = int(bool(relu))
    if relu and scale_shift is not None:
        assert y is None and scale_shift.is_contiguous() and scale_shift.numel() == 2 * xv.shape[3]
        yv, mode = scale_shift, 2
    else:
        yv = _nhwc(y) if relu else None
    N, H, W, C = xv.shape
    L = _lib.check_device(xv, dyv, yv, gamma, mean_rstd)
    dx = _like_cl((N, H, W, C), x)
    dres = _like_cl((N, H, W, C), x) if want_dres else None
    if accum_into is not None:
        dgamma, dbeta = accum_into
        assert dgamma.is_contiguous() and dbeta.is_contiguous()
    else:
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
    coef = torch.empty(3 * C, dtype=torch.float32, device=x.device)
    ws = None
    if partials is not None:
        assert partials.is_contiguous() and partials.shape[1] == 2 * C and res_carry is None and not dy_pitched
    else:
        ws = torch.empty(2 * C * 258, dtype=torch.float64, device=x.device)
    if res_carry is not None or dy_pitched:
        assert tuple(dy.shape) == tuple(x.shape) and dy.stride(1) == 1
        if res_carry is not None:
            assert want_dres and tuple(res_carry.shape) == tuple(x.shape)
    L.call("omni_bn_bwd_algo", _lib.ptr(xv), dy.data_ptr() if dy_pitched else _lib.ptr(dyv), dy.stride(3) if dy_pitched else C, _lib.ptr(yv),
           _lib.ptr(gamma), _lib.ptr(mean_rstd), _lib.ptr(partials), partials.shape[0] if partials is not None else 0, _lib.ptr(dx),
           _lib.ptr(dres), _lib.ptr(res_carry), res_carry.stride(3) if res_carry is not None else 0, _lib.ptr(dgamma), _lib.ptr(dbeta),
           _lib.ptr(ws), _lib.ptr(coef), N * H * W, C, mode, int(accum_into is not None), FUSE_ROWS, _lib.stream_of(x))
    if accum_into is not None:
        dgamma = dbeta = None
    return dx.permute(0, 3, 1, 2), (dres.permute(0, 3, 1, 2) if want_dres else None), dgamma, dbeta


def _simple(name, src, out_shape, dims, extra_in=()):
    L = _lib.check_device(src, *extra_in)
    out = torch.empty(out_shape, dtype=torch.float32, device=src.device)
    L.call(name, *[_lib.ptr(t) for t in (src, *extra_in)], _lib.ptr(out), *dims, _lib.stream_of(src))
    return out


def maxpool2_fwd(x):
    xv = _nhwc(x)
    N, H, W, C = xv.shape
    return _simple("omni_maxpool2_fwd", xv, (N, H // 2, W // 2, C), (N, H, W, C)).permute(0, 3, 1, 2)


def maxpool2_bwd(x, dy, carry=None):
    """carry: gradient fan-in of x (logical NCHW like x, NHWC memory, any pixel pitch), added to the routed gradient; dy may be
    pitched the same way (functional._carry_pitch)"""
    xv = _nhwc(x)
    N, H, W, C = xv.shape
    if carry is None and dy.is_contiguous(memory_format=torch.channels_last):
        return _simple("omni_maxpool2_bwd", xv, (N, H, W, C), (N, H, W, C), extra_in=(_nhwc(dy),)).permute(0, 3, 1, 2)
    assert (carry is None or tuple(carry.shape) == tuple(x.shape)) and tuple(dy.shape) == (N, C, H // 2, W // 2) and dy.stride(1) == 1
    L = _lib.check_device(xv)
    dx = torch.empty((N, H, W, C), dtype=torch.float32, device=x.device)
    L.call("omni_maxpool2_bwd_carry", _lib.ptr(xv), dy.data_ptr(), dy.stride(3), _lib.ptr(carry), carry.stride(3) if carry is not None else 0,
           _lib.ptr(dx), N, H, W, C, _lib.stream_of(x))
    return dx.permute(0, 3, 1, 2)


def avgpool2_fwd(x):
    xv = _nhwc(x)
    N, H, W, C = xv.shape
    return _simple("omni_avgpool2_fwd", xv, (N, H // 2, W // 2, C), (N, H, W, C)).permute(0, 3, 1, 2)


def avgpool2_bwd(dy, in_hw):
    dyv = _nhwc(dy)
    N, _, _, C = dyv.shape
    H, W = in_hw
    return _simple("omni_avgpool2_bwd", dyv, (N, H, W, C), (N, H, W, C)).permute(0, 3, 1, 2)


def subsample2_fwd(x):
    xv = _nhwc(x)
    N, H, W, C = xv.shape
    return _simple("omni_subsample2_fwd", xv, (N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C), (N, H, W, C)).permute(0, 3, 1, 2)


def subsample2_bwd(dy, in_hw, carry=None):
    """carry: gradient fan-in of x ((N,C,H,W) logical, NHWC memory, any pixel pitch): dx = carry + scattered dy in one pass"""
    dyv = _nhwc(dy)
    N, _, _, C = dyv.shape
    H, W = in_hw
    if carry is None:
        return _simple("omni_subsample2_bwd", dyv, (N, H, W, C), (N, H, W, C)).permute(0, 3, 1, 2)
    assert tuple(carry.shape) == (N, C, H, W)
    L = _lib.check_device(dyv)
    dx = torch.empty((N, H, W, C), dtype=torch.float32, device=dy.device)
    L.call("omni_subsample2_bwd_carry", _lib.ptr(dyv), carry.data_ptr(), carry.stride(3), _lib.ptr(dx), N, H, W, C, _lib.stream_of(dy))
    return dx.permute(0, 3, 1, 2)


def upsample2_add(lat, top):
    lv, tv = _nhwc(lat), _nhwc(top)
    N, H, W, C = lv.shape
    assert tv.shape == (N, H // 2, W // 2, C), (tv.shape, lv.shape)
    return _simple("omni_upsample2_add", lv, (N, H, W, C), (N, H, W, C), extra_in=(tv,)).permute(0, 3, 1, 2)


def upsample2_bwd(dout, carry=None):
    """carry: gradient fan-in of the top-down input ((N,C,H/2,W/2) logical, NHWC memory, any pixel pitch)"""
    dv = _nhwc(dout)
    N, H, W, C = dv.shape
    if carry is None:
        return _simple("omni_upsample2_bwd", dv, (N, H // 2, W // 2, C), (N, H, W, C)).permute(0, 3, 1, 2)
    assert tuple(carry.shape) == (N, C, H // 2, W // 2)
    L = _lib.check_device(dv)
    dtop = torch.empty((N, H // 2, W // 2, C), dtype=torch.float32, device=dout.device)
    L.call("omni_upsample2_bwd_carry", _lib.ptr(dv), carry.data_ptr(), carry.stride(3), _lib.ptr(dtop), N, H, W, C, _lib.stream_of(dout))
    return dtop.permute(0, 3, 1, 2)


def preprocess(images_u8, pixel_mean, pixel_std, size_divisibility=0, image_hw=None):
    """images (N,3,H,W) uint8 -> (N,4,PH,PW) CL fp32 normalised, zero padded (channel 3 = 0).
    image_hw (N,2) int32 on the device: valid size of every image inside its slot (the rest of the slot is padding)."""
    assert images_u8.dtype == torch.uint8 and images_u8.dim() == 4 and images_u8.shape[1] == 3
    images_u8 = images_u8.contiguous()
    N, _, H, W = images_u8.shape
    PH, PW = H, W
    if size_divisibility > 1:
        s = size_divisibility
        PH, PW = (H + s - 1) // s * s, (W + s - 1) // s * s
    L = _lib.check_device(images_u8)
    out = torch.empty((N, PH, PW, 4), dtype=torch.float32, device=images_u8.device)
    m, s = [float(v) for v in pixel_mean], [float(v) for v in pixel_std]
    if image_hw is not None:
        assert image_hw.dtype == torch.int32 and tuple(image_hw.shape) == (N, 2) and image_hw.is_contiguous()
        L.call("omni_preprocess_masked", _lib.ptr(images_u8), _lib.ptr(image_hw), _lib.ptr(out), N, H, W, PH, PW, m[0], m[1], m[2], s[0], s[1],
               s[2], _lib.stream_of(images_u8))
    else:
        L.call("omni_preprocess", _lib.ptr(images_u8), _lib.ptr(out), N, H, W, PH, PW, m[0], m[1], m[2], s[0], s[1], s[2],
               _lib.stream_of(images_u8))
    return out.permute(0, 3, 1, 2)


def preprocess_list(images, pixel_mean, pixel_std, size_divisibility=0, image_hw=None):
    """`preprocess(torch.stack(images), ...)` without the stacked copy: N <= 64 equal-size (3,H,W) uint8 device tensors, read through
    their own pointers (omni_preprocess_multi).  -> None when the list does not qualify (the caller stacks)."""
    import ctypes
    N = len(images)
    if not (0 < N <= 64):
        return None
    H, W = images[0].shape[-2:]
    for im in images:
        if im.dtype != torch.uint8 or im.dim() != 3 or tuple(im.shape) != (3, H, W) or not im.is_contiguous() or im.device != images[0].device:
            return None
    L = _lib.get()
    if not images[0].is_cuda and not L.emulated:
        return None
    PH, PW = H, W
    if size_divisibility > 1:
        s = size_divisibility
        PH, PW = (H + s - 1) // s * s, (W + s - 1) // s * s
    out = torch.empty((N, PH, PW, 4), dtype=torch.float32, device=images[0].device)
    m, s = [float(v) for v in pixel_mean], [float(v) for v in pixel_std]
    if image_hw is not None:
        assert image_hw.dtype == torch.int32 and tuple(image_hw.shape) == (N, 2) and image_hw.is_contiguous()
    ptrs = (ctypes.c_void_p * N)(*[im.data_ptr() for im in images])
    L.call("omni_preprocess_multi", ctypes.cast(ptrs, ctypes.c_void_p), _lib.ptr(image_hw), _lib.ptr(out), N, H, W, PH, PW, m[0], m[1], m[2],
           s[0], s[1], s[2], _lib.stream_of(images[0]))
    return out.permute(0, 3, 1, 2)


def relu_bwd(dy, y):
    """dz = dy * (y > 0); same memory layout in and out."""
    L = _lib.check_device(dy, y)
    dz = torch.empty_like(dy)
    L.call("omni_relu_bwd", _lib.ptr(dy), _lib.ptr(y), _lib.ptr(dz), dy.numel(), _lib.stream_of(dy))
    return dz


def bias_grad(dy2d, accum_into=None):
    """dy (P, C) contiguous -> (C,) column sums (or added to accum_into, returning None)."""
    P, C = dy2d.shape
    L = _lib.check_device(dy2d)
    db = accum_into if accum_into is not None else torch.empty(C, dtype=torch.float32, device=dy2d.device)
    ws = torch.empty(2 * C * 258, dtype=torch.float64, device=dy2d.device)
    from . import detmode as _det
    acc = 0 if accum_into is None else (3 if _det.on() else 1)     # 3: add without atomics (fixed-order finalize)
    L.call("omni_bias_grad", _lib.ptr(dy2d), P, C, _lib.ptr(db), _lib.ptr(ws), acc, _lib.stream_of(dy2d))
    return None if accum_into is not None else db


def maxpool3s2_fwd(x):
    xv = _nhwc(x)
    N, H, W, C = xv.shape
    return _simple("omni_maxpool3s2_fwd", xv, (N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C), (N, H, W, C)).permute(0, 3, 1, 2)


def maxpool3s2_bwd(x, dy):
    xv, dyv = _nhwc(x), _nhwc(dy)
    N, H, W, C = xv.shape
    return _simple("omni_maxpool3s2_bwd", xv, (N, H, W, C), (N, H, W, C), extra_in=(dyv,)).permute(0, 3, 1, 2)
