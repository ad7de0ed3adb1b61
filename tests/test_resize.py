"""Device-side input pipeline: csrc/resize.hip against PIL itself -- the reference's resize IS
`Image.resize((w, h), BILINEAR)` (detectron2 ResizeTransform.apply_image, reached from dataset_mapper.py:25-27), and Pillow
is installed, so the oracle here is the real third-party implementation: bit-exact."""
import numpy as np
import pytest
import torch

CASES = [  # H, W -> out_h, out_w, flip
    (37, 53, 64, 91, False),     # upscale both
    (120, 160, 48, 64, False),   # 2.5x antialiased downscale
    (97, 131, 97, 60, True),     # horizontal only + flip
    (64, 48, 50, 48, False),     # vertical only
    (33, 47, 33, 47, True),      # flip only
    (200, 150, 341, 256, True),  # ResizeShortestEdge-style: short edge 150 -> 256
]


def _pil(img_chw, out_h, out_w, flip):
    from PIL import Image
    hwc = np.ascontiguousarray(img_chw.transpose(1, 2, 0))
    out = np.asarray(Image.fromarray(hwc).resize((out_w, out_h), Image.BILINEAR))
    if flip:
        out = np.flip(out, axis=1)
    return np.ascontiguousarray(out.transpose(2, 0, 1))


def _run(dev):
    from omni3d_amd.kernels import resize
    rs = np.random.RandomState(0)
    for H, W, oh, ow, flip in CASES:
        img = rs.randint(0, 256, size=(3, H, W)).astype(np.uint8)
        img[:, : H // 3] = 255                      # saturated block: exercises the 8-bit clipping
        img[:, :, : W // 4] = 0
        want = _pil(img, oh, ow, flip)
        got = resize.resize_bilinear_u8(torch.from_numpy(img).to(dev), oh, ow, flip).cpu().numpy()
        assert got.shape == want.shape
        assert np.array_equal(got, want), (H, W, oh, ow, flip, int(np.abs(got.astype(int) - want.astype(int)).max()))


def test_coefficients_reproduce_pil_on_the_host():
    """numpy evaluation of the two passes with the launcher's coefficient tables == PIL (pins the table construction)"""
    from omni3d_amd.kernels.resize import pil_bilinear_coeffs
    rs = np.random.RandomState(1)
    img = rs.randint(0, 256, size=(3, 45, 70)).astype(np.uint8)
    oh, ow = 23, 111
    bh, kh, _ = pil_bilinear_coeffs(70, ow)
    bv, kv, _ = pil_bilinear_coeffs(45, oh)
    tmp = np.zeros((3, 45, ow), np.uint8)
    for xx in range(ow):
        x0, n = bh[xx]
        ss = (1 << 21) + (img[:, :, x0:x0 + n].astype(np.int64) * kh[xx, :n]).sum(-1)
        tmp[:, :, xx] = np.clip(ss >> 22, 0, 255)
    out = np.zeros((3, oh, ow), np.uint8)
    for yy in range(oh):
        y0, n = bv[yy]
        ss = (1 << 21) + (tmp[:, y0:y0 + n].astype(np.int64) * kv[yy, :n, None]).sum(1)
        out[:, yy] = np.clip(ss >> 22, 0, 255)
    assert np.array_equal(out, _pil(img, oh, ow, False))


def test_resize_matches_pil_emulated(emu_lib):
    _run("cpu")


@pytest.mark.gpu
def test_resize_matches_pil_gpu(hip_lib):
    _run("cuda")
