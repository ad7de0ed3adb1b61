"""Pins the IoU3D oracle (oracle/iou_box3d_oracle.c): analytic known answers and an independent
float64 exact-geometry oracle (scipy).  The reference holds no golden vectors for this path
 -- these are self-derived."""
import ctypes

import numpy as np
import pytest

from omni3d_amd import boxgen
import exact_iou3d

P = ctypes.c_void_p


def oracle_iou(lib, b1, b2):
    b1 = np.ascontiguousarray(b1, np.float32)
    b2 = np.ascontiguousarray(b2, np.float32)
    N, M = len(b1), len(b2)
    vol = np.zeros((N, M), np.float32)
    iou = np.zeros((N, M), np.float32)
    lib.iou_box3d_oracle(b1.ctypes.data_as(P), N, b2.ctypes.data_as(P), M, vol.ctypes.data_as(P), iou.ctypes.data_as(P))
    return vol, iou


def oracle_overlap(lib, dt, gt, eps_c=1e-4, eps_n=1e-8):
    dt = np.ascontiguousarray(dt, np.float32)
    gt = np.ascontiguousarray(gt, np.float32)
    iou = np.zeros((len(dt), len(gt)), np.float32)
    lib.box3d_overlap_oracle(dt.ctypes.data_as(P), len(dt), gt.ctypes.data_as(P), len(gt), ctypes.c_float(eps_c),
                             ctypes.c_float(eps_n), iou.ctypes.data_as(P))
    return iou


UNIT = (boxgen.UNIT + 0.5).astype(np.float32)  # pytorch3d's documented unit box


def test_known_answers(oracle_lib):
    c = boxgen.UNIT.astype(np.float32)
    _, iou = oracle_iou(oracle_lib, UNIT[None], UNIT[None])
    assert abs(iou[0, 0] - 1.0) < 1e-6
    vol, iou = oracle_iou(oracle_lib, UNIT[None], (UNIT + np.float32([0.5, 0, 0]))[None])
    assert abs(vol[0, 0] - 0.5) < 1e-6 and abs(iou[0, 0] - 1 / 3) < 1e-6
    vol, iou = oracle_iou(oracle_lib, UNIT[None], (UNIT + np.float32([2, 0, 0]))[None])
    assert vol[0, 0] == 0 and iou[0, 0] == 0
    th = np.pi / 4
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]], np.float32)
    vol, iou = oracle_iou(oracle_lib, c[None], (c @ R.T)[None])
    v = 2 * (np.sqrt(2) - 1)
    assert abs(vol[0, 0] - v) < 1e-5 and abs(iou[0, 0] - v / (2 - v)) < 1e-5  # = sqrt(2)/2
    vol, iou = oracle_iou(oracle_lib, c[None], (c * 0.5)[None])
    assert abs(vol[0, 0] - 0.125) < 1e-6 and abs(iou[0, 0] - 0.125) < 1e-6  # nested


def test_against_exact_float64_on_generic_pairs(oracle_lib, rng):
    b1 = boxgen.random_boxes(rng, 12)
    b2 = boxgen.random_boxes(rng, 12)
    vol, iou = oracle_iou(oracle_lib, b1, b2)
    for i in range(len(b1)):
        for j in range(len(b2)):
            v, u = exact_iou3d.iou3d(b1[i], b2[j])
            assert abs(vol[i, j] - v) < 2e-4 * max(1.0, v), (i, j, vol[i, j], v)
            assert abs(iou[i, j] - u) < 1e-4, (i, j, iou[i, j], u)


def test_validity_masks(oracle_lib, rng):
    dt = boxgen.random_boxes(rng, 6)
    gt = dt.copy()
    bad_flat = boxgen.corners(np.zeros((1, 3)), np.array([[1.0, 0.0, 2.0]]), np.eye(3)[None])[0]
    dt[1] = bad_flat           # zero-area faces
    dt[4, 6] += 0.5            # skewed vertex -> not coplanar
    iou = oracle_overlap(oracle_lib, dt, gt)
    assert (iou[1] == 0).all() and (iou[4] == 0).all()
    for i in (0, 2, 3, 5):
        assert abs(iou[i, i] - 1) < 1e-5


def test_oracle_gives_exact_zero_for_sphere_separated_boxes(oracle_lib, rng):
    """The kernel's bounding-sphere screening (csrc/iou_box3d.hip spheres_disjoint) writes vol = iou = 0 without running the
    clipping passes.  That is only a shortcut if the ALGORITHM, epsilon rules included, returns exactly 0 for such pairs: checked
    here on the oracle for 60k random sphere-separated pairs, among them near misses (gap of 1e-3 of the radii), boxes sharing a
    face plane at a distance (the coplanarity rule keeps such triangles "as is" for that one plane), and axis-aligned rows."""
    import ctypes
    n = 60_000
    b1 = boxgen.random_boxes(rng, n)
    b2 = boxgen.random_boxes(rng, n)
    unit = (boxgen.UNIT + 0.5).astype(np.float32)
    # rows of axis-aligned unit cubes in one plane: every pair shares two infinite face planes
    b1[:2000] = unit
    b2[:2000] = unit + np.stack([rng.uniform(1.8, 30, 2000), np.zeros(2000), np.zeros(2000)], 1)[:, None, :].astype(np.float32)

    def sphere(b):
        c = b.mean(1)
        return c, np.sqrt(((b - c[:, None]) ** 2).sum(2).max(1))
    c1, r1 = sphere(b1)
    c2, r2 = sphere(b2)
    # push box2 away along the centre line until the spheres are disjoint by the kernel's margin, a third of them only just
    d = c2 - c1
    dist = np.linalg.norm(d, axis=1)
    u = np.where(dist[:, None] > 1e-6, d / np.maximum(dist, 1e-6)[:, None], np.array([[1.0, 0, 0]]))
    gap = np.where(rng.uniform(size=n) < 0.33, 1e-3, rng.uniform(0.01, 5.0, n)) * (r1 + r2)
    want = (r1 + r2) * 1.0001 + 1e-4 + gap
    move = np.maximum(want - dist, 0.0)
    b2 = (b2 + (u * move[:, None])[:, None, :]).astype(np.float32)
    c2, r2 = sphere(b2)
    sep = np.linalg.norm(c2 - c1, axis=1) > (r1 + r2) * 1.0001 + 1e-4
    assert sep.mean() > 0.99
    out = np.full(n, -1.0, np.float32)
    P = ctypes.c_void_p
    a, b = np.ascontiguousarray(b1), np.ascontiguousarray(b2)
    oracle_lib.iou_box3d_pairs_oracle(a.ctypes.data_as(P), b.ctypes.data_as(P), n, out.ctypes.data_as(P))
    assert (out[sep] == 0.0).all(), (int((out[sep] != 0).sum()), float(np.abs(out[sep]).max()))
