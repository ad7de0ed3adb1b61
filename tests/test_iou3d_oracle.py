"""Pins the IoU3D oracle (oracle/iou_box3d_oracle.c): analytic known answers and an independent
float64 exact-geometry oracle (scipy).  The reference holds no golden vectors for this path
 -- these are self-derived."""
import ctypes

import numpy as np
import pytest

import boxgen
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
