"""Synthetic oriented 3D boxes in the reference's corner order
(/root/reference/cubercnn/util/math_util.py:151-181 == pytorch3d order quoted at
/root/reference/cubercnn/evaluation/omni3d_evaluation.py:117-142)."""
import numpy as np

# x = +-l/2 on verts {1,2,5,6}/{0,3,4,7}; y = +-h/2 on {2,3,6,7}/{0,1,4,5}; z = +-w/2 on {4..7}/{0..3}
UNIT = np.array([[-.5, -.5, -.5], [.5, -.5, -.5], [.5, .5, -.5], [-.5, .5, -.5],
                 [-.5, -.5, .5], [.5, -.5, .5], [.5, .5, .5], [-.5, .5, .5]], dtype=np.float64)


def rand_rot(rng, n):
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    return np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                     2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                     2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1).reshape(n, 3, 3)


def corners(center, dims_lhw, R):
    """center (n,3), dims (n,3) as (l along x, h along y, w along z), R (n,3,3) -> (n,8,3) float32"""
    v = UNIT[None] * dims_lhw[:, None, :]
    return (np.einsum('nij,nkj->nki', R, v) + center[:, None, :]).astype(np.float32)


def random_boxes(rng, n, spread=2.0, dmin=0.5, dmax=3.0):
    c = rng.uniform(-spread, spread, size=(n, 3))
    d = rng.uniform(dmin, dmax, size=(n, 3))
    return corners(c, d, rand_rot(rng, n))


def omni3d_like_pairs(rng, n_pairs, overlap_frac=0.5, degenerate_frac=0.01):
    """centres U[-5,5]xU[-2,2]xU[2,40], dims U[.2,5], random SO(3);
    a fraction of gt boxes are jittered copies of their dt box (forced overlap); a small
    fraction of dt boxes are degenerate (zero dimension or a skewed vertex)."""
    c = np.stack([rng.uniform(-5, 5, n_pairs), rng.uniform(-2, 2, n_pairs), rng.uniform(2, 40, n_pairs)], 1)
    d = rng.uniform(0.2, 5, size=(n_pairs, 3))
    R = rand_rot(rng, n_pairs)
    dt = corners(c, d, R)
    c2 = np.stack([rng.uniform(-5, 5, n_pairs), rng.uniform(-2, 2, n_pairs), rng.uniform(2, 40, n_pairs)], 1)
    d2 = rng.uniform(0.2, 5, size=(n_pairs, 3))
    R2 = rand_rot(rng, n_pairs)
    ov = rng.uniform(size=n_pairs) < overlap_frac
    c2[ov] = c[ov] + rng.normal(scale=0.3, size=(ov.sum(), 3)) * d[ov]
    d2[ov] = d[ov] * rng.uniform(0.7, 1.3, size=(ov.sum(), 3))
    # perturb the rotation of the overlapping copies a little
    Rj = rand_rot(rng, n_pairs)
    t = 0.15
    Rmix = R + t * (Rj - R)
    u, _, vt = np.linalg.svd(Rmix)
    Rmix = u @ vt
    Rmix[np.linalg.det(Rmix) < 0] *= -1
    R2[ov] = Rmix[ov]
    gt = corners(c2, d2, R2)
    deg = rng.uniform(size=n_pairs) < degenerate_frac
    idx = np.nonzero(deg)[0]
    for k, i in enumerate(idx):
        if k % 2 == 0:
            dd = d[i].copy(); dd[k % 3] = 0.0
            dt[i] = corners(c[i:i + 1], dd[None], R[i:i + 1])[0]
        else:
            dt[i, 6] += np.float32(0.5)  # skewed vertex -> non-coplanar faces
    return dt, gt, deg
