"""Independent float64 IoU3D oracle (test infrastructure): exact convex intersection volume via
half-space intersection (scipy/Qhull).  Valid for generic (non face-coplanar) pairs."""
import numpy as np
from scipy.optimize import linprog
from scipy.spatial import ConvexHull, HalfspaceIntersection

_PLANES = [[0, 1, 2, 3], [3, 2, 6, 7], [0, 1, 5, 4], [0, 3, 7, 4], [1, 2, 6, 5], [4, 5, 6, 7]]


def _halfspaces(box):
    box = np.asarray(box, dtype=np.float64)
    ctr = box.mean(0)
    hs = []
    for p in _PLANES:
        v = box[p]
        n = np.cross(v[1] - v[0], v[2] - v[0])
        n /= np.linalg.norm(n)
        if np.dot(ctr - v[0], n) > 0:
            n = -n  # outward normal
        hs.append(np.concatenate([n, [-np.dot(n, v.mean(0))]]))  # n.x + b <= 0
    return np.array(hs)


def box_volume(box):
    return ConvexHull(np.asarray(box, dtype=np.float64)).volume


def intersection_volume(b1, b2):
    hs = np.vstack([_halfspaces(b1), _halfspaces(b2)])
    # Chebyshev centre: max r s.t. n.x + r <= -b
    A = np.hstack([hs[:, :3], np.ones((len(hs), 1))])
    res = linprog(c=[0, 0, 0, -1], A_ub=A, b_ub=-hs[:, 3], bounds=[(None, None)] * 3 + [(0, None)])
    if not res.success or res.x[3] < 1e-9:
        return 0.0
    hi = HalfspaceIntersection(hs, res.x[:3])
    return ConvexHull(hi.intersections).volume


def iou3d(b1, b2):
    v = intersection_volume(b1, b2)
    v1, v2 = box_volume(b1), box_volume(b2)
    return v, v / (v1 + v2 - v)
