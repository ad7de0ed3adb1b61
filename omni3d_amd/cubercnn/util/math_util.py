"""The five geometry helpers of the reference's util/math_util.py that sit on the hot path
(SURVEY.md 2.1 #9).  In the training / inference path they are fused into csrc/cube_head.hip; these
entry points expose them with the reference's names and argument meaning."""
import torch

from ...kernels import det


def get_cuboid_verts_faces(box3d=None, R=None):
    """math_util.py:116-219 -> (verts (n,8,3), faces (n,12,3))."""
    if box3d is None:
        box3d = [0, 0, 0, 1, 1, 1]
    box3d = torch.as_tensor(box3d, dtype=torch.float32)
    squeeze = box3d.dim() == 1
    if squeeze:
        box3d = box3d.unsqueeze(0)
    n = len(box3d)
    R = torch.eye(3).repeat(n, 1, 1) if R is None else torch.as_tensor(R, dtype=torch.float32).reshape(n, 3, 3)
    dev = box3d.device if box3d.is_cuda else torch.device("cuda")
    verts = det.cuboid_corners(box3d.to(dev), R.to(dev).reshape(n, 9)).to(box3d.device)
    faces = torch.tensor([[0, 1, 2], [2, 3, 0], [1, 5, 6], [6, 2, 1], [4, 0, 3], [3, 7, 4], [5, 4, 7], [7, 6, 5], [4, 5, 1],
                          [1, 0, 4], [3, 2, 6], [6, 7, 3]]).float().unsqueeze(0).repeat([n, 1, 1]).to(verts.device)
    if squeeze:
        verts, faces = verts.squeeze(), faces.squeeze()
    return verts, faces


def compute_virtual_scale_from_focal_spaces(f, H, f0, H0):
    """math_util.py:581-592"""
    return (H0 * f) / (f0 * H)


def scaled_sigmoid(vals, min=0.0, max=1.0):
    """math_util.py:969-978"""
    return min + (max - min) * torch.sigmoid(vals)
