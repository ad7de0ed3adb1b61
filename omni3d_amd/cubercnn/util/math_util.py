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


def _viewing_rotation(K, u, v):
    """M (n,3,3): the rotation that takes the optical axis e_z onto the unit viewing ray through pixel (u, v) --
    Rodrigues' formula about a = (-o_y, o_x, 0) / |.| by the angle acos(o_z); identical to the reference's
    axis_angle_to_matrix(angle * axis / |axis|) (math_util.py:609-622).  valid (n,) = angle > 0."""
    fx, fy, sx, sy = K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2]
    o = torch.stack(((u - sx) / fx, (v - sy) / fy, torch.ones_like(u)), dim=1)
    o = o / torch.linalg.norm(o, dim=1, keepdim=True)
    c = o[:, 2].clamp(-1.0, 1.0)
    s = torch.sqrt((o[:, 0] ** 2 + o[:, 1] ** 2).clamp(min=0))
    valid = torch.acos(c) > 0
    safe = torch.where(s > 0, s, torch.ones_like(s))
    ax, ay = -o[:, 1] / safe, o[:, 0] / safe
    z = torch.zeros_like(ax)
    Kx = torch.stack((z, z, ay, z, z, -ax, -ay, ax, z), dim=1).view(-1, 3, 3)          # [a]_x for a = (ax, ay, 0)
    a = torch.stack((ax, ay, z), dim=1)
    eye = torch.eye(3, dtype=K.dtype, device=K.device).expand(len(K), 3, 3)
    M = c[:, None, None] * eye + s[:, None, None] * Kx + (1 - c)[:, None, None] * (a[:, :, None] * a[:, None, :])
    return M, valid


def R_to_allocentric(K, R, u=None, v=None):
    """math_util.py:595-648: egocentric -> allocentric (viewpoint-normalised) rotation(s) for objects seen at pixel (u, v).
    Batched tensor form: K (n,3,3), R (n,3,3), u, v (n,); array form: K 3x3, R 3x3 (u, v default to the principal point)."""
    if isinstance(K, torch.Tensor):
        M, valid = _viewing_rotation(K, u, v)
        return torch.where(valid[:, None, None], torch.bmm(M.transpose(2, 1), R), R)
    import numpy as np
    Kt = torch.as_tensor(np.asarray(K, dtype=np.float64)).reshape(1, 3, 3)
    ut = torch.tensor([float(Kt[0, 0, 2] if u is None else u)], dtype=torch.float64)
    vt = torch.tensor([float(Kt[0, 1, 2] if v is None else v)], dtype=torch.float64)
    M, valid = _viewing_rotation(Kt, ut, vt)
    R = np.asarray(R)
    return (M[0].numpy().T @ R) if bool(valid[0]) else R


def R_from_allocentric(K, R_view, u=None, v=None):
    """math_util.py:651-705: the inverse of R_to_allocentric (allocentric -> egocentric).  In the training / inference path
    this is fused into csrc/cube_head.hip (cube_decode / cube_loss); this is the standalone entry point."""
    if isinstance(K, torch.Tensor):
        M, valid = _viewing_rotation(K, u, v)
        return torch.where(valid[:, None, None], torch.bmm(M, R_view), R_view)
    import numpy as np
    Kt = torch.as_tensor(np.asarray(K, dtype=np.float64)).reshape(1, 3, 3)
    ut = torch.tensor([float(Kt[0, 0, 2] if u is None else u)], dtype=torch.float64)
    vt = torch.tensor([float(Kt[0, 1, 2] if v is None else v)], dtype=torch.float64)
    M, valid = _viewing_rotation(Kt, ut, vt)
    R_view = np.asarray(R_view)
    return (M[0].numpy() @ R_view) if bool(valid[0]) else R_view
