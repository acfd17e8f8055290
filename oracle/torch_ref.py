"""Independent backward for the oracle: the reference's op sequence written with
torch-CPU ops so that torch.autograd differentiates it.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (PyTorch3D is not runnable here):
coverage (pix_to_face) is an input -- the tests take it from oracle/raster_np.py, the array
rasterizer written independently of the C oracle, after checking the two agree; everything
differentiable -- look-at, rotation_theta (MeshUtils.py:57-80),
T = -R^T C (:104-111), the composed view/projection transform (P3), barycentrics
(P4), perspective correction, interpolate_face_attributes (P6), crop, object
score (MeshPoseSolveAll.py:237) and loss (:50-59) -- is re-expressed in torch so
that `loss.backward()` yields C.grad / theta.grad exactly as the reference's
autograd graph would.  Used to check the hand-derived backward in
nemo_oracle.c (and through it the HIP kernel).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def look_at_rotation(C, at=None, up=None):
    at = torch.zeros_like(C) if at is None else at
    up = torch.tensor([[0.0, 1.0, 0.0]], dtype=C.dtype).expand_as(C) if up is None else up
    z = F.normalize(at - C, eps=1e-5)
    x = F.normalize(torch.cross(up, z, dim=1), eps=1e-5)
    y = F.normalize(torch.cross(z, x, dim=1), eps=1e-5)
    R = torch.cat((x[:, None, :], y[:, None, :], z[:, None, :]), dim=1)
    return R.transpose(1, 2)


def rotation_theta(theta):
    c, s = torch.cos(theta), torch.sin(theta)
    o, z = torch.ones_like(c), torch.zeros_like(c)
    return torch.stack([c, -s, z, s, c, z, z, z, o], dim=-1).view(-1, 3, 3)


def campos_to_R_T(C, theta, extra_trans=None):
    R = torch.bmm(look_at_rotation(C), rotation_theta(theta))
    T = -torch.bmm(R.transpose(1, 2), C.unsqueeze(2))[:, :, 0]
    if extra_trans is not None:
        T = T + extra_trans
    return R, T


def project(verts, R, T, s):
    """verts [V,3]; R [3,3]; T [3] -> NDC x, y and view z."""
    view = verts @ R + T
    return torch.stack([s * view[:, 0] / view[:, 2], s * view[:, 1] / view[:, 2], view[:, 2]], dim=1)


def _edge(p, a, b):
    return (p[..., 0] - a[..., 0]) * (b[..., 1] - a[..., 1]) - (p[..., 1] - a[..., 1]) * (b[..., 0] - a[..., 0])


def loss_and_grads(cfg, verts, faces, bank, feat, clu, Cpos, theta, p2f_window, dtype=torch.float64,
                   extra=None):
    """Returns loss, dL/dC, dL/dtheta (and dL/dextra) through autograd."""
    S, h, w = cfg.S, cfg.h, cfg.w
    verts = torch.as_tensor(np.asarray(verts), dtype=dtype)
    faces_t = torch.as_tensor(np.asarray(faces), dtype=torch.long)
    bank = torch.as_tensor(np.asarray(bank), dtype=dtype)
    feat = torch.as_tensor(np.asarray(feat), dtype=dtype)  # [h,w,D]
    C = torch.tensor(np.asarray(Cpos, dtype=np.float64).reshape(1, 3), dtype=dtype, requires_grad=True)
    th = torch.tensor([float(theta)], dtype=dtype, requires_grad=True)
    ex = None
    if extra is not None:
        ex = torch.tensor(np.asarray(extra, dtype=np.float64).reshape(1, 3), dtype=dtype, requires_grad=True)
    R, T = campos_to_R_T(C, th, ex)
    scr = project(verts, R[0], T[0], cfg.ndc_scale)

    p2f = torch.as_tensor(np.asarray(p2f_window), dtype=torch.long)  # [h,w]
    cov = p2f >= 0
    ii, jj = torch.nonzero(cov, as_tuple=True)
    f = p2f[ii, jj]
    fv = scr[faces_t[f]]  # [n,3,3]
    v0, v1, v2 = fv[:, 0], fv[:, 1], fv[:, 2]
    px = -1.0 + (2.0 * (S - 1 - (jj + cfg.c0)).to(dtype) + 1.0) / S
    py = -1.0 + (2.0 * (S - 1 - (ii + cfg.r0)).to(dtype) + 1.0) / S
    p = torch.stack([px, py], dim=1)
    area = _edge(v2, v0, v1) + cfg.k_eps
    w0, w1, w2 = _edge(p, v1, v2) / area, _edge(p, v2, v0) / area, _edge(p, v0, v1) / area
    if cfg.perspective_correct:
        t0, t1, t2 = w0 * v1[:, 2] * v2[:, 2], v0[:, 2] * w1 * v2[:, 2], v0[:, 2] * v1[:, 2] * w2
        den = t0 + t1 + t2
        w0, w1, w2 = t0 / den, t1 / den, t2 / den
    fm = bank[faces_t[f]]  # [n,3,D]
    proj = w0[:, None] * fm[:, 0] + w1[:, None] * fm[:, 1] + w2[:, None] * fm[:, 2]
    obj = torch.zeros(h, w, dtype=dtype)
    obj = obj.index_put((ii, jj), (proj * feat[ii, jj]).sum(dim=1))
    if cfg.clutter and clu is not None:
        clu_t = torch.as_tensor(np.asarray(clu), dtype=dtype)
        loss = 1.0 - (torch.mean(torch.max(obj, clu_t)) - torch.mean(clu_t))
    else:
        loss = 1.0 - torch.mean(obj)
    loss.backward()
    out = dict(loss=float(loss.detach()), gC=C.grad[0].numpy().copy(), gtheta=float(th.grad[0]),
               obj=obj.detach().numpy())
    if ex is not None:
        out['gextra'] = ex.grad[0].numpy().copy()
    return out


def adam_reference(p0, grads, lr=0.05, betas=(0.4, 0.6), decay_every=100, gamma=0.2):
    """torch.optim.Adam + ExponentialLR.step every `decay_every` steps on a fixed gradient
    sequence (MeshPoseSolveAll.py:230-231,244-250); returns the parameter trajectory."""
    p = torch.nn.Parameter(torch.tensor(np.asarray(p0, dtype=np.float32)))
    opt = torch.optim.Adam([p], lr=lr, betas=betas)
    sch = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=gamma)
    traj = []
    for t, g in enumerate(grads):
        p.grad = torch.tensor(np.asarray(g, dtype=np.float32))
        opt.step()
        opt.zero_grad()
        traj.append(p.detach().numpy().copy())
        if (t + 1) % decay_every == 0:
            sch.step()
    return np.stack(traj)
