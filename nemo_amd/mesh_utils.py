"""Drop-in counterpart of the reference's code/lib/MeshUtils.py call surface.

Same names, argument meaning and tensor shapes as the reference module, so that
code written against it (code/MeshPoseSolveAll.py, PredUnlabeledDataset.py,
tools/PlotLossLandscape.py) runs unchanged -- but `MeshInterpolateModule.forward`
calls the HIP rasterizer / interpolation kernels of libnemo_hip.so instead of
PyTorch3D, and its backward is the hand-written kernel behind
nemo_interpolate_backward.  The small PyTorch3D value types the reference
passes around (camera, raster settings, rasterizer, fragments) are plain
holders here.  The fast path for whole batches is nemo_amd.pose_solver.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .engine import PoseEngine
from .pose_solver import (camera_position_from_spherical_angles, camera_position_to_spherical_angle,  # noqa: F401
                          normalize, standard_loss_func_with_clutter)


# ----------------------------------------------------------------------------- OFF files
def load_off(off_file_name, to_torch=False):
    """lib/MeshUtils.py:24-42: line 2 = "V F E", V vertex lines, F lines "3 i j k"."""
    with open(off_file_name) as fh:
        lines = fh.readlines()
    n_points = int(lines[1].split(' ')[0])
    verts = np.array([[float(t) for t in ln.split()] for ln in lines[2:2 + n_points]], dtype=np.float32).reshape(-1, 3)
    faces = np.array([[int(t) for t in ln.split()] for ln in lines[2 + n_points:] if ln.strip()],
                     dtype=np.int32).reshape(-1, 4)[:, 1:]
    if to_torch:
        return torch.from_numpy(verts), torch.from_numpy(np.ascontiguousarray(faces))
    return verts, faces


def save_off(off_file_name, vertices, faces):
    """lib/MeshUtils.py:45-54."""
    with open(off_file_name, 'w') as fl:
        fl.write('OFF\n%d %d 0\n' % (len(vertices), len(faces)))
        for v in vertices:
            fl.write('%.16f %.16f %.16f\n' % (v[0], v[1], v[2]))
        for f in faces:
            fl.write('3 %d %d %d\n' % (f[0], f[1], f[2]))


# ----------------------------------------------------------------------------- camera
def rotation_theta(theta, device_=None):
    """In-plane rotation [[c,-s,0],[s,c,0],[0,0,1]] per sample (lib/MeshUtils.py:57-80)."""
    if isinstance(theta, float):
        theta = torch.ones((1,), device=device_ or 'cpu') * theta
    theta = theta.reshape(-1)
    c, s = torch.cos(theta), torch.sin(theta)
    z, o = torch.zeros_like(c), torch.ones_like(c)
    return torch.stack([c, -s, z, s, c, z, z, z, o], dim=1).view(-1, 3, 3)


def look_at_rotation(camera_position, at=((0, 0, 0),), up=((0, 1, 0),), device='cpu'):
    """PyTorch3D look_at_rotation: axes as COLUMNS of R."""
    C = torch.as_tensor(camera_position, dtype=torch.float32, device=device).reshape(-1, 3)
    at = torch.as_tensor(at, dtype=torch.float32, device=C.device).reshape(-1, 3).expand_as(C)
    up = torch.as_tensor(up, dtype=torch.float32, device=C.device).reshape(-1, 3).expand_as(C)
    z_axis = F.normalize(at - C, eps=1e-5)
    x_axis = F.normalize(torch.cross(up, z_axis, dim=1), eps=1e-5)
    y_axis = F.normalize(torch.cross(z_axis, x_axis, dim=1), eps=1e-5)
    return torch.cat((x_axis[:, None, :], y_axis[:, None, :], z_axis[:, None, :]), dim=1).transpose(1, 2)


def campos_to_R_T(campos, theta, device='cpu', at=((0, 0, 0),), up=((0, 1, 0),), extra_trans=None):
    """lib/MeshUtils.py:104-111 (torch ops, differentiable)."""
    R = torch.bmm(look_at_rotation(campos, at=at, up=up, device=campos.device), rotation_theta(theta, device_=campos.device))
    T = -torch.bmm(R.transpose(1, 2), campos.unsqueeze(2))[:, :, 0]
    if extra_trans is not None:
        T = T + extra_trans
    return R, T


def pre_process_mesh_pascal(verts):
    """(x, y, z) -> (x, z, -y)  (lib/MeshUtils.py:115-117)."""
    return torch.cat((verts[:, 0:1], verts[:, 2:3], -verts[:, 1:2]), dim=1)


# ----------------------------------------------------------------------------- PyTorch3D-shaped holders
@dataclass
class OpenGLPerspectiveCameras:
    """Camera holder (code/MeshPoseSolveAll.py:180): only `fov` (degrees) is used; znear=1, aspect=1."""
    device: object = 'cpu'
    fov: float = 60.0

    def to(self, device):
        self.device = device
        return self


@dataclass
class RasterizationSettings:
    """code/MeshPoseSolveAll.py:181-186.  Only the naive K=1, blur 0 configuration is implemented."""
    image_size: int = 256
    blur_radius: float = 0.0
    faces_per_pixel: int = 1
    bin_size: Optional[int] = 0
    perspective_correct: bool = False


@dataclass
class MeshRasterizer:
    cameras: OpenGLPerspectiveCameras
    raster_settings: RasterizationSettings


@dataclass
class Fragments:
    pix_to_face: torch.Tensor
    zbuf: Optional[torch.Tensor]
    bary_coords: torch.Tensor
    dists: Optional[torch.Tensor] = None


class _CenterCrop:
    """center_crop_fun (lib/MeshUtils.py:148-150): BboxTools box of `out_shape` centred at max_shape // 2."""

    def __init__(self, out_shape, max_shape):
        self.out_shape = (int(out_shape[0]), int(out_shape[1]))
        self.max_shape = (int(max_shape[0]), int(max_shape[1]))
        self.r0 = self.max_shape[0] // 2 - self.out_shape[0] // 2
        self.c0 = self.max_shape[1] // 2 - self.out_shape[1] // 2

    def __call__(self, x):
        return x[..., self.r0:self.r0 + self.out_shape[0], self.c0:self.c0 + self.out_shape[1]]


def center_crop_fun(out_shape, max_shape):
    return _CenterCrop(out_shape, max_shape)


def vertex_memory_to_face_memory(memory_bank, faces):
    """lib/MeshUtils.py:144-145."""
    return memory_bank[faces.type(torch.long)]


def set_bary_coords_to_nearest(bary_coords_):
    """lib/MeshUtils.py:136-141."""
    ori_shape = bary_coords_.shape
    exr = bary_coords_ * (bary_coords_ < 0)
    flat = bary_coords_.reshape(-1, ori_shape[-1])
    idx = flat.argmax(1)
    return torch.zeros_like(flat).scatter(1, idx.unsqueeze(1), 1.0).view(*ori_shape) + exr


# ----------------------------------------------------------------------------- the module
class _RenderInterpolate(torch.autograd.Function):
    """rasterize + interpolate_face_attributes + crop as one differentiable op of (R, T)."""

    @staticmethod
    def forward(ctx, R, T, engine, mesh_idx):
        R, T = R.contiguous(), T.contiguous()
        p2f, bary = engine.rasterize(R, T, mesh_idx)
        out = engine.interpolate(p2f, bary, mesh_idx)  # [n, h, w, D]
        ctx.save_for_backward(R, T)
        ctx.engine, ctx.mesh_idx = engine, mesh_idx
        return out.permute(0, 3, 1, 2)  # a view, like the reference's transposes (lib/MeshUtils.py:131-132)

    @staticmethod
    def backward(ctx, grad_out):
        R, T = ctx.saved_tensors
        g = grad_out.permute(0, 2, 3, 1).contiguous()
        gR, gT = ctx.engine.interpolate_backward(R, T, g, ctx.mesh_idx)
        return gR, gT, None, None


class MeshInterpolateModule(nn.Module):
    """lib/MeshUtils.py:153-227.  forward(campos [n,3], theta [n]) -> [n, D, h, w]."""

    def __init__(self, vertices, faces, memory_bank, rasterizer, post_process=None, off_set_mesh=False):
        super().__init__()
        if off_set_mesh:
            raise NotImplementedError('deform_verts / off_set_mesh is not on the MeshPoseSolveAll path')
        self.rasterizer = rasterizer
        self.post_process = post_process
        self.off_set_mesh = off_set_mesh
        multi = isinstance(vertices, (list, tuple))
        self._verts = [pre_process_mesh_pascal(v) for v in (vertices if multi else [vertices])]
        self._faces = list(faces) if multi else [faces]
        self._banks = list(memory_bank) if multi else [memory_bank]
        self.n_mesh = len(self._verts)
        self.faces = self._faces if multi else self._faces[0]
        self.face_memory = None  # materialised lazily; the kernels index the bank through faces
        self.engine = None

    # the reference keeps face_memory; kept here for code that reads it
    def update_memory(self, memory_bank, faces=None):
        """lib/MeshUtils.py:183-193."""
        multi = isinstance(memory_bank, (list, tuple))
        self._banks = list(memory_bank) if multi else [memory_bank]
        if faces is not None:
            self._faces = list(faces) if multi else [faces]
        if self.engine is not None:
            self._build(self.engine.device)

    def _build(self, device):
        rs = self.rasterizer.raster_settings
        if rs.blur_radius != 0 or rs.faces_per_pixel != 1 or rs.bin_size not in (0, None):
            raise NotImplementedError('only blur_radius=0, faces_per_pixel=1, naive rasterization is implemented')
        S = int(rs.image_size)
        if isinstance(self.post_process, _CenterCrop):
            shape, origin = self.post_process.out_shape, (self.post_process.r0, self.post_process.c0)
        elif self.post_process is None:
            shape, origin = (S, S), (0, 0)
        else:
            raise NotImplementedError('post_process must be center_crop_fun(...) or None')
        D = int(self._banks[0].shape[1])
        self.engine = PoseEngine(S, shape, D, fov=float(self.rasterizer.cameras.fov),
                                 perspective_correct=bool(getattr(rs, 'perspective_correct', False)),
                                 clutter=True, device=device, crop_origin=origin)
        for v, f, b in zip(self._verts, self._faces, self._banks):
            self.engine.register_mesh(v.detach().float(), f, b.detach().float())
        self.face_memory = torch.cat([vertex_memory_to_face_memory(b, f).to(device) for b, f in
                                      zip(self._banks, self._faces)], dim=0)

    def to(self, *args, **kwargs):
        device = kwargs.get('device', args[0] if args else None)
        super().to(device)
        self.rasterizer.cameras = self.rasterizer.cameras.to(device)
        self._build(torch.device(device))
        return self

    def cuda(self, device=None):
        return self.to(torch.device('cuda:0' if device is None else device))

    def forward(self, campos, theta, blur_radius=0, deform_verts=None, mode='bilinear', **kwargs):
        if self.engine is None:
            raise RuntimeError('MeshInterpolateModule must be moved to a GPU first (.cuda()); there is no CPU path')
        if blur_radius != 0 or deform_verts is not None or mode != 'bilinear':
            raise NotImplementedError('blur_radius > 0, deform_verts and mode="nearest" are not on the solver path')
        R, T = campos_to_R_T(campos, theta, device=campos.device, **kwargs)
        n_cam = campos.shape[0]
        if n_cam > 1 and self.n_mesh > 1:
            if n_cam != self.n_mesh:
                raise ValueError('one camera per mesh expected')
            mesh_idx = torch.arange(n_cam, dtype=torch.int32, device=campos.device)
        else:
            mesh_idx = None  # one mesh, n cameras (the reference extends the mesh, :220-221)
        return _RenderInterpolate.apply(R, T, self.engine, mesh_idx)

    # extras for callers that want the fragments (rasterize / forward_interpolate of the reference)
    def fragments(self, R, T, mesh_idx=None):
        p2f, bary, zb = self.engine.rasterize(R.contiguous(), T.contiguous(), mesh_idx, want_zbuf=True)
        return Fragments(pix_to_face=p2f.long()[..., None], zbuf=zb[..., None], bary_coords=bary[:, :, :, None, :])
