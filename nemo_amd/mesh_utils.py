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


def _pack_face_ids(p2f, n_faces, mesh_idx):
    """PyTorch3D returns PACKED face ids (batch element n offset by the faces of the elements before it,
    SURVEY 8a row P4); the kernels return ids local to the job's mesh."""
    n = p2f.shape[0]
    nf = torch.as_tensor(n_faces, dtype=torch.int64, device=p2f.device)
    per = nf[mesh_idx.long()] if mesh_idx is not None else nf[:1].expand(n)
    off = torch.cumsum(per, 0) - per
    p2f = p2f.long()
    return torch.where(p2f >= 0, p2f + off.view(n, 1, 1), p2f)


class MeshInterpolateModule(nn.Module):
    """lib/MeshUtils.py:153-227.  forward(campos [n,3], theta [n]) -> [n, D, h, w]."""

    def __init__(self, vertices, faces, memory_bank, rasterizer, post_process=None, off_set_mesh=False):
        super().__init__()
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
        if blur_radius != 0:
            raise NotImplementedError('blur_radius > 0 (soft rasterization) is not on the solver path')
        if mode not in ('bilinear', 'nearest'):
            raise ValueError('mode must be "bilinear" or "nearest"')
        R, T = campos_to_R_T(campos, theta, device=campos.device, **kwargs)
        mesh_idx = self._mesh_idx(campos.shape[0], campos.device)
        self._apply_offsets(deform_verts)
        if mode == 'nearest':
            # lib/MeshUtils.py:126-127: barycentrics replaced by the one-hot of their arg-max; the scatter cuts the
            # autograd graph in the reference as well, so this mode carries no pose gradient
            with torch.no_grad():
                p2f, bary = self.engine.rasterize(R.detach().contiguous(), T.detach().contiguous(), mesh_idx)
                hot = torch.where((p2f >= 0)[..., None], set_bary_coords_to_nearest(bary), bary)
                return self.engine.interpolate(p2f, hot.contiguous(), mesh_idx).permute(0, 3, 1, 2)
        return _RenderInterpolate.apply(R, T, self.engine, mesh_idx)

    def _mesh_idx(self, n_cam, device):
        if n_cam > 1 and self.n_mesh > 1:                 # one camera per mesh (:218-219)
            if n_cam != self.n_mesh:
                raise ValueError('one camera per mesh expected')
            return torch.arange(n_cam, dtype=torch.int32, device=device)
        if self.n_mesh > 1:
            raise ValueError('a module holding several meshes needs one camera per mesh')
        return None                                       # one mesh, n cameras (the reference extends the mesh, :220-221)

    def _apply_offsets(self, deform_verts):
        """off_set_mesh (lib/MeshUtils.py:212-213): meshes.offset_verts(deform_verts) moves the packed vertices
        before rendering.  Forward only: the kernels have no dL/d(vertex) output, so a deformation that requires
        grad is refused."""
        if not self.off_set_mesh:
            return
        if deform_verts is None:
            raise ValueError('off_set_mesh=True needs deform_verts')
        if deform_verts.requires_grad:
            raise NotImplementedError('gradients with respect to deform_verts are not implemented')
        start = 0
        for (v_dev, _, _), base in zip(self.engine._keep, self._verts):
            n = base.shape[0]
            v_dev.copy_(base.to(v_dev.device, torch.float32) + deform_verts[start:start + n].to(v_dev.device, torch.float32))
            start += n

    # extras for callers that want the fragments (rasterize / forward_interpolate of the reference)
    def fragments(self, R, T, mesh_idx=None):
        """Fragments of rasterize (lib/MeshUtils.py:83-85) in PyTorch3D's shapes; pix_to_face holds PACKED ids."""
        if mesh_idx is None:
            mesh_idx = self._mesh_idx(R.shape[0], R.device)
        p2f, bary, zb = self.engine.rasterize(R.contiguous(), T.contiguous(), mesh_idx, want_zbuf=True)
        packed = _pack_face_ids(p2f, [int(f.shape[0]) for f in self._faces], mesh_idx)
        return Fragments(pix_to_face=packed[..., None], zbuf=zb[..., None], bary_coords=bary[:, :, :, None, :])


# ----------------------------------------------------------------------------- free functions of the reference module
class Meshes:
    """Holder with the part of PyTorch3D's Meshes the reference touches (lib/MeshUtils.py:168,176,213,221):
    lists of PRE-PROCESSED vertices and faces, `extend`, `offset_verts`, `to`."""

    def __init__(self, verts, faces, textures=None, _root=None, _map=None):
        self._verts = [torch.as_tensor(v, dtype=torch.float32) for v in verts]
        self._faces = [torch.as_tensor(f) for f in faces]
        self._engines = {}
        # extend(n) replicates a mesh for n cameras (lib/MeshUtils.py:221); the replicas are the same geometry, so
        # they are served by the engine of the Meshes they came from: element j renders mesh _map[j] of _root
        self._root = _root
        self._map = _map

    def __len__(self):
        return len(self._verts)

    def verts_list(self):
        return self._verts

    def faces_list(self):
        return self._faces

    def verts_packed(self):
        return torch.cat(self._verts, 0)

    def extend(self, n):
        root = self._root if self._root is not None else self
        base = self._map if self._map is not None else list(range(len(self)))
        return Meshes([v for v in self._verts for _ in range(n)], [f for f in self._faces for _ in range(n)],
                      _root=root, _map=[m for m in base for _ in range(n)])

    def offset_verts(self, offsets):
        out, start = [], 0
        for v in self._verts:
            out.append(v + offsets[start:start + v.shape[0]].to(v.device))
            start += v.shape[0]
        return Meshes(out, self._faces)   # new geometry: its own engine

    def to(self, device):
        return Meshes([v.to(device) for v in self._verts], [f.to(device) for f in self._faces])

    def cuda(self):
        return self.to('cuda')


def _engine_for(meshes: Meshes, rasterizer: MeshRasterizer, face_memory=None):
    """(engine, mesh index per batch element).  ONE PoseEngine per (root Meshes, raster settings, feature width):
    it renders the full S x S square, with `face_memory` [sum F, 3, D] presented to the kernels as a bank of 3 F rows
    indexed by faces (3f, 3f+1, 3f+2).  A face memory that changed since the last call (another tensor, or the same
    one updated in place) is copied into the registered bank -- no new context, no new registration; extend(n)
    replicas share their root's engine."""
    rs = rasterizer.raster_settings
    if rs.blur_radius != 0 or rs.faces_per_pixel != 1 or rs.bin_size not in (0, None):
        raise NotImplementedError('only blur_radius=0, faces_per_pixel=1, naive rasterization is implemented')
    root = meshes._root if meshes._root is not None else meshes
    emap = meshes._map if meshes._map is not None else list(range(len(meshes)))
    dev = root._verts[0].device
    D = None if face_memory is None else int(face_memory.shape[-1])
    key = (int(rs.image_size), float(rasterizer.cameras.fov), bool(getattr(rs, 'perspective_correct', False)), D)
    # rows of the (possibly tiled) face memory that belong to each root mesh: those of its first replica
    starts, first, acc = {}, {}, 0
    for j, m in enumerate(emap):
        if m not in first:
            first[m], starts[m] = j, acc
        acc += int(meshes._faces[j].shape[0])
    sig = None if face_memory is None else (face_memory.data_ptr(), face_memory._version, tuple(face_memory.shape),
                                            tuple(sorted(starts.items())))
    ent = root._engines.get(key)
    if ent is None:
        S = key[0]
        eng = PoseEngine(S, (S, S), 128 if D is None else D, fov=key[1], perspective_correct=key[2], device=dev,
                         crop_origin=(0, 0))
        for v, f in zip(root._verts, root._faces):
            F_ = int(f.shape[0])
            if face_memory is None:
                eng.register_mesh(v, f, torch.zeros((v.shape[0], 128), device=dev))
            else:  # corner c of face f reads row 3 f + c of the flattened face memory
                corner = torch.arange(3 * F_, device=dev, dtype=torch.int32).view(F_, 3)
                vv = v.to(dev)[f.long().to(dev)].reshape(3 * F_, 3).contiguous()
                eng.register_mesh(vv, corner, torch.zeros((3 * F_, D), device=dev))
        ent = root._engines[key] = {'eng': eng, 'sig': None}
    if face_memory is not None and ent['sig'] != sig:
        for m, (_, _, bank) in enumerate(ent['eng']._keep):
            if m in starts:
                F_ = bank.shape[0] // 3
                bank.copy_(face_memory[starts[m]:starts[m] + F_].reshape(3 * F_, D).to(dev, torch.float32))
        ent['sig'] = sig
    return ent['eng'], torch.as_tensor(emap, dtype=torch.int32, device=dev)


def rasterize(R, T, meshes, rasterizer, blur_radius=0):
    """lib/MeshUtils.py:83-101 (blur_radius = 0): Fragments of the full S x S render, one batch element per mesh
    of `meshes` (extend(n) for n cameras on one mesh), packed face ids."""
    if blur_radius > 0:
        raise NotImplementedError('blur_radius > 0 (soft rasterization) is not on the solver path')
    eng, mesh_idx = _engine_for(meshes, rasterizer)
    n = R.shape[0]
    if n != len(meshes):
        raise ValueError('%d cameras for %d meshes (use meshes.extend(n))' % (n, len(meshes)))
    p2f, bary, zb = eng.rasterize(R.detach().contiguous(), T.detach().contiguous(), mesh_idx, want_zbuf=True)
    elem = torch.arange(n, dtype=torch.int32, device=R.device)   # packed ids count the faces of every batch element
    packed = _pack_face_ids(p2f, [int(f.shape[0]) for f in meshes._faces], elem)
    return Fragments(pix_to_face=packed[..., None], zbuf=zb[..., None], bary_coords=bary[:, :, :, None, :])


def forward_interpolate(R, T, meshes, face_memory, rasterizer, blur_radius=0, mode='bilinear'):
    """lib/MeshUtils.py:122-133 -> [n, D, S, S]; differentiable in (R, T) for mode='bilinear'."""
    if blur_radius > 0:
        raise NotImplementedError('blur_radius > 0 (soft rasterization) is not on the solver path')
    eng, mesh_idx = _engine_for(meshes, rasterizer, face_memory)
    n = R.shape[0]
    if n != len(meshes):
        raise ValueError('%d cameras for %d meshes (use meshes.extend(n))' % (n, len(meshes)))
    if mode == 'nearest':
        with torch.no_grad():
            p2f, bary = eng.rasterize(R.detach().contiguous(), T.detach().contiguous(), mesh_idx)
            hot = torch.where((p2f >= 0)[..., None], set_bary_coords_to_nearest(bary), bary)
            return eng.interpolate(p2f, hot.contiguous(), mesh_idx).permute(0, 3, 1, 2)
    return _RenderInterpolate.apply(R, T, eng, mesh_idx)


def angel_gradient_modifier(base, grad_=None, alpha=(1.0, 1.0), center_=None):
    """lib/MeshUtils.py:237-261 (unused by the solver; commented out at code/MeshPoseSolveAll.py:241-242): rescales
    the radial and tangential parts of a camera-position gradient by alpha[0] / alpha[1]."""
    write_back = grad_ is None
    g = base.grad if write_back else grad_
    rel = base.detach() - center_ if center_ is not None else base.detach()
    with torch.no_grad():
        unit = rel / rel.pow(2).sum(dim=1).sqrt()
        radial = (unit * g).sum(dim=1, keepdim=True) * unit
        out = alpha[0] * radial + alpha[1] * (g - radial)
    if write_back:
        base.grad = out
    return out


def decompose_pose(pose, sorts=('distance', 'elevation', 'azimuth', 'theta')):
    """lib/MeshUtils.py:264-266."""
    return tuple(pose[:, sorts.index(k)] for k in ('distance', 'elevation', 'azimuth', 'theta'))
