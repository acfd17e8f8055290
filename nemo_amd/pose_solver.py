"""Host side of the pose-solve path: the batched counterpart of the reference's
code/MeshPoseSolveAll.py main loop (:199-254).

The reference solves one image at a time with ~100 tiny kernels per Adam step;
here a whole batch of (image, mesh) jobs goes through four launches that never
return to the host:  clutter score -> pose-grid search -> fused 300-step solve
-> (torch) decode of the final camera positions.

Function names follow the reference script where a counterpart exists.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import synthetic as syn
from .camera_params import cal_err, cal_rotation_matrix, pose_error  # noqa: F401  (re-exported)
from .engine import PoseEngine


def camera_position_to_spherical_angle(camera_pose: torch.Tensor):
    """lib/MeshUtils.py:230-234 -> (distance, elevation, azimuth)."""
    distance_o = torch.sum(camera_pose ** 2, dim=1) ** .5
    azimuth_o = torch.atan(camera_pose[:, 0] / camera_pose[:, 2]) % np.pi + np.pi * (
        camera_pose[:, 0] < 0).to(camera_pose.dtype)
    elevation_o = torch.asin(camera_pose[:, 1] / distance_o)
    return distance_o, elevation_o, azimuth_o


def camera_position_from_spherical_angles(distance, elevation, azimuth, degrees=False, device='cpu'):
    """PyTorch3D helper used at code/MeshPoseSolveAll.py:70,110."""
    d = torch.as_tensor(distance, dtype=torch.float32, device=device).reshape(-1)
    e = torch.as_tensor(elevation, dtype=torch.float32, device=device).reshape(-1)
    a = torch.as_tensor(azimuth, dtype=torch.float32, device=device).reshape(-1)
    if degrees:
        e, a = e * (np.pi / 180.0), a * (np.pi / 180.0)
    d, e, a = torch.broadcast_tensors(d, e, a)
    return torch.stack([d * torch.cos(e) * torch.sin(a), d * torch.sin(e), d * torch.cos(e) * torch.cos(a)], dim=1)


def normalize(x, dim=0):
    """code/MeshPoseSolveAll.py:123-124."""
    return x / torch.sum(x ** 2, dim=dim, keepdim=True)[0] ** .5


def loss_fun0(sim):
    """code/MeshPoseSolveAll.py:50-51."""
    return torch.ones(1, device=sim.device) - torch.mean(sim)


def loss_fun(obj_s, clu_s):
    """code/MeshPoseSolveAll.py:58-59 (occlusion-aware loss)."""
    return torch.ones(1, device=obj_s.device) - (torch.mean(torch.max(obj_s, clu_s)) - torch.mean(clu_s))


def standard_loss_func_with_clutter(obj_s, clu_s):
    """lib/MeshUtils.py:273-275 (K clutter maps, max over K)."""
    clu_s = torch.max(clu_s, dim=1)[0]
    return torch.ones(1, device=obj_s.device) - (torch.mean(torch.max(obj_s, clu_s)) - torch.mean(clu_s))


@dataclass
class SolverConfig:
    """Everything the reference scatters over argparse flags and hard-coded
    tables (code/MeshPoseSolveAll.py:8-28,138-151,180-195)."""
    image_size: Sequence[int] = (640, 800)
    down_sample_rate: int = 8
    feat_dim: int = 128
    fov: float = 12.0
    total_epochs: int = 300
    lr: float = 5e-2
    adam_beta_0: float = 0.4
    adam_beta_1: float = 0.6
    adam_eps: float = 1e-8
    lr_decay_every: int = 100
    lr_decay_gamma: float = 0.2
    n_azimuth: int = 12
    n_elevation: int = 4
    n_theta: int = 3
    turn_off_clutter: bool = False
    perspective_correct: bool = False
    k_eps: float = 1e-8
    # code/MeshPoseSolveAll.py:14,209-212: False = no stored grid renders (get_init_pos renders every (image, pose))
    pre_render: bool = True
    # 6-DoF variant (code/PredUnlabeledDataset.py:283-347): Adam also moves the 2-D translation; the grid
    # samples three distances and theta = 0
    six_dof: bool = False
    grid_distances: Optional[Sequence[float]] = None
    grid_thetas: Optional[Sequence[float]] = None

    @staticmethod
    def unlabeled_6dof(**kw) -> 'SolverConfig':
        """The settings of code/PredUnlabeledDataset.py:283-286 (cars: image 256x672 in that script)."""
        kw.setdefault('six_dof', True)
        kw.setdefault('grid_distances', (4.4, 5.0, 5.7))
        kw.setdefault('grid_thetas', (0.0,))
        return SolverConfig(**kw)


@dataclass
class SolveResult:
    theta: torch.Tensor
    distance: torch.Tensor
    elevation: torch.Tensor
    azimuth: torch.Tensor
    loss: torch.Tensor
    campos: torch.Tensor
    init_index: torch.Tensor
    trace: Optional[torch.Tensor] = None
    trans_2d: Optional[torch.Tensor] = None   # [N,2], 6-DoF variant only
    status: Optional[torch.Tensor] = None     # int32 [N] from the library: 1 = finite final pose, 0 = not

    def preds6(self) -> torch.Tensor:
        """[N,6] rows of final_pred.npz (code/PredUnlabeledDataset.py:344-347, README.md:96):
        distance, theta, elevation, azimuth, trans_2d (2)."""
        t = self.trans_2d if self.trans_2d is not None else torch.zeros_like(self.campos[:, :2])
        return torch.cat([torch.stack([self.distance, self.theta, self.elevation, self.azimuth], dim=1), t], dim=1)

    def packed(self) -> torch.Tensor:
        """[N, 8] f32 record gathered across ranks: dist, elev, azim, theta, loss, init, status, pad."""
        # the library's per-job flag (nemo_solve's d_status); results built by hand may lack it
        ok = self.status != 0 if self.status is not None else torch.isfinite(self.campos).all(dim=1) & torch.isfinite(self.theta)
        return torch.stack([self.distance, self.elevation, self.azimuth, self.theta, self.loss,
                            self.init_index.to(torch.float32), ok.to(torch.float32),
                            torch.zeros_like(self.loss)], dim=1).contiguous()


class BatchPoseSolver:
    """Batched MeshPoseSolveAll: one PoseEngine, one mesh + pose grid per
    subtype/category, jobs = (image, mesh)."""

    def __init__(self, cfg: SolverConfig = SolverConfig(), device='cuda:0'):
        self.cfg = cfg
        H, W = cfg.image_size
        self.geo = syn.Geometry.from_image_size((H, W), cfg.down_sample_rate)
        self.engine = PoseEngine(self.geo.S, (self.geo.h, self.geo.w), cfg.feat_dim, cfg.fov,
                                 cfg.perspective_correct, not cfg.turn_off_clutter, cfg.k_eps, device)
        self.device = self.engine.device
        self.engine.set_schedule(cfg.total_epochs, cfg.lr, (cfg.adam_beta_0, cfg.adam_beta_1), cfg.adam_eps,
                                 cfg.lr_decay_every, cfg.lr_decay_gamma)
        self._grids_C: List[np.ndarray] = []
        self._grids_t: List[np.ndarray] = []
        self.grid_C = None
        self.grid_theta = None

    def add_mesh(self, verts_pascal, faces, memory_bank, distance, distances=None) -> int:
        """xvert/xface as load_off returns them (PASCAL3D+ axes) + the vertex feature bank
        (code/MeshPoseSolveAll.py:200-204) + the category's render distance (:150-151,169)."""
        v = np.asarray(verts_pascal, dtype=np.float32)
        vp = np.stack([v[:, 0], v[:, 2], -v[:, 1]], 1)  # pre_process_mesh_pascal, lib/MeshUtils.py:115-117
        mid = self.engine.register_mesh(vp, faces, memory_bank)
        gC, gt = syn.pose_grid(distance, self.cfg.n_azimuth, self.cfg.n_elevation, self.cfg.n_theta,
                               distances if distances is not None else self.cfg.grid_distances, self.cfg.grid_thetas)
        self._grids_C.append(gC)
        self._grids_t.append(gt)
        self.grid_C = torch.as_tensor(np.stack(self._grids_C)).to(self.device).contiguous()
        self.grid_theta = torch.as_tensor(np.stack(self._grids_t)).to(self.device).contiguous()
        return mid

    def prerender(self, force: bool = False):
        """get_pre_render_samples (code/MeshPoseSolveAll.py:62-82), once per set of meshes: every mesh's
        coverage under each of its grid poses.  A no-op when the store is current; the store bakes in the bank
        and grid CONTENTS, so after changing those tensors in place pass force=True."""
        if not self.cfg.pre_render:
            return
        key = (self.grid_C.data_ptr(), self.grid_theta.data_ptr(), len(self._grids_C))
        if force or getattr(self, '_pre_key', None) != key:
            self.engine.grid_prerender(self.grid_C, self.grid_theta, [(m, m) for m in range(len(self._grids_C))])
            self._pre_key = key

    def clutter_vector(self, clutter_bank: torch.Tensor) -> torch.Tensor:
        """normalize(mean(clutter_bank, 0)) (code/MeshPoseSolveAll.py:207) -> [1, D]."""
        return normalize(torch.mean(clutter_bank, dim=0)).unsqueeze(0).contiguous()

    def clutter_scores_by_group(self, feats: torch.Tensor, group_of_image: Sequence[int],
                                clutter_of_group: Sequence[torch.Tensor]) -> torch.Tensor:
        """Clutter score maps [n_img,h,w] when images of different subtypes use different clutter banks
        (code/MeshPoseSolveAll.py:203-207 reads `clutter_<subtype>` inside the subtype loop): images are
        scored in runs of equal group, each run against its own [K,D] bank."""
        g = np.asarray(group_of_image)
        clu = torch.empty((feats.shape[0], self.geo.h, self.geo.w), dtype=torch.float32, device=self.device)
        start = 0
        while start < len(g):
            stop = start + 1
            while stop < len(g) and g[stop] == g[start]:
                stop += 1
            clu[start:stop] = self.engine.clutter_score(feats[start:stop], clutter_of_group[int(g[start])])
            start = stop
        return clu

    def solve_all(self, feats: torch.Tensor, clutter: Optional[torch.Tensor], mesh_idx: Optional[torch.Tensor] = None,
                  img_idx: Optional[torch.Tensor] = None, want_trace=False,
                  clutter_scores: Optional[torch.Tensor] = None) -> SolveResult:
        """feats [n_img,h,w,D] pixel-major f32 on the device; clutter [K,D] (K=1: the reference's
        mean vector; K>1: max over the K vectors, lib/MeshUtils.py:273-275) or, instead, ready-made
        clutter_scores [n_img,h,w] (clutter_scores_by_group); jobs default to image i with mesh mesh_idx[i]."""
        eng = self.engine
        if self.cfg.turn_off_clutter:
            clu = None
        elif clutter_scores is not None:
            clu = clutter_scores
        else:
            clu = eng.clutter_score(feats, clutter)
        n_jobs = img_idx.shape[0] if img_idx is not None else feats.shape[0]
        self.prerender()
        C0, th0, best, _ = eng.grid_search(feats, clu, self.grid_C, self.grid_theta, n_jobs=n_jobs,
                                           img_idx=img_idx, mesh_idx=mesh_idx, grid_idx=mesh_idx)
        status = torch.empty((n_jobs,), dtype=torch.int32, device=self.device)
        if self.cfg.six_dof:
            out = eng.solve6(feats, clu, C0, th0, None, img_idx=img_idx, mesh_idx=mesh_idx, want_trace=want_trace,
                             status=status)
        else:
            out = eng.solve(feats, clu, C0, th0, img_idx=img_idx, mesh_idx=mesh_idx, want_trace=want_trace,
                            status=status)
        pose, loss = out[0], out[1]
        campos = pose[:, :3]
        dist, elev, azim = camera_position_to_spherical_angle(campos)
        return SolveResult(pose[:, 3], dist, elev, azim, loss, campos, best, out[2] if want_trace else None,
                           pose[:, 4:6].contiguous() if self.cfg.six_dof else None, status)


def features_to_pixel_major(predicted_map: torch.Tensor) -> torch.Tensor:
    """[.., D, h, w] (ExtractFeatures.py:166-168 layout) -> [.., h, w, D] contiguous."""
    return predicted_map.movedim(-3, -1).contiguous()


def argmin_over_meshes(result: SolveResult, n_img: int, img_idx: torch.Tensor):
    """MultiCuboid without labels (BASELINE config 3): keep, per image, the job with the lowest
    final loss.  Returns the winning job index per image."""
    loss = torch.where(torch.isfinite(result.loss), result.loss, torch.full_like(result.loss, float('inf')))
    best = torch.full((n_img,), float('inf'), device=loss.device)
    best = best.scatter_reduce(0, img_idx.long(), loss, reduce='amin')
    jobs = torch.arange(loss.shape[0], device=loss.device)
    cand = torch.where(loss == best[img_idx.long()], jobs, torch.full_like(jobs, loss.shape[0]))
    win = torch.full((n_img,), loss.shape[0], device=loss.device, dtype=jobs.dtype)
    return win.scatter_reduce(0, img_idx.long(), cand, reduce='amin')
