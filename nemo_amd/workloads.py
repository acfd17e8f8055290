"""Synthetic benchmark workloads built ON the GPU (no oracle, no reference data).

`twelve_class_workload` is BASELINE.json configs[1]: SingleCuboid, 12 categories,
synthetic 640x800 images -> 80x100x128 feature maps, one cuboid mesh (about 1000
vertices) and one 144-pose grid per category, random-init unit-norm vertex banks.
Feature maps are the HIP rasterizer + interpolation of each image's ground-truth
pose plus noise.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from . import synthetic as syn
from .pose_solver import BatchPoseSolver, SolverConfig


@dataclass
class Workload:
    solver: BatchPoseSolver
    feats: torch.Tensor       # [N,h,w,D]
    clutter: torch.Tensor     # [1,D]
    mesh_idx: torch.Tensor    # [N] int32
    gt: np.ndarray            # [N,4] theta, elev, azim, dist
    name: str


def _render_features(solver, mesh_idx, gt, rng, sigma, chunk=256):
    eng = solver.engine
    dev = eng.device
    N = len(mesh_idx)
    feats = torch.empty((N, eng.h, eng.w, eng.D), device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(rng.integers(1 << 31)))
    for s in range(0, N, chunk):
        e = min(N, s + chunk)
        C = torch.as_tensor(syn.spherical_to_campos(gt[s:e, 3], gt[s:e, 1], gt[s:e, 2])).to(dev)
        th = torch.as_tensor(gt[s:e, 0].astype(np.float32)).to(dev)
        R, T = eng.campos_to_R_T(C.contiguous(), th.contiguous())
        mi = mesh_idx[s:e].contiguous()
        p2f, bary = eng.rasterize(R, T, mi)
        rend = eng.interpolate(p2f, bary, mi)
        noise = torch.randn(rend.shape, device=dev, generator=gen)
        x = torch.where((p2f >= 0)[..., None], rend + sigma * noise, noise)
        feats[s:e] = x / x.norm(dim=-1, keepdim=True).clamp_min(1e-12)
        del rend, noise, x
    return feats


def twelve_class_workload(n_images=1024, seed=0, device='cuda:0', number_vertices=1000, sigma=0.5,
                          cfg: SolverConfig = None, categories=None) -> Workload:
    cfg = cfg or SolverConfig()
    rng = np.random.default_rng(seed)
    solver = BatchPoseSolver(cfg, device)
    cats = list(categories or syn.CATEGORIES)
    for c in cats:
        v, f = syn.class_mesh(c, number_vertices)
        bank = syn.unit_rows(rng, len(v), cfg.feat_dim)
        solver.add_mesh(v, f, bank, syn.DISTANCE_RENDER[c])
    mesh_idx_np = (np.arange(n_images) % len(cats)).astype(np.int32)
    gt = np.zeros((n_images, 4), np.float64)
    for i, m in enumerate(mesh_idx_np):
        az, el, th, ds = syn.random_gt_poses(rng, 1, syn.DISTANCE_RENDER[cats[m]])
        gt[i] = (th[0], el[0], az[0], ds[0])
    mesh_idx = torch.as_tensor(mesh_idx_np).to(solver.device)
    feats = _render_features(solver, mesh_idx, gt, rng, sigma)
    cbank = torch.as_tensor(syn.unit_rows(rng, 64, cfg.feat_dim)).to(solver.device)
    clutter = solver.clutter_vector(cbank)
    name = f'SingleCuboid {len(cats)}-class, {n_images} synthetic {cfg.image_size[0]}x{cfg.image_size[1]} images'
    return Workload(solver, feats, clutter, mesh_idx, gt, name)
