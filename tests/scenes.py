"""Seeded synthetic scenes shared by the CPU and GPU tests (SURVEY.md 8d recipe).

The oracle renders the ground-truth view that feature maps are built from; that
is test infrastructure, never the product path.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from nemo_amd import synthetic as syn
from oracle import oracle as orc


def preprocess(verts):
    """pre_process_mesh_pascal (lib/MeshUtils.py:115-117)."""
    return np.stack([verts[:, 0], verts[:, 2], -verts[:, 1]], 1).astype(np.float32)


@dataclass
class Scene:
    cfg: object
    geo: syn.Geometry
    D: int
    verts: np.ndarray      # pre-processed
    faces: np.ndarray
    bank: np.ndarray
    cvec: np.ndarray
    feats: np.ndarray      # [N,h,w,D]
    clus: np.ndarray       # [N,h,w]
    gt: np.ndarray         # [N,4] az, el, theta, dist
    distance: float
    cate: str


def make_scene(cate='car', image_hw=(256, 384), number_vertices=300, D=128, n_img=2, seed=0,
               perspective_correct=False, clutter=True, k_eps=1e-8, sigma=0.5, jitter=None):
    rng = np.random.default_rng(seed)
    verts, faces = syn.class_mesh(cate, number_vertices, jitter=jitter)
    vp = preprocess(verts)
    geo = syn.Geometry.from_image_size(image_hw)
    s = syn.ndc_scale(12.0)
    cfg = orc.make_cfg(geo.S, geo.h, geo.w, D, s, geo.r0, geo.c0, perspective_correct, clutter, k_eps)
    bank = syn.unit_rows(rng, len(verts), D)
    cvec = syn.unit_rows(rng, 1, D)[0]
    dist = float(syn.DISTANCE_RENDER[cate])
    az, el, th, ds = syn.random_gt_poses(rng, n_img, dist)
    feats, clus = [], []
    for i in range(n_img):
        C = syn.spherical_to_campos(ds[i], el[i], az[i])
        R, T = orc.campos_to_R_T(C, th[i])
        scr = orc.project(vp, R, T, s)
        p2f, bary, _ = orc.rasterize(cfg, scr, faces)
        rend = orc.interpolate(cfg, p2f, bary, faces, bank)
        f = syn.make_feature_maps(rng, rend[None], (p2f >= 0)[None], sigma)[0]
        feats.append(f)
        clus.append(orc.clutter_score(cfg, f, cvec))
    return Scene(cfg, geo, D, vp, faces, bank, cvec, np.stack(feats), np.stack(clus),
                 np.stack([az, el, th, ds], 1).astype(np.float32), dist, cate)


def random_poses(rng, n, distance, spread=0.15):
    az = rng.uniform(0, 2 * np.pi, n)
    el = rng.uniform(-np.pi / 6, np.pi / 3, n)
    th = rng.uniform(-np.pi / 6, np.pi / 6, n).astype(np.float32)
    ds = distance * rng.uniform(1 - spread, 1 + spread, n)
    return syn.spherical_to_campos(ds, el, az), th
