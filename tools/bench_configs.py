#!/usr/bin/env python3
"""Timings of the BASELINE.json configurations that are not the bench.py line (configs[2..4]), full size,
one MI355X, inputs resident in HBM.  Parity for these shapes is in tests/test_gpu_parity.py
(test_multicuboid_argmin_over_meshes, test_dense_pose_grid_search, test_large_mesh_256d_double_resolution);
this script only measures.

    python tools/bench_configs.py [multi] [dense] [big]
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nemo_amd import synthetic as syn  # noqa: E402
from nemo_amd.pose_solver import BatchPoseSolver, SolverConfig, argmin_over_meshes  # noqa: E402
from nemo_amd.workloads import _render_features  # noqa: E402


def timed(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


def gt_poses(rng, cats_of_image):
    gt = np.zeros((len(cats_of_image), 4))
    for i, c in enumerate(cats_of_image):
        az, el, th, ds = syn.random_gt_poses(rng, 1, syn.DISTANCE_RENDER[c])
        gt[i] = (th[0], el[0], az[0], ds[0])
    return gt


def multi(n_images=1024, seed=3):
    """configs[2] on one GPU: MultiCuboid, mesh_len subtypes per class (79 meshes, jittered extents), every image
    solved with every subtype of its class, best final loss wins."""
    rng = np.random.default_rng(seed)
    cfg = SolverConfig()
    s = BatchPoseSolver(cfg)
    first, cats = {}, syn.CATEGORIES
    for c in cats:
        first[c] = len(s._grids_C)
        for _ in range(syn.MESH_LEN[c]):
            v, f = syn.class_mesh(c, 1000, jitter=rng.uniform(0.8, 1.25, 3))
            s.add_mesh(v, f, syn.unit_rows(rng, len(v), cfg.feat_dim), syn.DISTANCE_RENDER[c])
    cat_of = [cats[i % 12] for i in range(n_images)]
    true_mesh = np.array([first[c] + rng.integers(syn.MESH_LEN[c]) for c in cat_of], np.int32)
    feats = _render_features(s, torch.as_tensor(true_mesh).cuda(), gt_poses(rng, cat_of), rng, 0.5)
    img = np.concatenate([np.full(syn.MESH_LEN[c], i, np.int32) for i, c in enumerate(cat_of)])
    mesh = np.concatenate([first[c] + np.arange(syn.MESH_LEN[c], dtype=np.int32) for c in cat_of])
    img_t, mesh_t = torch.as_tensor(img).cuda(), torch.as_tensor(mesh).cuda()
    clutter = s.clutter_vector(torch.as_tensor(syn.unit_rows(rng, 64, cfg.feat_dim)).cuda())

    def run():
        res = s.solve_all(feats, clutter, mesh_t, img_idx=img_t)
        return argmin_over_meshes(res, n_images, img_t)

    dt, win = timed(run)
    hit = float((mesh[win.cpu().numpy()] == true_mesh).mean())
    return {'config': 'MultiCuboid, 79 meshes, argmin over the subtypes of the class', 'images': n_images, 'jobs': int(len(img)),
            'seconds': dt, 'images_per_s': n_images / dt, 'job_solves_per_s': len(img) / dt, 'generating_subtype_wins': hit}


def dense(n_images=1024, seed=4):
    """configs[3]: 36 azimuth x 9 elevation x 6 theta x 5 distance = 9720 grid poses per image, one class."""
    rng = np.random.default_rng(seed)
    cfg = SolverConfig()
    s = BatchPoseSolver(cfg)
    d = syn.DISTANCE_RENDER['car']
    v, f = syn.class_mesh('car', 1000)
    s.add_mesh(v, f, syn.unit_rows(rng, len(v), cfg.feat_dim), d)
    gC, gth = syn.pose_grid(d, n_azim=36, n_elev=9, n_theta=6, distances=tuple(d * x for x in (0.8, 0.9, 1.0, 1.1, 1.2)))
    dC, dth = torch.as_tensor(gC[None]).cuda().contiguous(), torch.as_tensor(gth[None]).cuda().contiguous()
    mesh_idx = torch.zeros(n_images, dtype=torch.int32, device='cuda')
    feats = _render_features(s, mesh_idx, gt_poses(rng, ['car'] * n_images), rng, 0.5)
    eng = s.engine
    clu = eng.clutter_score(feats, s.clutter_vector(torch.as_tensor(syn.unit_rows(rng, 64, cfg.feat_dim)).cuda()))
    os.environ.setdefault('NEMO_PRERENDER_MAP_GB', '64')
    t0 = time.perf_counter()
    eng.grid_prerender(dC, dth, [(0, 0)])
    torch.cuda.synchronize()
    t_pre = time.perf_counter() - t0
    dt, _ = timed(lambda: eng.grid_search(feats, clu, dC, dth, mesh_idx=mesh_idx, grid_idx=mesh_idx))
    G = len(gth)
    flops = 2.0 * cfg.feat_dim * G * n_images * eng.h * eng.w
    return {'config': 'dense pose grid', 'images': n_images, 'poses_per_image': G, 'prerender_seconds': t_pre,
            'seconds': dt, 'pose_evaluations_per_s': n_images * G / dt, 'useful_TFLOPs': flops / dt / 1e12}


def big(n_images=512, steps=300, seed=5):
    """configs[4]: subdivided cuboid (10 466 vertices), 256-d features, 160 x 200 window, clutter loss."""
    rng = np.random.default_rng(seed)
    cfg = SolverConfig(image_size=(1280, 1600), feat_dim=256, total_epochs=steps)
    s = BatchPoseSolver(cfg)
    v, f = syn.class_mesh('car', 10000)
    s.add_mesh(v, f, syn.unit_rows(rng, len(v), 256), syn.DISTANCE_RENDER['car'])
    mesh_idx = torch.zeros(n_images, dtype=torch.int32, device='cuda')
    feats = _render_features(s, mesh_idx, gt_poses(rng, ['car'] * n_images), rng, 0.5, chunk=32)
    clutter = s.clutter_vector(torch.as_tensor(syn.unit_rows(rng, 64, 256)).cuda())
    eng = s.engine
    eng.counter_reset()
    dt, res = timed(lambda: s.solve_all(feats, clutter, mesh_idx), reps=1)
    cov = eng.counter_read() / 2  # warm-up + one timed pass
    return {'config': 'subdivided cuboid, 256-d, 160x200 window', 'images': n_images, 'V': int(len(v)), 'F': int(len(f)),
            'seconds': dt, 'solves_per_s': n_images / dt, 'finite': bool(torch.isfinite(res.campos).all()),
            'covered_pixel_evaluations': cov, 'algorithmic_GBps': cov * eng.bytes_per_covered_pixel() / dt / 1e9}


if __name__ == '__main__':
    which = sys.argv[1:] or ['multi', 'dense', 'big']
    for name in which:
        print(json.dumps({name: {'multi': multi, 'dense': dense, 'big': big}[name]()}), flush=True)
