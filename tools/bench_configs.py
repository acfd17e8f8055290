#!/usr/bin/env python3
"""Timings of the BASELINE.json configurations that are not the bench.py line (configs[2..4]), full size,
one MI355X, inputs resident in HBM.  Parity for these shapes is in tests/test_gpu_parity.py
(test_multicuboid_argmin_over_meshes, test_dense_pose_grid_search, test_large_mesh_256d_double_resolution);
this script only measures.

    python tools/bench_configs.py [multi] [dense] [dense12] [big]
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


_ENV = ('NEMO_PRERENDER_MAP_GB', 'NEMO_GRID_NO_MFMA', 'NEMO_GRID_RECORDS', 'NEMO_GRID_GROUP')


class _env:
    """Environment switches for one context (they are read at nemo_ctx_create) -- put back afterwards, or the
    configurations that follow in this process inherit them (round 2's "0.835 s against 0.652 s" for `big`,
    profiles/r03_cfg5_discrepancy.md)."""

    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.saved = {k: os.environ.get(k) for k in _ENV}
        for k in _ENV:
            os.environ.pop(k, None)
        os.environ.update({k: str(v) for k, v in self.kv.items()})

    def __exit__(self, *exc):
        for k, v in self.saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def dense(n_images=1024, seed=4):
    """configs[3]: 36 azimuth x 9 elevation x 6 theta x 5 distance = 9720 grid poses per image, one class.  The default
    operand of the matrix-core contraction (the record store, alike poses grouped) and, for comparison, the 40 GB of
    dense renders and the record store in grid order."""
    out = {}
    for mode, env in (('default', {}), ('dense_maps', {'NEMO_GRID_RECORDS': 0}), ('records_in_grid_order', {'NEMO_GRID_GROUP': 0})):
        with _env(**env):
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
        t0 = time.perf_counter()
        eng.grid_prerender(dC, dth, [(0, 0)])
        torch.cuda.synchronize()
        t_pre = time.perf_counter() - t0
        dt, res = timed(lambda: eng.grid_search(feats, clu, dC, dth, mesh_idx=mesh_idx, grid_idx=mesh_idx))
        G = len(gth)
        eng.counter_reset()
        eng.grid_search(feats, clu, dC, dth, mesh_idx=mesh_idx, grid_idx=mesh_idx)
        cov = eng.counter_read()                                  # covered (pixel, pose, image) triples of one search
        out[mode] = {'store': eng.grid_store_kind(), 'images': n_images, 'poses_per_image': G, 'prerender_seconds': t_pre,
                     'seconds': dt, 'pose_evaluations_per_s': n_images * G / dt,
                     'window_TFLOPs': 2.0 * cfg.feat_dim * G * n_images * eng.h * eng.w / dt / 1e12,
                     'useful_TFLOPs_covered_pixels_only': 2.0 * cfg.feat_dim * cov / dt / 1e12, 'best': res[2][:8].tolist()}
        del s, eng, feats, clu
        torch.cuda.empty_cache()
    out['same_argmin_on_the_first_images'] = out['default']['best'] == out['dense_maps']['best'] == out['records_in_grid_order']['best']
    return {'config': 'dense pose grid, one class', **out}


def dense12(n_images=1024, seed=4):
    """configs[3] on the 12-class workload: 9720 poses x 12 meshes of dense renders (478 GB) do not fit.  Default: the
    19 GB record store, rows built on the fly.  Also timed: block-after-block rendering into the map buffer
    (NEMO_GRID_RECORDS=0, round 2's path) and the list path (no matrix cores)."""
    cfg = SolverConfig()
    out = {}
    for mode, env in (('default', {}), ('pose_blocks', {'NEMO_GRID_RECORDS': 0}), ('list_path', {'NEMO_GRID_NO_MFMA': 1})):
        with _env(**env):
            s = BatchPoseSolver(cfg)
            gCs, gts = [], []
            for c in syn.CATEGORIES:
                v, f = syn.class_mesh(c, 1000)
                d = syn.DISTANCE_RENDER[c]
                s.add_mesh(v, f, syn.unit_rows(np.random.default_rng(seed + 1), len(v), cfg.feat_dim), d)
                gC, gth = syn.pose_grid(d, n_azim=36, n_elev=9, n_theta=6, distances=tuple(d * x for x in (0.8, 0.9, 1.0, 1.1, 1.2)))
                gCs.append(gC); gts.append(gth)
            dC, dth = torch.as_tensor(np.stack(gCs)).cuda().contiguous(), torch.as_tensor(np.stack(gts)).cuda().contiguous()
            cats = [syn.CATEGORIES[i % 12] for i in range(n_images)]
            mesh_idx = torch.as_tensor((np.arange(n_images) % 12).astype(np.int32)).cuda()
            feats = _render_features(s, mesh_idx, gt_poses(np.random.default_rng(seed + 2), cats), np.random.default_rng(seed + 3), 0.5)
            eng = s.engine
        clu = eng.clutter_score(feats, s.clutter_vector(torch.as_tensor(syn.unit_rows(np.random.default_rng(seed + 4), 64, cfg.feat_dim)).cuda()))
        G = dth.shape[1]
        eng.reserve(n_images, G)
        t0 = time.perf_counter()
        eng.grid_prerender(dC, dth, [(m, m) for m in range(12)])
        torch.cuda.synchronize()
        t_pre = time.perf_counter() - t0
        n = n_images if mode != 'list_path' else 128          # the list path is slow: time a slice of the images
        idx = torch.arange(n, dtype=torch.int32, device='cuda')
        run = lambda: eng.grid_search(feats, clu, dC, dth, img_idx=idx, mesh_idx=mesh_idx[:n].contiguous(),  # noqa: E731
                                      grid_idx=mesh_idx[:n].contiguous())
        dt, res = timed(run, reps=1)
        eng.counter_reset(); run(); cov = eng.counter_read()
        out[mode] = {'store': eng.grid_store_kind(), 'images': n, 'poses_per_image': G, 'prerender_seconds': t_pre, 'seconds': dt,
                     'pose_evaluations_per_s': n * G / dt, 'useful_TFLOPs_covered_pixels_only': 2.0 * cfg.feat_dim * cov / dt / 1e12,
                     'best': res[2][:8].tolist()}
        del s, eng, feats, clu
        torch.cuda.empty_cache()
    out['same_argmin_on_the_first_images'] = out['default']['best'] == out['pose_blocks']['best'] == out['list_path']['best']
    return {'config': 'dense pose grid, 12 classes (9720 poses x 12 meshes)', **out}


def big(n_images=512, steps=300, seed=5):
    """configs[4]: subdivided cuboid (10 466 vertices), 256-d features, 160 x 200 window, clutter loss."""
    rng = np.random.default_rng(seed)
    cfg = SolverConfig(image_size=(1280, 1600), feat_dim=256, total_epochs=steps)
    s = BatchPoseSolver(cfg)
    v, f = syn.class_mesh('car', 10000)
    s.add_mesh(v, f, syn.unit_rows(rng, len(v), 256), syn.DISTANCE_RENDER['car'])
    mesh_idx = torch.zeros(n_images, dtype=torch.int32, device='cuda')
    feats = _render_features(s, mesh_idx, gt_poses(rng, ['car'] * n_images), rng, 0.5, chunk=32)
    # config 5 keeps the K-vector clutter bank: score = max over the K vectors (lib/MeshUtils.py:273-275), the
    # pixels x clutter-bank contraction on the matrix cores
    clutter = torch.as_tensor(syn.unit_rows(rng, 64, 256)).cuda()
    eng = s.engine
    eng.counter_reset()
    dt, res = timed(lambda: s.solve_all(feats, clutter, mesh_idx), reps=1)
    cov = eng.counter_read() / 2  # warm-up + one timed pass
    return {'config': 'subdivided cuboid, 256-d, 160x200 window, 64-vector clutter bank (max over K)', 'images': n_images, 'V': int(len(v)), 'F': int(len(f)),
            'seconds': dt, 'solves_per_s': n_images / dt, 'finite': bool(torch.isfinite(res.campos).all()),
            'covered_pixel_evaluations': cov, 'algorithmic_GBps': cov * eng.bytes_per_covered_pixel() / dt / 1e9}


if __name__ == '__main__':
    which = sys.argv[1:] or ['multi', 'dense', 'dense12', 'big']
    for name in which:
        print(json.dumps({name: {'multi': multi, 'dense': dense, 'dense12': dense12, 'big': big}[name]()}), flush=True)
