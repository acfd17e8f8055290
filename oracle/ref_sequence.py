"""CPU baseline of SURVEY 8(d): the reference's OWN op sequence for one image, single-threaded.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg and tests import it; the product path never
does).  PARITY UNPINNED like the rest of oracle/: PyTorch3D is not runnable here, so its two native pieces are
stood in for by the build's restatements -- the naive all-faces-per-pixel rasteriser over the full S x S square
(oracle/nemo_oracle.c, the loop of PyTorch3D's RasterizeMeshesNaiveCpu) and its backward, which
is obtained here by re-deriving the barycentrics of the covered pixels with torch ops so that autograd
differentiates them (row P5).  Everything else is what the reference executes, op for op, with torch on the CPU:

  per Adam step (code/MeshPoseSolveAll.py:235-250):
    campos_to_R_T: look_at_rotation, rotation_theta, bmm            lib/MeshUtils.py:104-111
    transform the V vertices, perspective divide                       MeshRasterizer.transform (row P3)
    rasterize_meshes naive over S x S, K = 1                           lib/MeshUtils.py:83-85
    interpolate_face_attributes: gather [S,S,3,D], mul, sum            lib/MeshUtils.py:129 (PyTorch3D 0.2.0 form)
    [S,S,D] -> [D,S,S], centre crop to [D,h,w]                         lib/MeshUtils.py:131-132,148-150,226
    object_score = sum(projected * predicted, dim 0)                   code/MeshPoseSolveAll.py:237
    loss_fun                                                           :58-59
    loss.backward(); torch.optim.Adam.step(); zero_grad()              :240-245
    camera_position_to_spherical_angle + 4 x .item()                   :247-248
    ExponentialLR.step every 100 steps                                 :249-250
  per image before the loop: 1x1 conv2d clutter score (:220) and get_init_pos_rendered over the 144
  pre-rendered maps (:84-100); per category: get_pre_render_samples, 144 renders under no_grad (:62-81).

`python -m oracle.ref_sequence --steps K ...` times K Adam steps, a few grid evaluations and a few pre-renders of
ONE synthetic car image (BASELINE configs[0]: 640 x 800, V = 1088, F = 1844, D = 128) on one thread and prints one
JSON line; bench.py starts one such process per host core at the same time (the reference's model: one
single-threaded process per category, InferenceNeMo.sh:67-78) and extrapolates to 144 + 300 evaluations.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np


def _torch():
    import torch
    torch.set_num_threads(1)
    return torch


class ReferenceSequence:
    """One (mesh, image) pair set up the way code/MeshPoseSolveAll.py:200-220 does."""

    def __init__(self, sc, image_index=0):
        torch = _torch()
        from oracle import oracle as orc, torch_ref
        self.torch, self.orc, self.tr, self.sc = torch, orc, torch_ref, sc
        cfg = sc.cfg
        self.cfg = cfg
        self.verts = torch.as_tensor(sc.verts)
        self.faces_np = np.ascontiguousarray(sc.faces, dtype=np.int32)
        faces = torch.as_tensor(sc.faces.astype(np.int64))
        self.faces = faces
        self.face_memory = torch.as_tensor(sc.bank)[faces]                      # [F,3,D]  (lib/MeshUtils.py:144-145)
        self.predicted_map = torch.as_tensor(np.ascontiguousarray(sc.feats[image_index].transpose(2, 0, 1)))  # [D,h,w]
        clutter = torch.as_tensor(sc.cvec)[None]                                 # normalize(mean(bank)) [1,D] (:207)
        self.clutter_score = torch.nn.functional.conv2d(self.predicted_map[None], clutter[:, :, None, None])[0, 0]  # :220
        S = cfg.S
        idx = torch.arange(S, dtype=torch.float32)
        ndc = -1.0 + (2.0 * (S - 1 - idx) + 1.0) / S
        self.py, self.px = torch.meshgrid(ndc, ndc, indexing='ij')              # NDC of pixel (row, col)

    # ---- MeshInterpolateModule.forward (lib/MeshUtils.py:209-227)
    def render(self, C, theta):
        torch, cfg = self.torch, self.cfg
        R, T = self.tr.campos_to_R_T(C, theta)
        scr = self.tr.project(self.verts, R[0], T[0], cfg.ndc_scale)            # [V,3]
        with torch.no_grad():                                                    # rasterize_meshes forward (native, naive)
            p2f, _, _ = self.orc.rasterize(cfg, scr.detach().numpy(), self.faces_np, crop=False)
        p2f = torch.as_tensor(p2f.astype(np.int64))                              # [S,S], -1 = empty
        mask = p2f >= 0
        f = p2f.clamp(min=0)
        fv = scr[self.faces[f]]                                                  # [S,S,3,3]: the rasterizer's bary, differentiable
        v0, v1, v2 = fv[..., 0, :], fv[..., 1, :], fv[..., 2, :]
        p = torch.stack([self.px, self.py], dim=-1)
        area = self.tr._edge(v2, v0, v1) + cfg.k_eps
        bary = torch.stack([self.tr._edge(p, v1, v2) / area, self.tr._edge(p, v2, v0) / area,
                            self.tr._edge(p, v0, v1) / area], dim=-1)            # [S,S,3]
        bary = torch.where(mask[..., None], bary, torch.full_like(bary, -1.0))
        vals = self.face_memory[f]                                               # interpolate_face_attributes: gather [S,S,3,D]
        out = (bary[..., None] * vals).sum(dim=-2)
        out = torch.where(mask[..., None], out, torch.zeros_like(out))           # [S,S,D]
        out = out.permute(2, 0, 1)                                               # [D,S,S]
        return out[:, cfg.r0:cfg.r0 + cfg.h, cfg.c0:cfg.c0 + cfg.w]              # centre crop

    def loss_fun(self, obj_s):
        torch = self.torch
        if self.cfg.clutter:
            return torch.ones(1) - (torch.mean(torch.max(obj_s, self.clutter_score)) - torch.mean(self.clutter_score))
        return torch.ones(1) - torch.mean(obj_s)

    def grid_eval(self, maps):
        """get_init_pos_rendered (:84-100) over the given pre-rendered maps."""
        torch = self.torch
        with torch.no_grad():
            losses = [self.loss_fun(torch.sum(m * self.predicted_map, dim=0)).item() for m in maps]
        return int(np.argmin(losses))

    def adam_steps(self, C0, theta0, steps, lr=0.05, betas=(0.4, 0.6)):
        torch = self.torch
        C = torch.nn.Parameter(torch.as_tensor(np.asarray(C0, np.float32)).reshape(1, 3))
        theta = torch.nn.Parameter(torch.as_tensor(np.asarray([theta0], np.float32)))
        optim = torch.optim.Adam(params=[C, theta], lr=lr, betas=betas)
        sched = torch.optim.lr_scheduler.ExponentialLR(optim, gamma=0.2)
        recs, losses = [], []
        for epoch in range(steps):
            projected_map = self.render(C, theta)
            object_score = torch.sum(projected_map * self.predicted_map, dim=0)
            loss = self.loss_fun(object_score)
            loss.backward()
            optim.step()
            optim.zero_grad()
            d = torch.sum(C ** 2, dim=1) ** .5
            az = torch.atan(C[:, 0] / C[:, 2]) % np.pi + np.pi * (C[:, 0] < 0).float()
            el = torch.asin(C[:, 1] / d)
            recs.append([theta.item(), el.item(), az.item(), d.item()])
            losses.append(loss.item())
            if (epoch + 1) % 100 == 0:
                sched.step()
        return C.detach().numpy()[0], float(theta.detach()), losses


def measure(steps=8, grid_evals=16, prerenders=4, seed=0, image_hw=(640, 800), number_vertices=1000):
    """Times of the three pieces of a solve on one thread, and the per-image solve time they add up to."""
    torch = _torch()
    from nemo_amd import synthetic as syn
    from tests.scenes import make_scene
    t0 = time.perf_counter()
    sc = make_scene('car', image_hw, number_vertices, n_img=1, seed=seed)
    seq = ReferenceSequence(sc)
    t_setup = time.perf_counter() - t0
    gC, gth = syn.pose_grid(sc.distance)
    t0 = time.perf_counter()
    with torch.no_grad():
        maps = [seq.render(torch.as_tensor(gC[g:g + 1]), torch.as_tensor(gth[g:g + 1])) for g in range(prerenders)]
    t_render = (time.perf_counter() - t0) / prerenders
    many = [maps[i % prerenders] for i in range(grid_evals)]
    t0 = time.perf_counter()
    seq.grid_eval(many)
    t_grid = (time.perf_counter() - t0) / grid_evals
    az, el, th, ds = sc.gt[0]
    C0 = syn.spherical_to_campos(ds * 1.02, el + 0.05, az + 0.08)
    t0 = time.perf_counter()
    _, _, losses = seq.adam_steps(C0, th + 0.03, steps)
    t_step = (time.perf_counter() - t0) / steps
    G, n_steps, images_per_category = len(gth), 300, 16
    return dict(t_step_s=t_step, t_grid_eval_s=t_grid, t_prerender_s=t_render, t_setup_s=t_setup, steps_timed=steps,
                loss_first=losses[0], loss_last=losses[-1], V=int(len(sc.verts)), F=int(len(sc.faces)), S=int(sc.cfg.S),
                # one solve = 300 steps + 144 cached-grid evaluations + this image's share of the 144 pre-renders
                # (cfg 1: 16 images per category, code/MeshPoseSolveAll.py:62-81 renders them once per subtype)
                solve_s_extrapolated=n_steps * t_step + G * t_grid + G * t_render / images_per_category)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--grid-evals', type=int, default=16)
    ap.add_argument('--prerenders', type=int, default=4)
    ap.add_argument('--seed', type=int, default=0)
    args = ap.parse_args(argv)
    print(json.dumps(measure(args.steps, args.grid_evals, args.prerenders, args.seed)), flush=True)
    return 0


if __name__ == '__main__':
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.exit(main())
