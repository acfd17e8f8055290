#!/usr/bin/env python3
"""Solve-only driver for profiling: builds the 12-class workload, runs grid search once, then the
fused solve `--reps` times.  Used under rocprofv3 (kernel trace / PMC passes)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nemo_amd.pose_solver import SolverConfig  # noqa: E402
from nemo_amd.workloads import twelve_class_workload  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--images', type=int, default=512)
ap.add_argument('--steps', type=int, default=60)
ap.add_argument('--reps', type=int, default=2)
ap.add_argument('--grid', action='store_true', help='also time the grid search')
ap.add_argument('--sort', action='store_true', help='order the jobs by descending coverage at the initial pose (longest first)')
ap.add_argument('--image-size', default='640,800', help='H,W of the synthetic images (feature maps are 1/8 of it)')
ap.add_argument('--distinct-images', type=int, default=0, help='locality probe: job j reads the feature map of image j %% K (poses and meshes stay per '
                'job, so with --lr0 the work is unchanged and only the live set of feature bytes shrinks to K maps)')
ap.add_argument('--lr0', action='store_true', help='zero learning rate: every step re-evaluates the initial pose (fixed work for ablations)')
args = ap.parse_args()
wl = twelve_class_workload(args.images, seed=1000, cfg=SolverConfig(total_epochs=args.steps, image_size=tuple(int(v) for v in args.image_size.split(',')),
                                                                   **({'lr': 0.0} if args.lr0 else {})))
s, eng = wl.solver, wl.solver.engine
clu = eng.clutter_score(wl.feats, wl.clutter)
C0, th0, _, _ = eng.grid_search(wl.feats, clu, s.grid_C, s.grid_theta, mesh_idx=wl.mesh_idx, grid_idx=wl.mesh_idx)
torch.cuda.synchronize()
img_idx, mesh_idx = None, wl.mesh_idx
if args.sort:
    _, _, ncov = eng.score_fwd_bwd(wl.feats, clu, C0, th0, mesh_idx=wl.mesh_idx, want_grad=False)
    perm = torch.argsort(ncov, descending=True)
    print('coverage at the initial pose: min %d median %d max %d' % (ncov.min().item(), ncov.median().item(), ncov.max().item()))
    img_idx, mesh_idx = perm.to(torch.int32), wl.mesh_idx[perm].contiguous()
    C0, th0 = C0[perm].contiguous(), th0[perm].contiguous()
    torch.cuda.synchronize()
if args.distinct_images > 0:
    base = torch.arange(args.images, device=C0.device, dtype=torch.int32) if img_idx is None else img_idx
    img_idx = (base % args.distinct_images).to(torch.int32).contiguous()
for r in range(args.reps):
    if os.environ.get('NEMO_HIP_LIB') and hasattr(eng.lib, 'nemo_debug_stamps'):
        import ctypes
        eng.lib.nemo_debug_stamps((ctypes.c_ulonglong * 16)())
    eng.counter_reset()
    t0 = time.perf_counter()
    eng.solve(wl.feats, clu, C0, th0, img_idx=img_idx, mesh_idx=mesh_idx)
    cov = eng.counter_read()
    dt = time.perf_counter() - t0
    print(f'solve: {args.images} images x {args.steps} steps: {dt*1e3:.2f} ms, {dt*1e6/args.steps:.1f} us/step, '
          f'cov/step/img {cov/args.steps/args.images:.0f}, alg GB/s {cov*eng.bytes_per_covered_pixel()/dt/1e9:.0f}')
    print(f'  covered pixel-steps {cov}')
    if hasattr(eng.lib, 'nemo_solve_table_stats'):
        form, blocks = eng.solve_table_stats()
        nb = -(-max(int(v.shape[0]) for v, _, _ in eng._keep) // 32) * (-(-eng.h // 4) * -(-eng.w // 8))   # 32 vertices x (4 x 8 pixels)
        print(f'  solve form {form} (0 streaming, 1 whole tables, 2 on demand); table blocks computed on demand: {blocks} = '
              f'{blocks / args.images:.0f} per job of {nb} ({100.0 * blocks / args.images / nb:.1f} %)')
    tl = eng.solve_last_timing() if hasattr(eng.lib, 'nemo_solve_last_timing') else None
    if tl:
        print(f'  tables form: bank_table_kernel {tl[0]:.2f} ms, solve_kernel {tl[1]:.2f} ms = {tl[1]*1e3/args.steps/((args.images+255)//256):.2f} us per job-step')
if os.environ.get('NEMO_HIP_LIB'):
    import ctypes
    st = (ctypes.c_ulonglong * 16)()
    if hasattr(eng.lib, 'nemo_debug_stamps') and eng.lib.nemo_debug_stamps(st) == 0:
        names = ['project+bbox', 'zclear', 'raster', 'list', 'score', 'band-barrier', 'reduce', 'lead(adam)']
        tot = sum(st[:8]) or 1
        print('score split: phaseA %.1f%% phaseB %.1f%% of total' % (100*st[8]/(tot+st[8]+st[9]-st[4]), 100*st[9]/(tot+st[8]+st[9]-st[4])))
        print('phase shares (wave-0 cycles, last solve): ' + ', '.join(f'{n} {100*st[i]/tot:.1f}%' for i, n in enumerate(names)))
        per = 1.0 / (args.images * args.steps)
        print('wave-0 us per job-step (100 MHz wall clock): ' + ', '.join(f'{n} {st[i]*per/100:.2f}' for i, n in enumerate(names)) +
              f', phaseA {st[8]*per/100:.2f}, phaseB {st[9]*per/100:.2f}')
        print(f'all waves, us per job-step: score loop busy {st[11]*per/100/16:.2f} (mean over 16 waves), wait at the band barrier {st[10]*per/100/16:.2f}')
    wb = (ctypes.c_ulonglong * 16)()
    if hasattr(eng.lib, 'nemo_debug_wave_busy') and eng.lib.nemo_debug_wave_busy(wb) == 0 and sum(wb):
        m = sum(wb) / 16.0
        print('score-loop busy time by wave index, relative to the mean (all solves since the last read): ' +
              ' '.join(f'{x / m:.3f}' for x in wb))
if args.grid:
    s.prerender()
    for r in range(args.reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.grid_search(wl.feats, clu, s.grid_C, s.grid_theta, mesh_idx=wl.mesh_idx, grid_idx=wl.mesh_idx)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f'grid: {dt*1e3:.2f} ms')
