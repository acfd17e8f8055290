#!/usr/bin/env python3
"""Device time of the two kernels of a tables-form solve on the bench workload (nemo_solve_last_timing)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nemo_amd.workloads import twelve_class_workload  # noqa: E402

cats = os.environ.get('CATS')
wl = twelve_class_workload(1024, seed=1000, device=torch.device('cuda', 0), categories=cats.split(',') if cats else None)
s, eng = wl.solver, wl.solver.engine
eng.reserve(1024, int(s.grid_theta.shape[1]))
clu = eng.clutter_score(wl.feats, wl.clutter)
C0, th0, _, _ = eng.grid_search(wl.feats, clu, s.grid_C, s.grid_theta, mesh_idx=wl.mesh_idx, grid_idx=wl.mesh_idx)
for rep in range(4):
    eng.solve(wl.feats, clu, C0, th0, mesh_idx=wl.mesh_idx)
    t = eng.solve_last_timing()
    print('rep %d: table %.2f ms, loop %.2f ms' % (rep, t[0], t[1]) if t else 'streaming')
V, hw, D = 1088, eng.h * eng.w, eng.D   # (flops of the car mesh; the 12 classes have 1024 - 1134 vertices)
print('table kernel: %.1f TFLOP/s' % (2.0 * V * hw * D * 1024 / (t[0] * 1e-3) / 1e12))
