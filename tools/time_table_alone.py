#!/usr/bin/env python3
"""bank_table_kernel on the bench workload WITHOUT a solve loop behind it (one Adam step per solve), ten launches back
to back: separates the kernel's own rate from the clock state a 40 ms vector-ALU loop leaves the chip in."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nemo_amd.pose_solver import SolverConfig  # noqa: E402
from nemo_amd.workloads import twelve_class_workload  # noqa: E402

cats = os.environ.get('CATS')
wl = twelve_class_workload(1024, seed=1000, device=torch.device('cuda', 0), cfg=SolverConfig(total_epochs=1),
                           categories=cats.split(',') if cats else None)
s, eng = wl.solver, wl.solver.engine
eng.reserve(1024, int(s.grid_theta.shape[1]))
clu = eng.clutter_score(wl.feats, wl.clutter)
C0, th0, _, _ = eng.grid_search(wl.feats, clu, s.grid_C, s.grid_theta, mesh_idx=wl.mesh_idx, grid_idx=wl.mesh_idx)
ts = []
for rep in range(10):
    eng.solve(wl.feats, clu, C0, th0, mesh_idx=wl.mesh_idx)
    ts.append(eng.solve_last_timing())
print('table ms:', ' '.join('%.2f' % t[0] for t in ts), '| one-step loop ms: %.3f' % ts[-1][1])
