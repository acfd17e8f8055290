#!/usr/bin/env python3
"""Distribution of the final-pose difference between the HIP solve and the CPU oracle, both free-running from the
same grid-search start (north_star bar: 0.5 degrees).  Writes a JSON histogram (default gpurun_out/; the judged copy
is profiles/r02_final_pose_hist.json).

    python tools/final_pose_stats.py [n_images] [out.json]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402  (checker only)
from tests.test_gpu_parity import final_pose_errors  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, 'gpurun_out', 'final_pose_hist.json')
orc.lib()
errs, same, o_gt = final_pose_errors(orc, n, seed=6)
edges = [0, 0.01, 0.02, 0.05, 0.1, 0.2, 0.5, 1, 2, 5, 10, 180]
hist, _ = np.histogram(errs, bins=edges)
conv = o_gt < 2.0
rec = {'what': 'geodesic distance (degrees) between the final pose of libnemo_hip.so and of the CPU oracle, free-running, '
               'same inputs and same grid-search start; 12-class workload at 640x800, 144-pose grid + 300 Adam steps',
       'images': n, 'same_grid_argmin': int(same.sum()), 'within_0.5_deg': float(np.mean(errs <= 0.5)),
       'median_deg': float(np.median(errs)), 'p95_deg': float(np.percentile(errs, 95)), 'max_deg': float(errs.max()),
       'bin_edges_deg': edges, 'counts': hist.tolist(),
       'converged': {'definition': "the oracle's own final pose is within 2 degrees of the synthetic ground truth", 'images': int(conv.sum()),
                     'within_0.5_deg': float(np.mean(errs[conv] <= 0.5)), 'max_deg': float(errs[conv].max()),
                     'counts': np.histogram(errs[conv], bins=edges)[0].tolist()},
       'not_converged': {'images': int((~conv).sum()), 'within_0.5_deg': float(np.mean(errs[~conv] <= 0.5)) if (~conv).any() else None,
                         'counts': np.histogram(errs[~conv], bins=edges)[0].tolist()}}
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(rec, open(out, 'w'), indent=1)
print(json.dumps(rec))
