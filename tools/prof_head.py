#!/usr/bin/env python3
"""Times nemo_feature_head (backbone tail + clutter score) at the bench shape: N images of 80x100, C=256 -> D=128."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nemo_amd.engine import PoseEngine  # noqa: E402
ap = argparse.ArgumentParser(); ap.add_argument('--images', type=int, default=512); args = ap.parse_args()
h, w, C, D = 80, 100, 256, 128
eng = PoseEngine(100, (h, w), D)
x = torch.randn(args.images, C, h, w, device='cuda'); W = torch.randn(D, C, device='cuda') / 16
cb = torch.nn.functional.normalize(torch.randn(1, D, device='cuda'), dim=1)
eng.feature_head(x, W, cb)
for r in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        eng.feature_head(x, W, cb)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    fl = 2.0 * args.images * h * w * C * D
    by = args.images * h * w * (C + D + 1) * 4.0
    print(f'feature_head: {args.images} images {dt*1e3:.2f} ms  {fl/dt/1e12:.1f} TFLOP/s  {by/dt/1e9:.0f} GB/s')
