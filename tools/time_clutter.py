#!/usr/bin/env python3
"""Times nemo_clutter_score on the bench-size feature maps (1024 x 80 x 100 x 128) for a few bank sizes."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nemo_amd.engine import PoseEngine
eng = PoseEngine(100, (80, 100), 128)
feats = torch.randn(1024, 80, 100, 128, device='cuda')
for K in (1, 4, 8, 16, 32, 64, 512, 1088, 2560):
    cb = torch.nn.functional.normalize(torch.randn(K, 128, device='cuda'), dim=1).contiguous()
    eng.clutter_score(feats, cb); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): out = eng.clutter_score(feats, cb)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    ref = (feats[:2].reshape(-1, 128) @ cb.T).max(dim=1)[0]
    err = (out[:2].reshape(-1) - ref).abs().max().item()
    print(f'K={K}: {dt*1e3:.3f} ms, {feats.numel()*4/dt/1e12:.2f} TB/s, {2.0*feats.numel()*K/dt/1e12:.1f} TFLOP/s, max |diff| vs torch {err:.2e}')
