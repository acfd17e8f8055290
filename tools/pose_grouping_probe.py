#!/usr/bin/env python3
"""How many (32-pose tile, pixel) pairs of the dense grid are live under different pose orders?  The record-based grid
search skips a tile at a pixel none of its 32 poses covers; this measures what a permuted record store could skip.
Coverage masks come from the library's own rasteriser.  usage: python tools/pose_grouping_probe.py [class]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nemo_amd import synthetic as syn                                     # noqa: E402
from nemo_amd.pose_solver import BatchPoseSolver, SolverConfig             # noqa: E402

cate = sys.argv[1] if len(sys.argv) > 1 else 'car'
cfg = SolverConfig()
s = BatchPoseSolver(cfg)
d = syn.DISTANCE_RENDER[cate]
v, f = syn.class_mesh(cate, 1000)
s.add_mesh(v, f, syn.unit_rows(np.random.default_rng(0), len(v), cfg.feat_dim), d)
NA, NE, NT, ND = 36, 9, 6, 5
gC, gth = syn.pose_grid(d, n_azim=NA, n_elev=NE, n_theta=NT, distances=tuple(d * x for x in (0.8, 0.9, 1.0, 1.1, 1.2)))
G = len(gth)
eng = s.engine
R, T = eng.campos_to_R_T(torch.as_tensor(gC).cuda(), torch.as_tensor(gth).cuda())
masks = []
for a in range(0, G, 1024):
    p2f, _ = eng.rasterize(R[a:a + 1024], T[a:a + 1024])
    masks.append((p2f >= 0).reshape(p2f.shape[0], -1))
M = torch.cat(masks).float()                                              # [G, hw]
hw = M.shape[1]
cover = M.mean().item()


def live(order, tile=32):
    Mo = M[torch.as_tensor(order, device=M.device)]
    pad = (-len(order)) % tile
    if pad:
        Mo = torch.cat([Mo, torch.zeros(pad, hw, device=M.device)])
    return Mo.reshape(-1, tile, hw).amax(1).mean().item() * (len(order) + pad) / len(order)


idx = np.arange(G).reshape(NA, NE, NT, ND)
out = {'class': cate, 'poses': G, 'pixels': hw, 'mean_coverage': cover, 'live_fraction_of_32_pose_tiles': {}}
L = out['live_fraction_of_32_pose_tiles']
L['grid order (az, el, theta, dist)'] = live(idx.reshape(-1))
L['(theta, dist, az, el)'] = live(idx.transpose(2, 3, 0, 1).reshape(-1))
L['(theta, dist, el, az)'] = live(idx.transpose(2, 3, 1, 0).reshape(-1))
L['(dist, theta, el, az)'] = live(idx.transpose(3, 2, 1, 0).reshape(-1))
area = M.sum(1)
L['sorted by covered area'] = live(torch.argsort(area).cpu().numpy())
# greedy: coarse signatures (8 x 10 blocks of the window), tiles grown from the unassigned pose of largest area by
# taking the 31 unassigned poses with the smallest Hamming distance to it
h, w = eng.h, eng.w
sig = torch.nn.functional.avg_pool2d(M.reshape(G, 1, h, w), 4).reshape(G, -1)
free = torch.ones(G, dtype=torch.bool, device=M.device)
order = []
by_area = torch.argsort(area, descending=True)
for seed in by_area.tolist():
    if not free[seed]:
        continue
    dist = (sig - sig[seed]).abs().sum(1)
    dist[~free] = float('inf')
    pick = torch.topk(dist, min(32, int(free.sum())), largest=False).indices
    free[pick] = False
    order += pick.tolist()
L['greedy by 4x4-block signature'] = live(np.asarray(order))
out['any_pose_of_a_128_block_greedy_order'] = live(np.asarray(order), 128)


def greedy_bits(block):
    """the form a device kernel can run: one bit per block x block pixels (any pixel covered), Hamming distance to the
    seed, seeds in descending order of covered area, ties by pose index"""
    bits = (torch.nn.functional.max_pool2d(M.reshape(G, 1, h, w), block, ceil_mode=True).reshape(G, -1) > 0)
    free = torch.ones(G, dtype=torch.bool, device=M.device)
    order = []
    for seed in torch.argsort(area, descending=True, stable=True).tolist():
        if not free[seed]:
            continue
        dist = (bits ^ bits[seed]).sum(1).float()
        dist[~free] = float('inf')
        pick = torch.topk(dist, min(32, int(free.sum())), largest=False).indices
        free[pick] = False
        order += pick.tolist()
    return np.asarray(order), bits.shape[1]


for blk in (4, 8):
    o, nb = greedy_bits(blk)
    L['greedy, %d-bit any-covered masks of %dx%d blocks, Hamming' % (nb, blk, blk)] = live(o)
    out['any_pose_of_a_128_block_greedy_bits_%d' % blk] = live(o, 128)
out['live_fraction_of_16_pose_tiles_grid_order'] = live(idx.reshape(-1), 16)
out['any_pose_of_a_128_block_grid_order'] = live(idx.reshape(-1), 128)
print(json.dumps(out, indent=1))
