"""Loss-landscape scans (tools/PlotLossLandscape.py:111-169): 1-D sweeps of azimuth, elevation, theta and
distance around an annotated pose, loss = 1 - mean(object score) (no clutter term, :127).

The reference renders the 284 poses of one image one after the other through MeshInterpolateModule; here
every (image, pose) of a batch is one job of a single `nemo_score_fwd_bwd` launch (forward only).
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import synthetic as syn

AZIMUTH_SHIFTS = np.linspace(-3.14, 3.14, 121)          # :111
ELEVATION_SHIFTS = np.linspace(-3.14 / 2, 3.14 / 2, 61)  # :112
THETA_SHIFTS = np.linspace(-3.14 / 2, 3.14 / 2, 61)      # :113
DISTANCE_SHIFTS = np.linspace(-2, 2, 41)                 # :114


def scan_poses(azimuth, elevation, theta, distance, az_shifts=AZIMUTH_SHIFTS, el_shifts=ELEVATION_SHIFTS,
               th_shifts=THETA_SHIFTS, d_shifts=DISTANCE_SHIFTS):
    """The pose list of one image's four scans, in the reference's order: (C [n,3] f64, theta [n] f32, slices)."""
    az, el, th, ds = [], [], [], []
    for s in az_shifts:                                   # :118-121: azimuth wrapped to [0, 2 pi)
        az.append((azimuth + s + 2 * np.pi) % (2 * np.pi)); el.append(elevation); th.append(theta); ds.append(distance)
    for s in el_shifts:                                   # :131-134
        az.append(azimuth); el.append(elevation + s); th.append(theta); ds.append(distance)
    for s in th_shifts:                                   # :144-148
        az.append(azimuth); el.append(elevation); th.append(theta + s); ds.append(distance)
    for s in d_shifts:                                    # :158-162
        az.append(azimuth); el.append(elevation); th.append(theta); ds.append(distance + s)
    n = [len(az_shifts), len(el_shifts), len(th_shifts), len(d_shifts)]
    o = np.cumsum([0] + n)
    sl = {k: slice(int(o[i]), int(o[i + 1])) for i, k in enumerate(('azimuth', 'elevation', 'theta', 'distance'))}
    C = syn.spherical_to_campos(np.asarray(ds), np.asarray(el), np.asarray(az))
    return C, np.asarray(th, dtype=np.float32), sl


def scan_losses(engine, feats: torch.Tensor, annos: Sequence[Sequence[float]], distance: float,
                mesh_idx: Optional[Sequence[int]] = None, **shifts) -> Dict[str, np.ndarray]:
    """`feats` [n_img,h,w,D] on the engine's device; `annos[i]` = (azimuth, elevation, theta) of image i
    (PASCAL3D+ annotation, :118-147); `distance` the category's render distance (:120).  The engine must have
    been created with clutter=False (the scan's loss has no clutter term).  Returns the four scans as
    [n_img, n_shift] arrays of `1 - mean(sim)`."""
    if engine.clutter:
        raise ValueError('scan_losses needs an engine created with clutter=False (loss = 1 - mean(sim))')
    Cs, ths, imgs, meshes, sl = [], [], [], [], None
    for i, (az, el, th) in enumerate(annos):
        C, t, sl = scan_poses(float(az), float(el), float(th), float(distance), **shifts)
        Cs.append(C); ths.append(t)
        imgs.append(np.full(len(t), i, np.int32))
        meshes.append(np.full(len(t), 0 if mesh_idx is None else int(mesh_idx[i]), np.int32))
    dev = engine.device
    C = torch.as_tensor(np.concatenate(Cs), dtype=torch.float32, device=dev)
    th = torch.as_tensor(np.concatenate(ths), dtype=torch.float32, device=dev)
    loss, _, _ = engine.score_fwd_bwd(feats, None, C, th, img_idx=torch.as_tensor(np.concatenate(imgs), device=dev),
                                      mesh_idx=torch.as_tensor(np.concatenate(meshes), device=dev), want_grad=False)
    loss = loss.cpu().numpy().reshape(len(annos), -1)
    return {k: loss[:, s] for k, s in sl.items()}
