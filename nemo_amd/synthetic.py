"""Seeded synthetic workloads for the pose solver.

There are no datasets or checkpoints offline, so every test and benchmark runs on
synthetic data of the reference's shapes:

* cuboid meshes follow the construction of the reference's mesh generator
  (reference tools/CreateCuboidMesh.py:33-108, ``meshelize``): six grids of
  vertices, two triangles per grid cell, faces ordered z-min, z-max, y-min,
  y-max, x-min, x-max;
* vertex / clutter banks are L2-normalised Gaussian rows (the reference's banks
  are unit-norm rows, code/models/FeatureBanks.py:153-163);
* feature maps are a noisy render of the mesh at a ground-truth pose on object
  pixels and unit-norm noise elsewhere.

Nothing here touches the GPU or the oracle; the render used to build feature
maps is passed in by the caller.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

# Per-category constants of the reference solver (code/MeshPoseSolveAll.py:146-151).
CATEGORIES = ['aeroplane', 'bicycle', 'boat', 'bottle', 'bus', 'car', 'chair', 'diningtable',
              'motorbike', 'sofa', 'train', 'tvmonitor']
DISTANCE_RENDER = {'car': 5, 'bus': 5.2, 'motorbike': 4.5, 'bottle': 5.75, 'boat': 8, 'bicycle': 5.2,
                   'aeroplane': 7, 'sofa': 5.4, 'tvmonitor': 5.5, 'chair': 4, 'diningtable': 7,
                   'train': 3.75}
IMAGE_SIZES = {'car': (256, 672), 'bus': (320, 800), 'motorbike': (512, 512), 'boat': (480, 1120),
               'bicycle': (608, 608), 'aeroplane': (320, 1024), 'sofa': (352, 736),
               'tvmonitor': (480, 480), 'chair': (544, 384), 'diningtable': (320, 800),
               'bottle': (512, 736), 'train': (256, 608)}
# Synthetic cuboid extents (x, y, z) in CAD units; the real ones come from the
# PASCAL3D+ CAD models, which are not available offline.  'car' is the survey's
# car-like box (V=1088, F=1844 at number_vertices=1000).
EXTENTS = {'aeroplane': (1.00, 0.90, 0.28), 'bicycle': (0.24, 0.96, 0.58), 'boat': (0.42, 1.30, 0.40),
           'bottle': (0.26, 0.26, 0.90), 'bus': (0.34, 1.00, 0.40), 'car': (0.40, 0.94, 0.30),
           'chair': (0.42, 0.44, 0.74), 'diningtable': (0.86, 1.20, 0.52), 'motorbike': (0.28, 0.90, 0.52),
           'sofa': (0.90, 0.50, 0.44), 'train': (0.22, 0.90, 0.28), 'tvmonitor': (0.80, 0.22, 0.66)}
# MultiCuboid subtype counts per class (reference code/dataset/kp_list.py:14).
MESH_LEN = {'aeroplane': 8, 'bicycle': 6, 'boat': 6, 'bottle': 8, 'bus': 6, 'car': 10, 'chair': 10,
            'diningtable': 6, 'motorbike': 5, 'sofa': 6, 'train': 4, 'tvmonitor': 4}


def _grid_faces(base, n_outer, n_inner):
    """Two triangles per cell of an (n_outer x n_inner) vertex grid, cell-major."""
    m = np.arange(n_outer - 1)[:, None]
    n = np.arange(n_inner - 1)[None, :]
    a = base + m * n_inner + n
    b = a + 1
    c = base + (m + 1) * n_inner + n
    d = c + 1
    tri = np.stack([np.stack([a, b, c], -1), np.stack([d, b, c], -1)], axis=2)
    return tri.reshape(-1, 3)


def cuboid_mesh(x_range, y_range, z_range, number_vertices=1000):
    """Cuboid surface mesh with ~number_vertices vertices.

    Same vertex and face order as the reference generator
    (tools/CreateCuboidMesh.py:33-108), so meshes written by either are
    interchangeable.  Returns (verts float64 [V,3], faces int64 [F,3]).
    """
    w, h, d = x_range[1] - x_range[0], y_range[1] - y_range[0], z_range[1] - z_range[0]
    total_area = (w * h + h * d + w * d) * 2
    cell = total_area / (number_vertices * 2)
    edge = (cell * 2) ** .5
    xs = x_range[0] + np.linspace(0, w, int(w / edge + 1))
    ys = y_range[0] + np.linspace(0, h, int(h / edge + 1))
    zs = z_range[0] + np.linspace(0, d, int(d / edge + 1))
    xn, yn, zn = xs.size, ys.size, zs.size

    verts, faces, base = [], [], 0

    def side(outer, inner, make):
        nonlocal base
        o, i = np.meshgrid(outer, inner, indexing='ij')
        verts.append(make(o.ravel(), i.ravel()))
        faces.append(_grid_faces(base, outer.size, inner.size))
        base += outer.size * inner.size

    side(ys, xs, lambda o, i: np.stack([i, o, np.full_like(o, zs[0])], 1))
    side(ys, xs, lambda o, i: np.stack([i, o, np.full_like(o, zs[-1])], 1))
    side(zs, xs, lambda o, i: np.stack([i, np.full_like(o, ys[0]), o], 1))
    side(zs, xs, lambda o, i: np.stack([i, np.full_like(o, ys[-1]), o], 1))
    side(zs, ys, lambda o, i: np.stack([np.full_like(o, xs[0]), i, o], 1))
    side(zs, ys, lambda o, i: np.stack([np.full_like(o, xs[-1]), i, o], 1))
    del xn, yn, zn
    return np.concatenate(verts, 0), np.concatenate(faces, 0)


def class_mesh(cate='car', number_vertices=1000, jitter=None):
    """float32 vertices (PASCAL3D+ axes) and int32 faces of a class cuboid."""
    ex = np.asarray(EXTENTS[cate], dtype=np.float64)
    if jitter is not None:
        ex = ex * jitter
    v, f = cuboid_mesh((-ex[0] / 2, ex[0] / 2), (-ex[1] / 2, ex[1] / 2), (-ex[2] / 2, ex[2] / 2),
                       number_vertices)
    return v.astype(np.float32), f.astype(np.int32)


def unit_rows(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)


def spherical_to_campos(dist, elev, azim):
    """camera_position_from_spherical_angles, radians, float32."""
    dist, elev, azim = (np.asarray(a, dtype=np.float32) for a in (dist, elev, azim))
    x = dist * np.cos(elev) * np.sin(azim)
    y = dist * np.sin(elev)
    z = dist * np.cos(elev) * np.cos(azim)
    return np.stack(np.broadcast_arrays(x, y, z), -1).astype(np.float32)


def pose_grid(distance, n_azim=12, n_elev=4, n_theta=3, distances=None, thetas=None):
    """The reference's init grid (code/MeshPoseSolveAll.py:193-195, :66): azimuth-major,
    then elevation, then theta; optional innermost distance axis and explicit theta samples
    (PredUnlabeledDataset.py:153-154, :283-286: theta_sample = [0], dist_sample = [4.4, 5, 5.7])."""
    az = np.linspace(0, np.pi * 2, n_azim, endpoint=False)
    el = np.linspace(-np.pi / 6, np.pi / 3, n_elev)
    th = np.linspace(-np.pi / 6, np.pi / 6, n_theta) if thetas is None else np.asarray(thetas, dtype=np.float64)
    ds = [distance] if distances is None else list(distances)
    samples = [(a, e, t, d) for a in az for e in el for t in th for d in ds]
    s = np.asarray(samples, dtype=np.float64)
    C = spherical_to_campos(s[:, 3], s[:, 1], s[:, 0])
    return C, s[:, 2].astype(np.float32)


@dataclass
class Geometry:
    """Raster / window geometry of one category (code/MeshPoseSolveAll.py:171,191)."""
    S: int
    h: int
    w: int
    r0: int
    c0: int

    @staticmethod
    def from_image_size(image_hw, stride=8):
        H, W = image_hw
        S = max(H, W) // stride
        h, w = H // stride, W // stride
        # BboxTools box_by_shape((h,w), centre=(S//2,S//2)): rows [cy - h//2, cy - h//2 + h)
        return Geometry(S, h, w, S // 2 - h // 2, S // 2 - w // 2)


def ndc_scale(fov_deg=12.0):
    """s = 1/tan(fov/2) evaluated in float32 the way the OpenGL projection matrix builds it
    (SURVEY P3): fov -> radians, tan(fov/2), 2*znear/(right-left) with znear=1."""
    fov = np.float32(np.pi / 180) * np.float32(fov_deg)
    t = np.tan(fov / np.float32(2))
    top = np.float32(t) * np.float32(1.0)
    return float(np.float32(2.0) * np.float32(1.0) / (top - (-top)))


def adam_table(steps, lr=0.05, b1=0.4, b2=0.6, decay_every=100, gamma=0.2):
    """Per-step Adam scalars prepared in double like the Python side of torch.optim.Adam and
    ExponentialLR.step every `decay_every` steps (code/MeshPoseSolveAll.py:230-231,249-250):
    tab[t] = (lr_t / (1 - b1^(t+1)), sqrt(1 - b2^(t+1)))."""
    tab = np.zeros((steps, 2), dtype=np.float32)
    cur = float(lr)
    for t in range(steps):
        step = t + 1
        tab[t, 0] = cur / (1.0 - b1 ** step)
        tab[t, 1] = math.sqrt(1.0 - b2 ** step)
        if step % decay_every == 0:
            cur = cur * gamma
    return tab


def random_gt_poses(rng, n, distance):
    az = rng.uniform(0, 2 * np.pi, n)
    el = rng.uniform(-np.pi / 6, np.pi / 3, n)
    th = rng.uniform(-np.pi / 6, np.pi / 6, n)
    ds = distance * rng.uniform(0.9, 1.1, n)
    return az, el, th, ds


def make_feature_maps(rng, rendered, covered, sigma=0.5):
    """rendered [N,h,w,D] (zeros off-object), covered [N,h,w] bool -> L2-normalised maps:
    normalise(render + sigma*noise) on object pixels, unit-norm noise elsewhere."""
    noise = rng.standard_normal(rendered.shape).astype(np.float32)
    x = np.where(covered[..., None], rendered + np.float32(sigma) * noise, noise)
    n = np.linalg.norm(x, axis=-1, keepdims=True)
    return (x / np.maximum(n, 1e-12)).astype(np.float32)
