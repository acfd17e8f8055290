"""A second, independently written restatement of PyTorch3D's naive rasterizer (K = 1, blur_radius = 0):
float32 numpy array operations over ALL pixels x ALL faces, no per-pixel loop, no shared code with
oracle/nemo_oracle.c.

TEST INFRASTRUCTURE ONLY (never imported by nemo_amd/).  PARITY UNPINNED by the reference like the C oracle:
PyTorch3D is neither in /root/reference nor installable here.  What this file adds is a second lineage for the
DISCRETE half of the path -- which face owns which pixel -- written from the published algorithm as SURVEY.md
section 8(a) row P4 and DESIGN.md section 5 state it, so that the bounding-box test, kEps handling, strict
inequalities, tie rule, pixel -> NDC flip and crop origin of the C oracle (and through it of the HIP kernels) are
cross-checked by something that was not copied from them.  tests/test_oracle_cpu.py requires the two to agree
bit for bit.

Semantics restated (call sites in the reference: code/lib/MeshUtils.py:83-85 via MeshRasterizer, settings at
code/MeshPoseSolveAll.py:181-186):
  * output pixel (row i, column j) of the S x S image has the NDC centre
        x = -1 + (2 (S - 1 - j) + 1) / S,   y = -1 + (2 (S - 1 - i) + 1) / S        (+x left, +y up);
  * edge(p; a, b) = (p.x - a.x) (b.y - a.y) - (p.y - a.y) (b.x - a.x), every product and difference rounded to float32;
  * a face is skipped when |edge(v0; v1, v2)| <= kEps, and at pixels outside its closed xy bounding box;
  * area = float32(float64(edge(v2; v0, v1)) + kEps);  w0 = edge(p; v1, v2) / area, w1 = edge(p; v2, v0) / area,
    w2 = edge(p; v0, v1) / area;
  * perspective_correct: t0 = (w0 z1) z2, t1 = (z0 w1) z2, t2 = (z0 z1) w2, w_k = t_k / ((t0 + t1) + t2);
  * pz = (w0 z0 + w1 z1) + w2 z2; fragments with pz < 0 or any w_k <= 0 are dropped;
  * per pixel the fragment with the smallest pz wins, the LOWEST face index among equal pz;
  * empty pixels: pix_to_face = -1, bary = -1, zbuf = -1.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def pixel_centres(S: int) -> np.ndarray:
    """NDC coordinate of pixel index 0..S-1 along either axis (index 0 = left / top = +1 side)."""
    k = (S - 1 - np.arange(S)).astype(f32)
    return (f32(-1.0) + (f32(2.0) * k + f32(1.0)) / f32(S)).astype(f32)


def _edge(px, py, ax, ay, bx, by):
    return ((px - ax) * (by - ay) - (py - ay) * (bx - ax)).astype(f32)


def rasterize(screen_verts, faces, S: int, perspective_correct: bool = False, k_eps: float = 1e-8, chunk: int = 128):
    """screen_verts [V,3] f32 = (x_ndc, y_ndc, z_view); faces [F,3] int.  Returns pix_to_face [S,S] int32,
    bary [S,S,3] f32, zbuf [S,S] f32 of the full square image."""
    sv = np.ascontiguousarray(screen_verts, dtype=f32)
    fc = np.asarray(faces).astype(np.int64)
    c = pixel_centres(S)
    PX = np.broadcast_to(c[None, :], (S, S)).reshape(-1, 1)   # column j -> x
    PY = np.broadcast_to(c[:, None], (S, S)).reshape(-1, 1)   # row i    -> y
    n = S * S
    best_z = np.full(n, np.inf, dtype=f32)
    best_f = np.full(n, -1, dtype=np.int32)
    best_b = np.full((n, 3), -1.0, dtype=f32)
    with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
        for s in range(0, len(fc), chunk):
            f = fc[s:s + chunk]
            v0, v1, v2 = sv[f[:, 0]], sv[f[:, 1]], sv[f[:, 2]]
            x0, y0, z0 = v0[None, :, 0], v0[None, :, 1], v0[None, :, 2]
            x1, y1, z1 = v1[None, :, 0], v1[None, :, 1], v1[None, :, 2]
            x2, y2, z2 = v2[None, :, 0], v2[None, :, 1], v2[None, :, 2]
            face_area = _edge(x0, y0, x1, y1, x2, y2).astype(np.float64)
            usable = ~((face_area <= k_eps) & (face_area >= -k_eps))                    # [1,F]
            xmin, xmax = np.minimum(np.minimum(x0, x1), x2), np.maximum(np.maximum(x0, x1), x2)
            ymin, ymax = np.minimum(np.minimum(y0, y1), y2), np.maximum(np.maximum(y0, y1), y2)
            in_box = (PX >= xmin) & (PX <= xmax) & (PY >= ymin) & (PY <= ymax)           # [n,F]
            area = (_edge(x2, y2, x0, y0, x1, y1).astype(np.float64) + k_eps).astype(f32)
            w0 = (_edge(PX, PY, x1, y1, x2, y2) / area).astype(f32)
            w1 = (_edge(PX, PY, x2, y2, x0, y0) / area).astype(f32)
            w2 = (_edge(PX, PY, x0, y0, x1, y1) / area).astype(f32)
            if perspective_correct:
                t0 = ((w0 * z1) * z2).astype(f32)
                t1 = ((z0 * w1) * z2).astype(f32)
                t2 = ((z0 * z1) * w2).astype(f32)
                den = ((t0 + t1) + t2).astype(f32)
                w0, w1, w2 = (t0 / den).astype(f32), (t1 / den).astype(f32), (t2 / den).astype(f32)
            pz = ((w0 * z0 + w1 * z1) + w2 * z2).astype(f32)
            ok = usable & in_box & ~(pz < 0) & (w0 > 0) & (w1 > 0) & (w2 > 0)
            pz_ok = np.where(ok, pz + f32(0.0), f32(np.inf))
            k = np.argmin(pz_ok, axis=1)                       # first minimum = lowest face index inside the chunk
            rows = np.arange(n)
            zk = pz_ok[rows, k]
            better = zk < best_z                               # strict: an earlier chunk keeps a tie
            best_z = np.where(better, zk, best_z)
            best_f = np.where(better, (s + k).astype(np.int32), best_f)
            bk = np.stack([w0[rows, k], w1[rows, k], w2[rows, k]], axis=1)
            best_b = np.where(better[:, None], bk, best_b)
    zbuf = np.where(best_f >= 0, best_z, f32(-1.0)).astype(f32)
    return best_f.reshape(S, S), best_b.reshape(S, S, 3), zbuf.reshape(S, S)


def crop(a, S, h, w, r0=None, c0=None):
    """center_crop_fun(map_shape, (S, S)) of the reference (code/lib/MeshUtils.py:148-150): a box of the output
    shape centred at (S // 2, S // 2)."""
    r0 = S // 2 - h // 2 if r0 is None else r0
    c0 = S // 2 - w // 2 if c0 is None else c0
    return a[r0:r0 + h, c0:c0 + w]
