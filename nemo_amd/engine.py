"""PoseEngine: torch-tensor front-end of libnemo_hip.so.

PyTorch is plumbing here (device memory, streams); every computation on the pose
path is a hand-written HIP kernel behind the C ABI of include/nemo_hip.h.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from .synthetic import adam_table, ndc_scale


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk(t, dtype, name, device):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError(f'{name}: expected a torch tensor')
    if t.device != device:
        raise ValueError(f'{name}: tensor on {t.device}, engine on {device}')
    if t.dtype != dtype:
        raise TypeError(f'{name}: expected {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise ValueError(f'{name}: tensor must be contiguous')
    return t


class PoseEngine:
    """One rasterizer/loss configuration + its registered meshes on one GPU.

    Mirrors what the reference builds at code/MeshPoseSolveAll.py:180-191,204:
    cameras(fov), RasterizationSettings(image_size=S, blur 0, K=1, naive) and the
    centre-crop to the feature-map shape.
    """

    def __init__(self, raster_size: int, map_shape: Sequence[int], feat_dim: int = 128,
                 fov: float = 12.0, perspective_correct: bool = False, clutter: bool = True,
                 k_eps: float = 1e-8, device: str | torch.device = 'cuda:0', crop_origin=None,
                 validate_indices: bool = False):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise _lib.NemoHipError('PoseEngine needs a GPU device; there is no CPU fallback')
        if self.device.index is None:  # 'cuda' -> the current device, so that tensors on cuda:k compare equal
            self.device = torch.device('cuda', torch.cuda.current_device())
        torch.cuda.set_device(self.device)
        # index tensors (img_idx / mesh_idx / grid_idx) are range-checked only on request: the check reads
        # them back, i.e. synchronises
        self.validate_indices = bool(validate_indices) or os.environ.get('NEMO_VALIDATE', '0') not in ('', '0')
        self.S = int(raster_size)
        self.h, self.w = int(map_shape[0]), int(map_shape[1])
        self.D = int(feat_dim)
        self.ndc_scale = ndc_scale(fov)
        self.clutter = bool(clutter)
        r0, c0 = (-1, -1) if crop_origin is None else crop_origin
        self.cfg = _lib.NemoConfig(self.S, self.h, self.w, r0, c0, self.D, self.ndc_scale,
                                   int(perspective_correct), int(clutter), float(k_eps))
        self.ctx = C.c_void_p()
        _lib.check(None, self.lib.nemo_ctx_create(C.byref(self.cfg), C.byref(self.ctx)), 'nemo_ctx_create')
        self._keep = []  # tensors registered with the library must outlive it
        self.n_meshes = 0
        self.steps = 0
        self._res_jobs = 0      # workspace reservation (nemo_workspace_reserve): jobs, jobs x grid poses
        self._res_cells = 0

    def close(self):
        if getattr(self, 'ctx', None) is not None and self.ctx:
            self.lib.nemo_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ setup
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def register_mesh(self, verts, faces, bank) -> int:
        """verts: PRE-PROCESSED [V,3] f32 (x,z,-y); faces [F,3] int; bank [V,D] f32."""
        v = torch.as_tensor(np.asarray(verts) if not torch.is_tensor(verts) else verts,
                            dtype=torch.float32).to(self.device).contiguous()
        f = torch.as_tensor(np.asarray(faces) if not torch.is_tensor(faces) else faces).to(
            device=self.device, dtype=torch.int32).contiguous()
        b = torch.as_tensor(np.asarray(bank) if not torch.is_tensor(bank) else bank,
                            dtype=torch.float32).to(self.device).contiguous()
        if b.shape != (v.shape[0], self.D):
            raise ValueError(f'bank shape {tuple(b.shape)} != ({v.shape[0]}, {self.D})')
        if int(f.min()) < 0 or int(f.max()) >= v.shape[0]:
            raise ValueError('face index outside the vertex array')
        mid = _lib.check(self.ctx, self.lib.nemo_mesh_register(self.ctx, _ptr(v), v.shape[0], _ptr(f),
                                                               f.shape[0], _ptr(b)), 'nemo_mesh_register')
        self._keep.append((v, f, b))
        self.n_meshes = mid + 1
        return mid

    def reserve(self, max_jobs: int, max_grid_poses: int = 0):
        """nemo_workspace_reserve: size the library's workspaces ONCE for batches of up to `max_jobs` jobs and
        grids of up to `max_grid_poses` poses; afterwards no hot call allocates.  The convenience wrappers below
        grow the reservation on demand (a set-up step the first time a larger batch shows up); callers that
        must not allocate in their loop (graph capture, serving) call this up front."""
        _lib.check(self.ctx, self.lib.nemo_workspace_reserve(self.ctx, int(max_jobs), int(max_grid_poses)),
                   'nemo_workspace_reserve')
        self._res_jobs = max(self._res_jobs, int(max_jobs))
        self._res_cells = max(self._res_cells, int(max_jobs) * int(max_grid_poses))

    def _need(self, n_jobs: int, G: int = 0):
        # the library grows each workspace on its own (launch order by jobs, partial sums by jobs x poses), so the
        # call's own (N, G) is what to ask for: max(N so far) x this G would over-allocate the partial sums
        if n_jobs > self._res_jobs or n_jobs * G > self._res_cells:
            self.reserve(n_jobs, G)

    def set_schedule(self, steps=300, lr=0.05, betas=(0.4, 0.6), eps=1e-8, decay_every=100, gamma=0.2):
        tab = np.ascontiguousarray(adam_table(steps, lr, betas[0], betas[1], decay_every, gamma))
        _lib.check(self.ctx, self.lib.nemo_set_schedule(self.ctx, tab.ctypes.data_as(C.c_void_p), steps,
                                                        betas[0], betas[1], eps), 'nemo_set_schedule')
        self.steps = steps
        return tab

    # ------------------------------------------------------------------ helpers
    def _f32(self, t, name):
        return _chk(t, torch.float32, name, self.device)

    def _i32(self, t, name):
        return _chk(t, torch.int32, name, self.device)

    def _empty(self, shape, dtype=torch.float32):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def _feat(self, t, name='feat'):
        """Pixel-major feature maps [n_img, h, w, D] (NOT the reference's [n, D, h, w]: see
        pose_solver.features_to_pixel_major)."""
        t = self._f32(t, name)
        if t.dim() != 4 or tuple(t.shape[1:]) != (self.h, self.w, self.D):
            raise ValueError(f'{name}: expected pixel-major [n_img, {self.h}, {self.w}, {self.D}], got {tuple(t.shape)}')
        return t

    def _clu(self, t, feat, name='clu'):
        t = self._f32(t, name)
        if t is not None and tuple(t.shape) != (feat.shape[0], self.h, self.w):
            raise ValueError(f'{name}: expected [{feat.shape[0]}, {self.h}, {self.w}], got {tuple(t.shape)}')
        return t

    def _jobs(self, N, feat=None, img_idx=None, mesh_idx=None, grid_idx=None, n_grids=None, **per_job):
        """Per-job arrays have N rows; index tensors stay inside what they index (values checked on request)."""
        for name, (t, tail) in per_job.items():
            if t is not None and tuple(t.shape) != (N,) + tuple(tail):
                raise ValueError(f'{name}: expected shape {(N,) + tuple(tail)}, got {tuple(t.shape)}')
        img_idx, mesh_idx, grid_idx = (self._i32(img_idx, 'img_idx'), self._i32(mesh_idx, 'mesh_idx'),
                                       self._i32(grid_idx, 'grid_idx'))
        for name, t in (('img_idx', img_idx), ('mesh_idx', mesh_idx), ('grid_idx', grid_idx)):
            if t is not None and tuple(t.shape) != (N,):
                raise ValueError(f'{name}: expected shape ({N},), got {tuple(t.shape)}')
        if img_idx is None and feat is not None and feat.shape[0] < N:
            raise ValueError(f'{N} jobs but only {feat.shape[0]} feature maps (pass img_idx to reuse images)')
        if mesh_idx is None and self.n_meshes < 1:
            raise ValueError('no mesh registered')
        if self.validate_indices:
            for name, t, hi in (('img_idx', img_idx, feat.shape[0] if feat is not None else None),
                                ('mesh_idx', mesh_idx, self.n_meshes), ('grid_idx', grid_idx, n_grids)):
                if t is not None and hi is not None and t.numel():
                    lo_v, hi_v = int(t.min()), int(t.max())
                    if lo_v < 0 or hi_v >= hi:
                        raise ValueError(f'{name}: values in [{lo_v}, {hi_v}] outside [0, {hi})')
        return img_idx, mesh_idx, grid_idx

    # ------------------------------------------------------------------ hot path
    def campos_to_R_T(self, C_, theta, extra=None):
        C_, theta = self._f32(C_, 'C'), self._f32(theta, 'theta')
        N = C_.shape[0]
        R, T = self._empty((N, 3, 3)), self._empty((N, 3))
        _lib.check(self.ctx, self.lib.nemo_campos_to_R_T(self.ctx, _ptr(C_), _ptr(theta),
                                                         _ptr(self._f32(extra, 'extra')), N, _ptr(R),
                                                         _ptr(T), self._stream()), 'nemo_campos_to_R_T')
        return R, T

    def rasterize(self, R, T, mesh_idx=None, want_zbuf=False):
        R, T = self._f32(R, 'R'), self._f32(T, 'T')
        N = R.shape[0]
        _, mesh_idx, _ = self._jobs(N, mesh_idx=mesh_idx, R=(R.reshape(N, -1), (9,)), T=(T, (3,)))
        self._need(N)
        p2f = self._empty((N, self.h, self.w), torch.int32)
        bary = self._empty((N, self.h, self.w, 3))
        zb = self._empty((N, self.h, self.w)) if want_zbuf else None
        _lib.check(self.ctx, self.lib.nemo_rasterize(self.ctx, _ptr(self._i32(mesh_idx, 'mesh_idx')), _ptr(R),
                                                     _ptr(T), N, _ptr(p2f), _ptr(bary), _ptr(zb),
                                                     self._stream()), 'nemo_rasterize')
        return (p2f, bary, zb) if want_zbuf else (p2f, bary)

    def interpolate(self, p2f, bary, mesh_idx=None):
        p2f, bary = self._i32(p2f, 'pix_to_face'), self._f32(bary, 'bary')
        N = p2f.shape[0]
        out = self._empty((N, self.h, self.w, self.D))
        _lib.check(self.ctx, self.lib.nemo_interpolate(self.ctx, _ptr(self._i32(mesh_idx, 'mesh_idx')), _ptr(p2f),
                                                       _ptr(bary), N, _ptr(out), self._stream()),
                   'nemo_interpolate')
        return out

    def interpolate_backward(self, R, T, grad_out, mesh_idx=None):
        R, T, grad_out = self._f32(R, 'R'), self._f32(T, 'T'), self._feat(grad_out, 'grad_out')
        N = R.shape[0]
        _, mesh_idx, _ = self._jobs(N, feat=grad_out, mesh_idx=mesh_idx, R=(R.reshape(N, -1), (9,)), T=(T, (3,)))
        self._need(N)
        gR, gT = self._empty((N, 3, 3)), self._empty((N, 3))
        _lib.check(self.ctx, self.lib.nemo_interpolate_backward(
            self.ctx, _ptr(self._i32(mesh_idx, 'mesh_idx')), _ptr(R), _ptr(T), _ptr(grad_out), N, _ptr(gR),
            _ptr(gT), self._stream()), 'nemo_interpolate_backward')
        return gR, gT

    def clutter_score(self, feat, cbank):
        feat, cbank = self._feat(feat), self._f32(cbank, 'cbank')
        if cbank.dim() == 1:
            cbank = cbank[None]
        if cbank.dim() != 2 or cbank.shape[1] != self.D:
            raise ValueError(f'cbank: expected [K, {self.D}], got {tuple(cbank.shape)}')
        n_img = feat.shape[0]
        out = self._empty((n_img, self.h, self.w))
        _lib.check(self.ctx, self.lib.nemo_clutter_score(self.ctx, _ptr(feat), n_img, _ptr(cbank), cbank.shape[0],
                                                         _ptr(out), self._stream()), 'nemo_clutter_score')
        return out

    def feature_head(self, x, weight, cbank=None):
        """Backbone tail (KeypointRepresentationNet.py:217-223) fused with the clutter score: x [N,C,h,w] f32 ->
        (features [N,h,w,D] pixel-major, clutter score [N,h,w] or None)."""
        x, weight = self._f32(x, 'x'), self._f32(weight, 'weight')
        N, Cin = x.shape[0], x.shape[1]
        if tuple(x.shape[2:]) != (self.h, self.w) or tuple(weight.shape) != (self.D, Cin):
            raise ValueError('feature_head: x %s / weight %s do not match the %dx%d map and D=%d' %
                             (tuple(x.shape), tuple(weight.shape), self.h, self.w, self.D))
        feat = self._empty((N, self.h, self.w, self.D))
        cb = self._f32(cbank, 'cbank') if cbank is not None else None
        clu = self._empty((N, self.h, self.w)) if cb is not None else None
        _lib.check(self.ctx, self.lib.nemo_feature_head(self.ctx, _ptr(x), N, Cin, _ptr(weight), _ptr(cb),
                                                        cb.shape[0] if cb is not None else 0, _ptr(feat), _ptr(clu),
                                                        self._stream()), 'nemo_feature_head')
        return feat, clu

    def score_fwd_bwd(self, feat, clu, C_, theta, img_idx=None, mesh_idx=None, want_grad=True):
        C_, theta = self._f32(C_, 'C'), self._f32(theta, 'theta')
        N = C_.shape[0]
        feat = self._feat(feat)
        clu = self._clu(clu, feat)
        img_idx, mesh_idx, _ = self._jobs(N, feat, img_idx, mesh_idx, C=(C_, (3,)), theta=(theta, ()))
        self._need(N)
        loss = self._empty((N,))
        grad = self._empty((N, 4)) if want_grad else None
        ncov = self._empty((N,), torch.int32)
        _lib.check(self.ctx, self.lib.nemo_score_fwd_bwd(
            self.ctx, N, _ptr(img_idx), _ptr(mesh_idx),
            _ptr(feat), _ptr(clu), _ptr(C_), _ptr(theta), _ptr(loss),
            _ptr(grad), _ptr(ncov), self._stream()), 'nemo_score_fwd_bwd')
        return loss, grad, ncov

    def grid_search(self, feat, clu, grid_C, grid_theta, n_jobs=None, img_idx=None, mesh_idx=None,
                    grid_idx=None, losses=None):
        grid_C, grid_theta = self._f32(grid_C, 'grid_C'), self._f32(grid_theta, 'grid_theta')
        G = grid_theta.shape[-1]
        n_grids = grid_theta.shape[0] if grid_theta.dim() > 1 else 1
        if tuple(grid_C.shape[-2:]) != (G, 3) or grid_C.numel() != n_grids * G * 3:
            raise ValueError(f'grid_C {tuple(grid_C.shape)} does not match grid_theta {tuple(grid_theta.shape)}')
        N = n_jobs if n_jobs is not None else (img_idx.shape[0] if img_idx is not None else feat.shape[0])
        feat = self._feat(feat)
        clu = self._clu(clu, feat)
        img_idx, mesh_idx, grid_idx = self._jobs(N, feat, img_idx, mesh_idx, grid_idx, n_grids)
        self._need(N, G)
        if losses is None:
            losses = self._empty((N, G))
        elif tuple(losses.shape) != (N, G):
            raise ValueError(f'losses: expected [{N}, {G}], got {tuple(losses.shape)}')
        C0, th0 = self._empty((N, 3)), self._empty((N,))
        best = self._empty((N,), torch.int32)
        _lib.check(self.ctx, self.lib.nemo_grid_search(
            self.ctx, N, _ptr(img_idx), _ptr(mesh_idx),
            _ptr(grid_idx), _ptr(feat), _ptr(clu),
            _ptr(grid_C), _ptr(grid_theta), G, _ptr(self._f32(losses, 'losses')), _ptr(C0), _ptr(th0),
            _ptr(best), self._stream()), 'nemo_grid_search')
        return C0, th0, best, losses

    def grid_prerender(self, grid_C, grid_theta, pairs):
        """get_pre_render_samples (code/MeshPoseSolveAll.py:62-82): render the coverage of every grid pose
        once per (mesh, grid) pair; later grid_search calls with the SAME grid tensors reuse it.
        `pairs` = iterable of (mesh id, grid id); an empty list drops the store."""
        import numpy as np
        pairs = [(int(m), int(g)) for m, g in pairs]
        pm = np.ascontiguousarray([p[0] for p in pairs], dtype=np.int32)
        pg = np.ascontiguousarray([p[1] for p in pairs], dtype=np.int32)
        if not pairs:
            _lib.check(self.ctx, self.lib.nemo_grid_prerender(self.ctx, None, None, 0, 0, None, None, 0, self._stream()),
                       'nemo_grid_prerender')
            self._pre_keep = None
            return
        grid_C, grid_theta = self._f32(grid_C, 'grid_C'), self._f32(grid_theta, 'grid_theta')
        n_grids, G = (grid_theta.shape[0], grid_theta.shape[-1]) if grid_theta.dim() > 1 else (1, grid_theta.shape[-1])
        _lib.check(self.ctx, self.lib.nemo_grid_prerender(self.ctx, _ptr(grid_C), _ptr(grid_theta), n_grids, G,
                                                          pm.ctypes.data, pg.ctypes.data, len(pairs), self._stream()),
                   'nemo_grid_prerender')
        self._pre_keep = (grid_C, grid_theta)  # the store is keyed by these device buffers: keep them alive

    def solve_table_capacity(self):
        """Jobs per call the reserved dot-product tables hold (nemo_solve_table_capacity); 0 = the solve streams."""
        return int(self.lib.nemo_solve_table_capacity(self.ctx))

    def solve_last_timing(self):
        """(table_ms, loop_ms) of the last tables-form solve (nemo_solve_last_timing; synchronises); None if it streamed."""
        import ctypes as C
        a, b = C.c_float(), C.c_float()
        if self.lib.nemo_solve_last_timing(self.ctx, C.byref(a), C.byref(b)) != 0:
            return None
        return float(a.value), float(b.value)

    def solve_table_stats(self):
        """(mode, blocks): form of the last solve (0 streaming, 1 whole tables, 2 tables on demand) and the table blocks
        computed on demand since counter_reset (nemo_solve_table_stats; synchronises)."""
        import ctypes as C
        m, b = C.c_int32(), C.c_uint64()
        _lib.check(self.ctx, self.lib.nemo_solve_table_stats(self.ctx, C.byref(m), C.byref(b)), 'nemo_solve_table_stats')
        return int(m.value), int(b.value)

    def solve_uses_tables(self, n_jobs=1):
        return self.solve_table_capacity() >= max(1, int(n_jobs))

    GRID_STORE_KINDS = ('none', 'dense_maps', 'pose_blocks', 'records')

    def grid_store_kind(self):
        """Which operand of the matrix-core grid search the context holds (nemo_grid_store_kind)."""
        return self.GRID_STORE_KINDS[self.lib.nemo_grid_store_kind(self.ctx)]

    def solve(self, feat, clu, C0, theta0, img_idx=None, mesh_idx=None, want_trace=False, out=None, static=False,
              status=None):
        """nemo_solve.  `out` = (pose [N,4], loss [N]) buffers to fill instead of fresh ones; static=True goes
        through nemo_solve_static (launch-order workspace in the library's static device array: nothing is
        allocated by the library; with `out` given, nothing by this wrapper either).  `status` = int32 [N] buffer
        for the per-job flag (1 = finite final pose, 0 = not), or None."""
        C0, theta0 = self._f32(C0, 'C0'), self._f32(theta0, 'theta0')
        N = C0.shape[0]
        feat = self._feat(feat)
        clu = self._clu(clu, feat)
        img_idx, mesh_idx, _ = self._jobs(N, feat, img_idx, mesh_idx, C0=(C0, (3,)), theta0=(theta0, ()))
        if not static:
            self._need(N)
        pose, loss = out if out is not None else (self._empty((N, 4)), self._empty((N,)))
        if tuple(pose.shape) != (N, 4) or tuple(loss.shape) != (N,):
            raise ValueError('out: expected (pose [N,4], loss [N])')
        pose, loss = self._f32(pose, 'pose'), self._f32(loss, 'loss')
        trace = self._empty((N, self.steps, 9)) if want_trace else None
        if status is not None and tuple(self._i32(status, 'status').shape) != (N,):
            raise ValueError(f'status: expected int32 [{N}]')
        fn, name = (self.lib.nemo_solve_static, 'nemo_solve_static') if static else (self.lib.nemo_solve, 'nemo_solve')
        _lib.check(self.ctx, fn(
            self.ctx, N, _ptr(img_idx), _ptr(mesh_idx),
            _ptr(feat), _ptr(clu), _ptr(C0), _ptr(theta0), _ptr(pose),
            _ptr(loss), _ptr(trace), _ptr(status), self._stream()), name)
        return (pose, loss, trace) if want_trace else (pose, loss)

    def solve6(self, feat, clu, C0, theta0, trans0=None, img_idx=None, mesh_idx=None, want_trace=False, status=None):
        """6-DoF solve (code/PredUnlabeledDataset.py:318-347): returns pose6 [N,6] = (C, theta, trans_2d), loss
        (and the [N,steps,13] trace); `status` as in solve()."""
        C0, theta0 = self._f32(C0, 'C0'), self._f32(theta0, 'theta0')
        N = C0.shape[0]
        feat = self._feat(feat)
        clu = self._clu(clu, feat)
        trans0 = self._f32(trans0, 'trans0')
        img_idx, mesh_idx, _ = self._jobs(N, feat, img_idx, mesh_idx, C0=(C0, (3,)), theta0=(theta0, ()),
                                          trans0=(trans0, (2,)))
        self._need(N)
        pose, loss = self._empty((N, 6)), self._empty((N,))
        trace = self._empty((N, self.steps, 13)) if want_trace else None
        if status is not None and tuple(self._i32(status, 'status').shape) != (N,):
            raise ValueError(f'status: expected int32 [{N}]')
        _lib.check(self.ctx, self.lib.nemo_solve6(
            self.ctx, N, _ptr(img_idx), _ptr(mesh_idx),
            _ptr(feat), _ptr(clu), _ptr(C0), _ptr(theta0),
            _ptr(trans0), _ptr(pose), _ptr(loss), _ptr(trace), _ptr(status), self._stream()), 'nemo_solve6')
        return (pose, loss, trace) if want_trace else (pose, loss)

    # ------------------------------------------------------------------ accounting
    def bytes_per_covered_pixel(self) -> int:
        return int(self.lib.nemo_bytes_per_covered_pixel(self.ctx))

    def counter_reset(self):
        _lib.check(self.ctx, self.lib.nemo_counter_reset(self.ctx, self._stream()), 'nemo_counter_reset')

    def counter_read(self) -> int:
        v = C.c_uint64()
        _lib.check(self.ctx, self.lib.nemo_counter_read(self.ctx, C.byref(v)), 'nemo_counter_read')
        return int(v.value)
