"""ctypes front-end of the CPU oracle (oracle/nemo_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package nemo_amd/.
PARITY UNPINNED (see the header of nemo_oracle.c): the reference has no golden
vectors for this path and PyTorch3D cannot run here.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libnemo_oracle.so')


class Cfg(C.Structure):
    _fields_ = [('S', C.c_int), ('h', C.c_int), ('w', C.c_int), ('r0', C.c_int), ('c0', C.c_int),
                ('D', C.c_int), ('ndc_scale', C.c_float), ('perspective_correct', C.c_int),
                ('clutter', C.c_int), ('k_eps', C.c_double)]


class EvalOut(C.Structure):
    _fields_ = [('loss', C.c_float), ('gC', C.c_float * 3), ('gtheta', C.c_float),
                ('gextra', C.c_float * 3), ('ncov', C.c_int)]


def build(force=False):
    src = os.path.join(_HERE, 'nemo_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s', 'CC=gcc'])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def make_cfg(S, h, w, D, ndc_scale, r0=None, c0=None, perspective_correct=False, clutter=True,
             k_eps=1e-8):
    r0 = S // 2 - h // 2 if r0 is None else r0
    c0 = S // 2 - w // 2 if c0 is None else c0
    return Cfg(S, h, w, r0, c0, D, ndc_scale, int(perspective_correct), int(clutter), k_eps)


def sincos(x):
    s, c = C.c_float(), C.c_float()
    lib().orc_sincosf(C.c_float(float(x)), C.byref(s), C.byref(c))
    return s.value, c.value


def campos_to_R_T(Cpos, theta, extra=None):
    Cpos = _f(Cpos).reshape(3)
    R, T = np.zeros(9, np.float32), np.zeros(3, np.float32)
    ex = _f(extra) if extra is not None else None
    lib().orc_campos_to_R_T(_p(Cpos), C.c_float(float(theta)), _p(ex), _p(R), _p(T))
    return R.reshape(3, 3), T


def project(verts, R, T, s):
    verts = _f(verts)
    out = np.zeros_like(verts)
    lib().orc_project(_p(verts), C.c_int(len(verts)), _p(_f(R)), _p(_f(T)), C.c_float(s), _p(out))
    return out


def rasterize(cfg, screen_verts, faces, crop=True):
    """Naive S x S rasterisation; returns window (crop=True) or full pix_to_face, bary."""
    sv, fc = _f(screen_verts), _i(faces)
    S = cfg.S
    p2f = np.zeros((S, S), np.int32)
    bary = np.zeros((S, S, 3), np.float32)
    zbuf = np.zeros((S, S), np.float32)
    lib().orc_rasterize_naive(C.byref(cfg), _p(sv), _p(fc), C.c_int(len(fc)), _p(p2f), _p(bary), _p(zbuf))
    if not crop:
        return p2f, bary, zbuf
    sl = (slice(cfg.r0, cfg.r0 + cfg.h), slice(cfg.c0, cfg.c0 + cfg.w))
    return (np.ascontiguousarray(p2f[sl]), np.ascontiguousarray(bary[sl]),
            np.ascontiguousarray(zbuf[sl]))


def interpolate(cfg, p2f, bary, faces, bank):
    out = np.zeros((cfg.h, cfg.w, cfg.D), np.float32)
    lib().orc_interpolate(C.byref(cfg), _p(_i(p2f)), _p(_f(bary)), _p(_i(faces)), _p(_f(bank)), _p(out))
    return out


def clutter_score(cfg, feat, cbank):
    cbank = _f(cbank).reshape(-1, cfg.D)
    out = np.zeros((cfg.h, cfg.w), np.float32)
    lib().orc_clutter_score(C.byref(cfg), _p(_f(feat)), _p(cbank), C.c_int(len(cbank)), _p(out))
    return out


def evaluate(cfg, verts, faces, bank, feat, clu, Cpos, theta, extra=None, want_maps=False):
    verts, faces, bank, feat = _f(verts), _i(faces), _f(bank), _f(feat)
    clu = _f(clu) if clu is not None else None
    Cpos = _f(Cpos).reshape(3)
    ex = _f(extra) if extra is not None else None
    out = EvalOut()
    n = cfg.h * cfg.w
    p2f = np.zeros(n, np.int32) if want_maps else None
    bary = np.zeros(3 * n, np.float32) if want_maps else None
    obj = np.zeros(n, np.float32) if want_maps else None
    lib().orc_eval(C.byref(cfg), _p(verts), C.c_int(len(verts)), _p(faces), C.c_int(len(faces)),
                   _p(bank), _p(feat), _p(clu), _p(Cpos), C.c_float(float(theta)), _p(ex),
                   C.byref(out), _p(p2f), _p(bary), _p(obj))
    res = dict(loss=out.loss, gC=np.array(out.gC[:], np.float32), gtheta=out.gtheta,
               gextra=np.array(out.gextra[:], np.float32), ncov=out.ncov)
    if want_maps:
        res.update(pix_to_face=p2f.reshape(cfg.h, cfg.w), bary=bary.reshape(cfg.h, cfg.w, 3),
                   obj=obj.reshape(cfg.h, cfg.w))
    return res


def solve(cfg, verts, faces, bank, feat, clu, C0, theta0, adam_tab, b1=0.4, b2=0.6, eps=1e-8,
          want_trace=False):
    verts, faces, bank, feat = _f(verts), _i(faces), _f(bank), _f(feat)
    clu = _f(clu) if clu is not None else None
    adam_tab = _f(adam_tab)
    steps = len(adam_tab)
    outC, oth, ol = np.zeros(3, np.float32), C.c_float(), C.c_float()
    trace = np.zeros((steps, 9), np.float32) if want_trace else None
    lib().orc_solve(C.byref(cfg), _p(verts), C.c_int(len(verts)), _p(faces), C.c_int(len(faces)),
                    _p(bank), _p(feat), _p(clu), _p(_f(C0).reshape(3)), C.c_float(float(theta0)),
                    C.c_int(steps), C.c_float(b1), C.c_float(b2), C.c_float(eps), _p(adam_tab),
                    _p(outC), C.byref(oth), C.byref(ol), _p(trace))
    return outC, oth.value, ol.value, trace


def grid_search(cfg, verts, faces, bank, feat, clu, gridC, grid_theta):
    verts, faces, bank, feat = _f(verts), _i(faces), _f(bank), _f(feat)
    clu = _f(clu) if clu is not None else None
    gridC, grid_theta = _f(gridC), _f(grid_theta)
    losses = np.zeros(len(grid_theta), np.float32)
    lib().orc_grid_search.restype = C.c_int
    best = lib().orc_grid_search(C.byref(cfg), _p(verts), C.c_int(len(verts)), _p(faces),
                                 C.c_int(len(faces)), _p(bank), _p(feat), _p(clu), _p(gridC),
                                 _p(grid_theta), C.c_int(len(grid_theta)), _p(losses))
    return int(best), losses


def solve6(cfg, verts, faces, bank, feat, clu, C0, theta0, adam_tab, trans0=None, b1=0.4, b2=0.6, eps=1e-8,
           want_trace=False):
    """6-DoF solve (PredUnlabeledDataset.py:318-339): returns p6 = (C, theta, t0, t1), loss, trace [steps,13]."""
    verts, faces, bank, feat = _f(verts), _i(faces), _f(bank), _f(feat)
    clu = _f(clu) if clu is not None else None
    adam_tab = _f(adam_tab)
    steps = len(adam_tab)
    out_p, ol = np.zeros(6, np.float32), C.c_float()
    trace = np.zeros((steps, 13), np.float32) if want_trace else None
    t0 = _f(trans0).reshape(2) if trans0 is not None else None
    lib().orc_solve6(C.byref(cfg), _p(verts), C.c_int(len(verts)), _p(faces), C.c_int(len(faces)),
                     _p(bank), _p(feat), _p(clu), _p(_f(C0).reshape(3)), C.c_float(float(theta0)), _p(t0),
                     C.c_int(steps), C.c_float(b1), C.c_float(b2), C.c_float(eps), _p(adam_tab),
                     _p(out_p), C.byref(ol), _p(trace))
    return out_p, ol.value, trace


def solve_batch(cfg, verts, faces, bank, feats, cvec, gridC, grid_theta, adam_tab, b1=0.4, b2=0.6,
                eps=1e-8, nthreads=0):
    """clutter score + grid init + full solve for N images (OpenMP over images)."""
    verts, faces, bank, feats = _f(verts), _i(faces), _f(bank), _f(feats)
    N = feats.shape[0]
    adam_tab = _f(adam_tab)
    outC = np.zeros((N, 3), np.float32)
    oth, ol = np.zeros(N, np.float32), np.zeros(N, np.float32)
    oi = np.zeros(N, np.int32)
    gridC, grid_theta = _f(gridC), _f(grid_theta)
    lib().orc_solve_batch(C.byref(cfg), _p(verts), C.c_int(len(verts)), _p(faces), C.c_int(len(faces)),
                          _p(bank), _p(feats), _p(_f(cvec).reshape(-1)), C.c_int(N), _p(gridC),
                          _p(grid_theta), C.c_int(len(grid_theta)), C.c_int(len(adam_tab)),
                          C.c_float(b1), C.c_float(b2), C.c_float(eps), _p(adam_tab), _p(outC),
                          _p(oth), _p(ol), _p(oi), C.c_int(nthreads))
    return outC, oth, ol, oi


def adam_step(p, m, v, g, b1, b2, step_size, bc2_sqrt, eps=1e-8):
    p, m, v, g = _f(p).copy(), _f(m).copy(), _f(v).copy(), _f(g)
    lib().orc_adam_step(_p(p), _p(m), _p(v), _p(g), C.c_int(p.size), C.c_float(b1), C.c_float(b2),
                        C.c_float(step_size), C.c_float(bc2_sqrt), C.c_float(eps))
    return p, m, v
