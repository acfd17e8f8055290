"""On-disk formats either side of the solver.

* feature record (written by code/ExtractFeatures.py:137-139,166-168,223, read by
  code/MeshPoseSolveAll.py:173,199-218): one .npz per category with
  ``memory_<subtype>`` [n_vertices, D], ``clutter_<subtype>`` [n_noise, D],
  ``names_<subtype>`` (image names) and one ``<image name>`` -> [D, h, w] f32 map per image;
* annotation (one .npz per image, code/MeshPoseSolveAll.py:256-258,270): theta, elevation,
  azimuth, distance, cad_index;
* per-category error record (code/MeshPoseSolveAll.py:273-274) and the accuracy table of
  code/CalAccuracy.py (pi/6, pi/18 accuracy in per cent, median error in degrees, image-weighted mean).

Reading never unpickles: records are opened with ``allow_pickle=False``; a member that needs
pickling (the reference's ragged ``subtype_error`` object array) is refused with a clear error and
is not needed by any consumer here (CalAccuracy reads ``total_error`` only).
"""
from __future__ import annotations

import os
import struct
import time
import zipfile
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch

from .camera_params import cal_err, cal_rotation_matrix

CATEGORIES = ['aeroplane', 'bicycle', 'boat', 'bottle', 'bus', 'car', 'chair', 'diningtable', 'motorbike', 'sofa',
              'train', 'tvmonitor']
SHORT_NAMES = ['plane', 'bike', 'boat', 'bottle', 'bus', 'car', 'chair', 'table', 'mbike', 'sofa', 'train', 'tv']


# ------------------------------------------------------------------ feature record
class FeatureRecord:
    """Lazy view of a feature .npz (members are read on access, one at a time)."""

    def __init__(self, path: str):
        self.path = path
        self._z = np.load(path, allow_pickle=False)
        keys = list(self._z.files)
        self.subtypes = sorted(k[len('memory_'):] for k in keys if k.startswith('memory_'))
        meta = {p + s for s in self.subtypes for p in ('memory_', 'clutter_', 'names_')}
        self.image_keys = [k for k in keys if k not in meta]

    def memory(self, subtype: str) -> np.ndarray:
        return np.ascontiguousarray(self._z['memory_%s' % subtype], dtype=np.float32)

    def clutter(self, subtype: str) -> np.ndarray:
        return np.ascontiguousarray(self._z['clutter_%s' % subtype], dtype=np.float32)

    def names(self, subtype: str) -> List[str]:
        return [str(n) for n in np.asarray(self._z['names_%s' % subtype]).ravel().tolist()]

    def feature(self, name: str) -> np.ndarray:
        """[D, h, w] f32, the layout ExtractFeatures.py:166-168 stores."""
        return np.asarray(self._z[name], dtype=np.float32)

    # ---- direct reads: np.savez stores members uncompressed, so a map's bytes sit in the file as they are
    def _layout(self, name: str):
        """(byte offset of the array data in the .npz, shape) of a STORED little-endian f32 C-order member, else None
        (compressed record, other dtype: the caller falls back to the zipfile path)."""
        if not hasattr(self, '_zinfo'):
            with zipfile.ZipFile(self.path) as zf:
                self._zinfo = {i.filename: i for i in zf.infolist()}
            self._layouts = {}
        if name in self._layouts:
            return self._layouts[name]
        info = self._zinfo.get(name + '.npy')
        lay = None
        if info is not None and info.compress_type == zipfile.ZIP_STORED:
            with open(self.path, 'rb') as fh:
                fh.seek(info.header_offset)
                local = fh.read(30)
                n_name, n_extra = struct.unpack('<HH', local[26:30])
                start = info.header_offset + 30 + n_name + n_extra
                fh.seek(start)
                version = np.lib.format.read_magic(fh)
                shape, fortran, dtype = (np.lib.format.read_array_header_1_0(fh) if version == (1, 0)
                                         else np.lib.format.read_array_header_2_0(fh))
                if dtype == np.dtype('<f4') and not fortran:
                    lay = (fh.tell(), tuple(shape))
        self._layouts[name] = lay
        return lay

    def read_feature_into(self, name: str, dst: np.ndarray, fd: Optional[int] = None) -> None:
        """Fill `dst` ([D, h, w] f32, C-contiguous; e.g. a view of pinned memory) with the map `name`: a straight
        pread of the member's bytes when the record is uncompressed (no intermediate copy, no CRC pass), else
        through the zip reader."""
        lay = self._layout(name)
        if lay is None:
            f = self.feature(name)
            if f.shape != dst.shape:
                raise ValueError('feature map %r has shape %s, expected %s' % (name, f.shape, dst.shape))
            dst[...] = f
            return
        off, shape = lay
        if shape != tuple(dst.shape):
            raise ValueError('feature map %r has shape %s, expected %s' % (name, shape, tuple(dst.shape)))
        if getattr(self, 'use_mmap', False):  # copy out of a read-only mapping of the record instead of pread
            if not hasattr(self, '_mm'):
                import mmap
                self._mm_fh = open(self.path, 'rb')
                self._mm = np.frombuffer(mmap.mmap(self._mm_fh.fileno(), 0, access=mmap.ACCESS_READ), dtype=np.uint8)
            n = dst.size * 4
            np.copyto(dst.reshape(-1).view(np.uint8), self._mm[off:off + n])
            return
        own = fd is None
        if own:
            fd = os.open(self.path, os.O_RDONLY)
        try:
            view = memoryview(dst).cast('B')
            done = 0
            while done < len(view):
                n = os.preadv(fd, [view[done:]], off + done)
                if n <= 0:
                    raise IOError('short read of %r in %s' % (name, self.path))
                done += n
        finally:
            if own:
                os.close(fd)

    def feature_shape(self, name: str):
        lay = self._layout(name)
        return lay[1] if lay is not None else tuple(self.feature(name).shape)

    def close(self):
        self._z.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def upload_pixel_major(self, names: Sequence[str], device, chunk: int = 64, workers: int = 0,
                           stats: Optional[dict] = None) -> torch.Tensor:
        """Features of `names` as ONE [N, h, w, D] f32 device tensor (the kernels' layout).  Double-buffered: while
        chunk k goes host -> device (pinned memory, asynchronous copy on the CURRENT stream), `workers` host threads
        read chunk k + 1 straight from the file into the other pinned buffer.  On a GPU the chunks land in a [N, D, h, w]
        device buffer and ONE transpose to pixel-major follows the last copy: nothing but DMA copies is queued while
        the maps come in, because a kernel queued behind a running solve (whose workgroups fill every CU's register
        file) would not start before that solve ends and would hold its pinned buffer until then -- pinned copies are
        not slowed by a running solve (57 GB/s either way), whereas copies from pageable memory, a mapping of the
        file included, drop from 41-48 to 7.6 GB/s under one (tools/from_files_breakdown.py).  The host never holds
        a transposed copy.  stats (optional dict) receives read / wait times and the byte count."""
        if workers <= 0:
            workers = min(16, os.cpu_count() or 1)  # reader threads (a GPU box gives a job 16 CPUs)
        D, h, w = self.feature_shape(names[0])
        dev = torch.device(device)
        pin = dev.type == 'cuda'
        # GPU: file order [N, D, h, w] first, transposed once at the end; CPU: transposed chunk by chunk
        out = torch.empty((len(names), D, h, w) if pin else (len(names), h, w, D), dtype=torch.float32, device=dev)
        n_stage = min(chunk, len(names))
        stages = [torch.empty((n_stage, D, h, w), dtype=torch.float32, pin_memory=pin) for _ in range(2 if len(names) > chunk else 1)]
        views = [st.numpy() for st in stages]
        free_ev = [None] * len(stages)  # the copy out of a stage buffer must finish before it is refilled
        fds = [os.open(self.path, os.O_RDONLY) for _ in range(max(1, workers))]
        t_decode = t_wait = 0.0

        def fill(k, part):
            def one(j):
                self.read_feature_into(part[j], views[k][j], fds[j % len(fds)])
            if workers > 1 and len(part) > 1:
                list(pool.map(one, range(len(part))))
            else:
                for j in range(len(part)):
                    one(j)

        pool = ThreadPoolExecutor(max_workers=max(1, workers))
        try:
            for ci, s in enumerate(range(0, len(names), chunk)):
                k = ci % len(stages)
                part = names[s:s + chunk]
                if free_ev[k] is not None:
                    t0 = time.perf_counter()
                    free_ev[k].synchronize()
                    t_wait += time.perf_counter() - t0
                t0 = time.perf_counter()
                fill(k, part)
                t_decode += time.perf_counter() - t0
                if pin:
                    out[s:s + len(part)].copy_(stages[k][:len(part)], non_blocking=True)
                    free_ev[k] = torch.cuda.Event()
                    free_ev[k].record(torch.cuda.current_stream(dev))
                else:
                    out[s:s + len(part)] = stages[k][:len(part)].permute(0, 2, 3, 1)
        finally:
            pool.shutdown()
            for fd in fds:
                os.close(fd)
        if stats is not None:
            stats.update(decode_s=stats.get('decode_s', 0.0) + t_decode, stage_wait_s=stats.get('stage_wait_s', 0.0) + t_wait,
                         bytes=stats.get('bytes', 0) + len(names) * D * h * w * 4)
        return out.permute(0, 2, 3, 1).contiguous() if pin else out


def write_feature_record(path: str, memory: Dict[str, np.ndarray], clutter: Dict[str, np.ndarray],
                         names: Dict[str, Sequence[str]], features: Dict[str, np.ndarray]) -> None:
    """Writes the ExtractFeatures.py format (:137-139,166-168,223); `features[name]` is [D, h, w]."""
    get = {}
    for st in memory:
        get['memory_%s' % st] = np.asarray(memory[st], dtype=np.float32)
        get['clutter_%s' % st] = np.asarray(clutter[st], dtype=np.float32)
        get['names_%s' % st] = np.asarray(list(names[st]), dtype=str)
    for n, f in features.items():
        get[n] = np.asarray(f, dtype=np.float32)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    np.savez(path, **get)


# ------------------------------------------------------------------ annotations
@dataclass
class Annotation:
    theta: float
    elevation: float
    azimuth: float
    distance: float
    cad_index: int


def load_annotation(path: str) -> Annotation:
    """theta / elevation / azimuth / distance / cad_index of one image (code/MeshPoseSolveAll.py:256-258,270).
    Opened without unpickling; PASCAL3D+ annotation files made by dataset/CreatePascal3DNeMo.py also carry
    object-array members, which are simply not touched."""
    with np.load(path, allow_pickle=False) as z:
        g = lambda k: float(np.asarray(z[k]).reshape(-1)[0])  # noqa: E731
        return Annotation(g('theta'), g('elevation'), g('azimuth'), g('distance'), int(np.asarray(z['cad_index']).reshape(-1)[0]))


def write_annotation(path: str, a: Annotation) -> None:
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    np.savez(path, theta=a.theta, elevation=a.elevation, azimuth=a.azimuth, distance=a.distance, cad_index=a.cad_index)


# ------------------------------------------------------------------ errors and accuracy
def rotation_error(anno: Annotation, theta, elevation, azimuth, distance) -> float:
    """code/MeshPoseSolveAll.py:259-266: geodesic distance of the two rotations, pi/2 if either is not finite."""
    am = cal_rotation_matrix(anno.theta, anno.elevation, anno.azimuth, anno.distance)
    pm = cal_rotation_matrix(float(theta), float(elevation), float(azimuth), float(distance))
    if not (np.all(np.isfinite(am)) and np.all(np.isfinite(pm))):
        return float(np.pi / 2)
    return float(np.real(cal_err(am, pm)))


def save_category_errors(path: str, total_error: Sequence[float], cad_index: Optional[Sequence[int]] = None) -> None:
    """The per-category record CalAccuracy.py reads (`total_error`, code/MeshPoseSolveAll.py:273-274).  The
    reference also stores the ragged per-subtype lists as a pickled object array; here the subtype of every
    error is stored as a plain int array instead (`cad_index`, 1-based), from which those lists follow."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    kw = dict(total_error=np.asarray(total_error, dtype=np.float64))
    if cad_index is not None:
        kw['cad_index'] = np.asarray(cad_index, dtype=np.int32)
    np.savez(path, **kw)


def load_category_errors(path: str) -> np.ndarray:
    with np.load(path, allow_pickle=False) as z:
        return np.asarray(z['total_error'], dtype=np.float64)


def accuracy_rows(errors_by_category: Dict[str, np.ndarray], categories: Iterable[str] = CATEGORIES) -> np.ndarray:
    """code/CalAccuracy.py:15-24: rows = pi/6 accuracy %, pi/18 accuracy %, median error in degrees, count;
    one column per category plus the image-weighted mean as the last column."""
    cats = list(categories)
    out = np.zeros((4, len(cats) + 1), dtype=np.float32)  # float32 like the reference's table (:14): the printed digits follow it
    for i, c in enumerate(cats):
        e = np.asarray(errors_by_category[c], dtype=np.float64)
        out[0, i] = float(np.mean(e < np.pi / 6)) * 100
        out[1, i] = float(np.mean(e < np.pi / 18)) * 100
        out[2, i] = float(180 / np.pi * np.median(e))
        out[3, i] = e.size
    n = out[3, :-1]
    for r in range(3):
        out[r, -1] = np.sum(n * out[r, :-1]) / np.sum(n)
    out[3, -1] = np.sum(n)
    return out


def format_accuracy_table(rows: np.ndarray, short_names: Sequence[str] = SHORT_NAMES) -> str:
    """The text code/CalAccuracy.py:26-44 prints."""
    lines = ['Metric | ' + ''.join(c + '\t' for c in short_names) + 'Mean']
    for label, r in (('Pi/6   | ', 0), ('Pi/18  | ', 1), ('MedErr | ', 2)):
        lines.append(label + ''.join('%.1f\t' % rows[r, i] for i in range(rows.shape[1] - 1)) + '%.1f' % rows[r, -1])
    return '\n'.join(lines)


def accuracy_table_from_dir(load_accuracy: str, data_pendix: str = '', categories: Iterable[str] = CATEGORIES):
    """code/CalAccuracy.py end to end: reads <dir>/<category><pendix>.npz, returns (rows, text)."""
    cats = list(categories)
    errs = {c: load_category_errors(os.path.join(load_accuracy, '%s%s.npz' % (c, data_pendix))) for c in cats}
    rows = accuracy_rows(errs, cats)
    names = [SHORT_NAMES[CATEGORIES.index(c)] if c in CATEGORIES else c for c in cats]
    return rows, format_accuracy_table(rows, names)
