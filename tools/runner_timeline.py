#!/usr/bin/env python3
"""Timeline of the file-driven runner (nemo_amd.solve_all.solve_shard) on the bench's synthetic one-category record:
per batch, when its maps were on the device, when its launches started and when they ended (device clock), and the
host time at which each was queued.  usage: tools/runner_timeline.py [--images 1024] [--batch 256] [--streams 2]"""
import argparse
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nemo_amd import mesh_utils as mu, records, solve_all, synthetic as syn  # noqa: E402
from nemo_amd.workloads import twelve_class_workload  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--images', type=int, default=1024)
ap.add_argument('--batch', type=int, default=256)
ap.add_argument('--streams', type=int, default=2)
ap.add_argument('--read-threads', type=int, default=0)
ap.add_argument('--chunk', type=int, default=64)
ap.add_argument('--mmap', action='store_true')
a = ap.parse_args()
_orig_upload = records.FeatureRecord.upload_pixel_major


def _upload(self, nm, dv, chunk=64, workers=0, stats=None):
    self.use_mmap = a.mmap
    return _orig_upload(self, nm, dv, chunk=a.chunk, workers=a.read_threads, stats=stats)


records.FeatureRecord.upload_pixel_major = _upload
dev = torch.device('cuda:0')
wl = twelve_class_workload(a.images, seed=77, device=dev, categories=['car'])
eng = wl.solver.engine
host = wl.feats.permute(0, 3, 1, 2).contiguous().cpu().numpy()
d = tempfile.mkdtemp(prefix='nemo_tl_', dir='/dev/shm')
try:
    names = ['img_%05d' % i for i in range(a.images)]
    v, f, bank = (t.cpu().numpy() for t in eng._keep[0])
    os.makedirs(os.path.join(d, 'cad'))
    mu.save_off(os.path.join(d, 'cad', '01.off'), np.stack([v[:, 0], -v[:, 2], v[:, 1]], 1), f)
    rec_path = os.path.join(d, 'car.npz')
    records.write_feature_record(rec_path, {'mesh01': bank}, {'mesh01': syn.unit_rows(np.random.default_rng(5), 64, eng.D)},
                                 {'mesh01': names}, {n: host[i] for i, n in enumerate(names)})
    h, w = host.shape[2:]
    del host, wl
    torch.cuda.empty_cache()
    args = solve_all.build_parser().parse_args(['--type_', 'car', '--record_file', rec_path, '--mesh_path', os.path.join(d, 'cad'),
                                                '--anno_path', '', '--batch', str(a.batch), '--streams', str(a.streams),
                                                '--image_size', '%d,%d' % (8 * h, 8 * w)])
    print(f'read threads {a.read_threads or 16}, chunk {a.chunk}, mmap {a.mmap}')
    for rep in range(3):
        st = {}
        with records.FeatureRecord(rec_path) as rec:
            torch.cuda.synchronize(); t0 = time.perf_counter()
            solve_all.solve_shard(args, rec, ['mesh01'], names, [0] * len(names), 'cuda:0', stats=st)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f'run {rep}: {dt*1e3:.1f} ms = {a.images/dt:.0f} solves/s (batch {a.batch}, {a.streams} streams); read {1e3*st["decode_s"]:.0f} ms, '
              f'stage wait {1e3*st["stage_wait_s"]:.0f} ms', flush=True)
        if rep == 2 and os.environ.get('TIMELINE', '1') != '0':
            for row in st['timeline']:
                if len(row) > 2:
                    print('  %-32s queued at %6.1f ms | maps on device %6.1f, start %6.1f, end %6.1f ms' % (row[0], row[1], row[2], row[3], row[4]))
                else:
                    print('  %-32s at %6.1f ms' % row)
finally:
    shutil.rmtree(d, ignore_errors=True)
