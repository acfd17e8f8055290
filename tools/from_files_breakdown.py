#!/usr/bin/env python3
"""Where the file-driven runner's time goes: writes the bench's synthetic one-category record to /dev/shm and times
the phases of nemo_amd.solve_all.solve_shard (solver set-up, wait for each batch's upload, each batch's solve)."""
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
a = ap.parse_args()
dev = torch.device('cuda:0')
wl = twelve_class_workload(a.images, seed=77, device=dev, categories=['car'])
eng = wl.solver.engine
host = wl.feats.permute(0, 3, 1, 2).contiguous().cpu().numpy()
d = tempfile.mkdtemp(prefix='nemo_ff_', dir='/dev/shm')
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
                                                '--anno_path', '', '--batch', str(a.batch), '--image_size', '%d,%d' % (8 * h, 8 * w)])
    marks = []
    orig_build, orig_upload = solve_all.build_solver, records.FeatureRecord.upload_pixel_major

    def build(*x, **k):
        t0 = time.perf_counter(); r = orig_build(*x, **k); torch.cuda.synchronize(); marks.append(('build_solver', time.perf_counter() - t0)); return r

    def upload(self, nm, dv, **k):
        t0 = time.perf_counter(); r = orig_upload(self, nm, dv, **k); torch.cuda.current_stream(dv).synchronize()
        marks.append(('upload %d maps (helper thread)' % len(nm), time.perf_counter() - t0)); return r

    solve_all.build_solver, records.FeatureRecord.upload_pixel_major = build, upload
    for rep in range(2):
        marks.clear()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        solve_all.solve_category(args, 'cuda:0')
        dt = time.perf_counter() - t0
        print(f'run {rep}: {dt*1e3:.1f} ms = {a.images/dt:.0f} solves/s; ' + ', '.join(f'{n} {1e3*t:.1f} ms' for n, t in marks), flush=True)
    # the solve of one batch alone, for reference
    with records.FeatureRecord(rec_path) as rec:
        feats = rec.upload_pixel_major(names[:a.batch], dev)
        solver, clutter = orig_build(args, rec, ['mesh01'], dev)
        solver.engine.reserve(a.batch, int(solver.grid_theta.shape[1])); solver.prerender()
        mi = torch.zeros(a.batch, dtype=torch.int32, device=dev)
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            clu = solver.clutter_scores_by_group(feats, [0] * a.batch, clutter)
            solver.solve_all(feats, None, mi, clutter_scores=clu)
            torch.cuda.synchronize(); print(f'one batch of {a.batch}: clutter + grid + solve {1e3*(time.perf_counter()-t0):.1f} ms', flush=True)
        # copies while the GPU is busy with a long solve: does the running kernel slow them down?
        big = torch.cat([feats] * 4)
        mi4 = torch.zeros(4 * a.batch, dtype=torch.int32, device=dev)
        solver.engine.reserve(4 * a.batch, int(solver.grid_theta.shape[1]))
        clu4 = solver.clutter_scores_by_group(big, [0] * (4 * a.batch), clutter)
        offs = [rec._layout(n)[0] for n in names[:a.batch]]
        nbytes = feats[0].numel() * 4
        import mmap
        fh = open(rec_path, 'rb'); mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_COPY)
        src = torch.from_numpy(np.frombuffer(mm, dtype=np.uint8))
        stage = torch.empty((a.batch, nbytes), dtype=torch.uint8, device=dev)
        pin = torch.empty((a.batch, nbytes), dtype=torch.uint8).pin_memory()
        from concurrent.futures import ThreadPoolExecutor
        streams = [torch.cuda.Stream(dev) for _ in range(4)]

        def copies():
            def work(t):
                with torch.cuda.stream(streams[t]):
                    for j in range(t, a.batch, 4):
                        stage[j].copy_(src[offs[j]:offs[j] + nbytes], non_blocking=True)
                streams[t].synchronize()
            with ThreadPoolExecutor(4) as pool:
                list(pool.map(work, range(4)))

        def pinned():
            with torch.cuda.stream(streams[0]):
                stage.copy_(pin, non_blocking=True)
            streams[0].synchronize()

        for label, fn in (('mapped file -> device, 4 threads', copies), ('pinned -> device, one async copy', pinned)):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter(); fn(); idle = time.perf_counter() - t0
            solver.solve_all(big, None, mi4, clutter_scores=clu4)   # ~130 ms of GPU work, asynchronous
            t0 = time.perf_counter(); fn(); busy = time.perf_counter() - t0
            torch.cuda.synchronize()
            print(f'{label}: {a.batch * nbytes / idle / 1e9:.1f} GB/s on an idle GPU, {a.batch * nbytes / busy / 1e9:.1f} GB/s under a running solve', flush=True)
        # host side alone: the maps of one batch into pinned memory
        del rec._layouts, rec._zinfo
        t0 = time.perf_counter(); lays = [rec._layout(n) for n in names]; print(f'layouts of {len(names)} members: {1e3*(time.perf_counter()-t0):.1f} ms', flush=True)
        pin4 = pin.view(a.batch, -1)
        views = [memoryview(pin4[j].numpy()) for j in range(a.batch)]
        mapped = np.frombuffer(mm, dtype=np.uint8)
        for nthr in (4, 8, 16):
            fds = [os.open(rec_path, os.O_RDONLY) for _ in range(nthr)]

            def rd(j):
                os.preadv(fds[j % nthr], [views[j]], offs[j])

            def cp(j):
                np.copyto(pin4[j].numpy(), mapped[offs[j]:offs[j] + nbytes])

            for label, fn in (('pread', rd), ('copy from the mapping', cp)):
                with ThreadPoolExecutor(nthr) as pool:
                    list(pool.map(fn, range(a.batch)))
                    t0 = time.perf_counter(); list(pool.map(fn, range(a.batch))); dt = time.perf_counter() - t0
                print(f'{label}, {nthr} threads: {a.batch * nbytes / dt / 1e9:.1f} GB/s', flush=True)
            [os.close(fd) for fd in fds]
        st = {}
        t0 = time.perf_counter(); rec.upload_pixel_major(names[:a.batch], dev, stats=st); torch.cuda.synchronize()
        print(f'upload_pixel_major: {1e3*(time.perf_counter()-t0):.1f} ms, read {1e3*st["decode_s"]:.1f} ms, stage wait {1e3*st["stage_wait_s"]:.1f} ms', flush=True)
        st = {}
        t0 = time.perf_counter(); rec.upload_pixel_major(names[:a.batch], dev, stats=st); torch.cuda.synchronize()
        print(f'upload_pixel_major: {1e3*(time.perf_counter()-t0):.1f} ms, read {1e3*st["decode_s"]:.1f} ms, stage wait {1e3*st["stage_wait_s"]:.1f} ms', flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
