#!/usr/bin/env python3
"""Headline benchmark: pose solves/sec on the SingleCuboid 12-class workload
(BASELINE.json configs[1]: 1024 synthetic 640x800 images per GPU, 128-d random-init
vertex features, 144-pose grid init + 300 Adam steps per image).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over the batch: clutter score -> pose-grid
search -> fused 300-step solve -> decode (+ one RCCL all-gather of the poses when
N > 1).  Inputs are resident in HBM before the timed region.  For N > 1 there is
one rank per GPU: either the caller starts them (python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N ...) or, when WORLD_SIZE is not set, this
script starts that launcher itself as a child process -- before anything here
touches a GPU -- and relays rank 0's JSON line and the exit code.  Images are
sharded per rank (weak scaling), no collective on the data path.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_baseline(wl, n_sample):
    """Oracle (CPU port of the reference op sequence: naive all-faces-per-pixel raster over the
    full SxS square, gather-interpolate, loss, backward, Adam; 144-pose init + 300 steps) on a
    bounded sample of the same workload, one image per host thread."""
    from oracle import oracle as orc  # checker only; never on the product path
    from nemo_amd import synthetic as syn
    s = wl.solver
    eng = s.engine
    cfg = orc.make_cfg(eng.S, eng.h, eng.w, eng.D, eng.ndc_scale, perspective_correct=s.cfg.perspective_correct,
                       clutter=not s.cfg.turn_off_clutter, k_eps=s.cfg.k_eps)
    tab = syn.adam_table(s.cfg.total_epochs, s.cfg.lr, s.cfg.adam_beta_0, s.cfg.adam_beta_1,
                         s.cfg.lr_decay_every, s.cfg.lr_decay_gamma)
    feats = wl.feats[:n_sample].cpu().numpy()
    cvec = wl.clutter[0].cpu().numpy()
    mesh = wl.mesh_idx[:n_sample].cpu().numpy()
    meshes = [(v.cpu().numpy(), f.cpu().numpy(), b.cpu().numpy()) for v, f, b in eng._keep]
    gC, gT = s.grid_C.cpu().numpy(), s.grid_theta.cpu().numpy()
    orc.lib()
    cores = min(16, os.cpu_count() or 1, n_sample)
    out = [None] * n_sample

    def work(k):  # worker k takes images k, k + cores, ...
        for i in range(k, n_sample, cores):
            v, f, b = meshes[mesh[i]]
            out[i] = orc.solve_batch(cfg, v, f, b, feats[i:i + 1], cvec, gC[mesh[i]], gT[mesh[i]], tab,
                                     s.cfg.adam_beta_0, s.cfg.adam_beta_1, s.cfg.adam_eps, nthreads=1)

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(cores)]
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    return {'value': n_sample / dt, 'unit': 'solves/s', 'cores': cores, 'per_core': n_sample / dt / cores, 'kind': 'port',
            'sample': f'{n_sample} images of the same workload, i.e. the 12-class mix of BASELINE configs[1] (configs[0], the '
                      f"reference's own CPU-runnable case, is 16 car images: same per-image work, one class), 144-pose init + "
                      f'{s.cfg.total_epochs} Adam steps each, {cores} host threads (the oracle is single-threaded per image; '
                      f'ctypes releases the GIL), {dt:.1f} s wall; host has {os.cpu_count()} logical CPUs.  The port runs a '
                      f'hand-derived backward, which is faster than the torch-autograd op sequence of the reference'}


def cpu_baseline_autograd(n_proc, steps=8):
    """SURVEY 8(d)'s CPU baseline: the reference's own op sequence (look-at -> transform -> naive all-faces-per-pixel
    raster over the full S x S square -> gather-interpolate -> crop -> mul/sum -> loss -> torch.autograd backward ->
    torch.optim.Adam, code/MeshPoseSolveAll.py:235-250) on BASELINE configs[0]'s image (car, 640x800, V = 1088), one
    single-threaded worker PROCESS per host core, all at the same time (InferenceNeMo.sh:67-78 runs one process per
    category).  Each worker (oracle/ref_sequence.py, checker-side code) times `steps` Adam steps, 16 cached-grid
    evaluations and 4 pre-renders; a solve is EXTRAPOLATED from those as 300 steps + 144 grid evaluations + 144/16
    pre-renders.  PyTorch3D's native rasteriser is stood in for by the oracle's C restatement of it."""
    env = dict(os.environ, OMP_NUM_THREADS='1', MKL_NUM_THREADS='1', HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
    env['PYTHONPATH'] = ROOT + os.pathsep + env.get('PYTHONPATH', '')
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, '-m', 'oracle.ref_sequence', '--steps', str(steps), '--seed', str(100 + k)],
                              cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for k in range(n_proc)]
    outs = []
    for pr in procs:
        try:
            so, se = pr.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            pr.kill()
            so, se = pr.communicate()
        if pr.returncode == 0 and so.strip():
            outs.append(json.loads(so.strip().splitlines()[-1]))
    dt = time.perf_counter() - t0
    if not outs:
        return {'error': 'no worker finished', 'cores': n_proc}
    solve_s = float(np.mean([o['solve_s_extrapolated'] for o in outs]))
    return {'value': len(outs) / solve_s, 'unit': 'solves/s', 'cores': len(outs), 'per_core': 1.0 / solve_s, 'kind': 'port',
            'extrapolated': True, 'solve_s_per_image_per_core': solve_s,
            'ms_per_adam_step': 1e3 * float(np.mean([o['t_step_s'] for o in outs])),
            'ms_per_cached_grid_evaluation': 1e3 * float(np.mean([o['t_grid_eval_s'] for o in outs])),
            'ms_per_prerender': 1e3 * float(np.mean([o['t_prerender_s'] for o in outs])),
            'sample': f"BASELINE configs[0]'s image (car, 640x800, 100x100 raster, V = 1088, F = 1844, D = 128): {len(outs)} "
                      f'single-threaded worker processes at once (torch.set_num_threads(1)), each timing {steps} Adam steps of the '
                      f"reference's op sequence with torch.autograd + torch.optim.Adam, 16 cached-grid evaluations and 4 pre-renders; "
                      f'one solve = 300 steps + 144 evaluations + 144/16 pre-renders EXTRAPOLATED from those; {dt:.1f} s wall; '
                      f'host has {os.cpu_count()} logical CPUs.  The rasteriser is the C restatement of '
                      f"PyTorch3D's naive CPU loop (PyTorch3D itself cannot run here)"}


MFMA_FP32_PEAK_TFLOPS = 157.3  # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md)


def roofline_mfma(eng, feats, K=2560, reps=3):
    """The path's matrix-core contraction, timed on its own (never part of `value`): clutter_kernel = pixels x clutter
    bank, max over the bank (lib/MeshUtils.py:273-275), on the bench's resident maps against a bank of the reference's
    size, K = num_noise * max_group = 5 * 512 = 2560 (TrainNeMo.sh:24-25).  flops = 2 h w D K per image."""
    from nemo_amd import synthetic as syn
    bank = torch.as_tensor(syn.unit_rows(np.random.default_rng(7), K, eng.D)).to(eng.device).contiguous()
    eng.clutter_score(feats, bank)
    torch.cuda.synchronize(eng.device)
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.clutter_score(feats, bank)
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize(eng.device)
    ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    n = feats.shape[0]
    flops = 2.0 * eng.h * eng.w * eng.D * K * n
    tf = flops / (ms * 1e-3) / 1e12
    return {'bound': 'mfma', 'achieved': tf, 'peak': MFMA_FP32_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': tf / MFMA_FP32_PEAK_TFLOPS,
            'traffic': None, 'kernel': 'clutter_kernel', 'kernel_ms': ms, 'flops_per_launch': flops, 'dtype': 'f32',
            'config': {'images': int(n), 'feature_map': [eng.h, eng.w, eng.D], 'clutter_vectors': K},
            'note': 'separate from `value`: the bench workload scores clutter with the reference\'s one mean vector (K = 1, a '
                    'streaming kernel); this block runs the K-vector contraction of the same maps on the fp32 matrix cores'}


def from_files_leg(n_images, dev, batch):
    """The file-driven entry (python -m nemo_amd.solve_all = the reference's code/MeshPoseSolveAll.py:173,217-219:
    feature .npz + CAD .off in, poses out) on a synthetic one-category record of `n_images` car images, written to a
    memory-backed directory.  Reported separately from `value` (H2D of feature maps reported separately):
    host read rate of the maps into pinned memory, pinned host -> device rate, and end-to-end solves/s with the next
    batch read + uploaded under the current batch's solve."""
    import shutil
    import tempfile
    from nemo_amd import mesh_utils as mu, records, solve_all, synthetic as syn
    from nemo_amd.workloads import twelve_class_workload
    wl = twelve_class_workload(n_images, seed=77, device=dev, categories=['car'])
    eng = wl.solver.engine
    host = wl.feats.permute(0, 3, 1, 2).contiguous().cpu().numpy()      # [n, D, h, w]: the layout ExtractFeatures.py stores
    base = '/dev/shm' if os.path.isdir('/dev/shm') and shutil.disk_usage('/dev/shm').free > 3 * host.nbytes else None
    d = tempfile.mkdtemp(prefix='nemo_bench_', dir=base)
    try:
        names = ['img_%05d' % i for i in range(n_images)]
        v, f, bank = (t.cpu().numpy() for t in eng._keep[0])
        pascal = np.stack([v[:, 0], -v[:, 2], v[:, 1]], 1)              # undo pre_process_mesh_pascal
        os.makedirs(os.path.join(d, 'cad'))
        mu.save_off(os.path.join(d, 'cad', '01.off'), pascal, f)
        rec_path = os.path.join(d, 'car.npz')
        cbank = syn.unit_rows(np.random.default_rng(5), 64, eng.D)
        t0 = time.perf_counter()
        records.write_feature_record(rec_path, {'mesh01': bank}, {'mesh01': cbank}, {'mesh01': names},
                                     {n: host[i] for i, n in enumerate(names)})
        t_write = time.perf_counter() - t0
        nbytes = host.nbytes
        del host, wl
        torch.cuda.empty_cache()
        # (a) host side alone: the maps from the record into pinned memory, up to 16 reader threads
        with records.FeatureRecord(rec_path) as rec:
            D, h, w = rec.feature_shape(names[0])
            stage = torch.empty((min(256, n_images), D, h, w), dtype=torch.float32, pin_memory=True)
            view = stage.numpy()
            from concurrent.futures import ThreadPoolExecutor
            nthr = min(16, os.cpu_count() or 1)
            fds = [os.open(rec_path, os.O_RDONLY) for _ in range(nthr)]
            t0 = time.perf_counter()
            with ThreadPoolExecutor(nthr) as pool:
                list(pool.map(lambda j: rec.read_feature_into(names[j], view[j], fds[j % nthr]), range(stage.shape[0])))
            t_read = time.perf_counter() - t0
            [os.close(fd) for fd in fds]
        read_gbs = stage.numel() * 4 / t_read / 1e9
        # (b) pinned host -> device alone
        dst = torch.empty_like(stage, device=dev)
        dst.copy_(stage, non_blocking=True)
        torch.cuda.synchronize(dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            dst.copy_(stage, non_blocking=True)
        b.record()
        torch.cuda.synchronize(dev)
        h2d_gbs = 3 * stage.numel() * 4 / (a.elapsed_time(b) * 1e-3) / 1e9
        del dst, stage
        # (b2) what the runner does per batch, on an idle GPU: file -> pinned -> device, double-buffered, + one transpose
        with records.FeatureRecord(rec_path) as rec:
            nb = min(batch, n_images)
            for _ in range(2):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                up = rec.upload_pixel_major(names[:nb], dev)
                torch.cuda.synchronize(dev)
                t_up = time.perf_counter() - t0
            file_gbs = up.numel() * 4 / t_up / 1e9
            del up
        # (c) end to end through the runner (second run: libraries and the page cache are warm, like a second category)
        args = solve_all.build_parser().parse_args(['--type_', 'car', '--record_file', rec_path, '--mesh_path',
                                                    os.path.join(d, 'cad'), '--anno_path', '', '--batch', str(batch),
                                                    '--image_size', '%d,%d' % (8 * h, 8 * w)])
        dt = None
        for _ in range(2):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            _, _, recs = solve_all.solve_category(args, str(dev))
            dt = time.perf_counter() - t0
        ok = float(np.isfinite(recs[:, :5]).all(axis=1).mean())
        return {'images': n_images, 'record_bytes': nbytes, 'solves_per_s': n_images / dt, 'seconds': dt, 'batch': batch,
                'host_read_gbs': read_gbs, 'h2d_gbs': h2d_gbs, 'file_to_device_gbs': file_gbs, 'finite_fraction': ok, 'record_write_s': t_write,
                'note': 'python -m nemo_amd.solve_all path on one synthetic category (car): uncompressed feature .npz on '
                        + ('/dev/shm' if base else 'the temp directory') + ', maps read with pread into pinned memory by up to 16 threads '
                        '(host_read_gbs: first, cold pass), copied to the device by DMA on a side stream (h2d_gbs) and transposed to '
                        'pixel-major once per batch (file_to_device_gbs: the three together, warm), on a helper thread while the '
                        'previous batch is solved; '
                        'includes mesh registration and the grid pre-render of the category; never part of `value`'}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def launch_ranks(n_gpus, argv):
    """`bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py ...`
    as a CHILD (the one-command multi-GPU entry, as the reference's InferenceNeMo.sh:42-50,67-81 is one command).
    Called before this process has made any GPU call; nothing is exec'ed."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n_gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.call(cmd, env=env)


PMC_FILE = 'profiles/r03_pmc_traffic.json'            # streaming form of the solve
PMC_TABLES_FILE = 'profiles/r03_pmc_solve_tables.json'  # whole tables: bank_table_kernel + solve_kernel<TAB>
PMC_LAZY_FILE = 'profiles/r03_pmc_solve_lazy.json'      # tables on demand: solve_kernel<TAB, LAZY>
TABLE_BYTES_PER_COVERED_PIXEL_STEP = 16  # tables form: three 4-byte dot products + the 4-byte clutter score


def _pmc(fn, *keys):
    """A figure from a committed PMC record (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate runs, FETCH_SIZE
    doubled as the gfx950 note in MI355X_MICROARCH.md prescribes).  bench.py cannot collect counters itself; None if
    the file or the key is missing."""
    try:
        with open(os.path.join(ROOT, fn)) as f:
            v = json.load(f)
        for k in keys:
            v = v[k]
        return float(v)
    except (OSError, KeyError, ValueError, TypeError):
        return None


def pmc_traffic(cov_steps):
    """HBM bytes per launch of the streaming solve kernel, SCALED from the committed PMC passes: bytes per covered
    pixel-step of that pass x this launch's covered pixel-steps."""
    per = _pmc(PMC_FILE, 'solve_kernel', 'hbm_bytes_per_covered_pixel_step')
    return None if per is None else per * float(cov_steps)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--images', type=int, default=1024, help='images per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--from-files-images', type=int, default=1024, help='images in the file-driven leg (0 = skip it)')
    ap.add_argument('--from-files-batch', type=int, default=256)
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='process-group backend for N > 1: nccl = RCCL over xGMI (the product path); gloo = rehearsal of the '
                         'multi-rank code on a box with fewer GPUs than ranks (records gathered through host memory)')
    ap.add_argument('--share-gpu', action='store_true', help='rehearsal only: every rank uses GPU 0 (needs --backend gloo)')
    ap.add_argument('--cpu-sample', type=int, default=0, help='images in the CPU baseline sample (0 = 3 per host thread, at most 16 threads: about 13 s)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        print(f'bench.py: --gpus {args.gpus} but the launcher started {world} ranks', file=sys.stderr)
        sys.exit(2)
    # The process-per-core CPU baseline runs FIRST, before this process has touched the GPU: its workers are CPU-only
    # children, and a GPU box bounds the number of processes that may hold the device at once.
    cpu_autograd = None
    if world == 1 and not args.no_cpu_baseline and torch.cuda.device_count() > 0:
        cpu_autograd = cpu_baseline_autograd(min(16, os.cpu_count() or 1))
    if not torch.cuda.is_available():
        print(f'bench.py (rank {rank}): no GPU visible; the pose-solve path has no CPU fallback', file=sys.stderr)
        sys.exit(3)
    if args.share_gpu:
        if world > 1 and args.backend != 'gloo':
            print('bench.py: --share-gpu needs --backend gloo (RCCL refuses two ranks on one GPU)', file=sys.stderr)
            sys.exit(2)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group('gloo')
    from nemo_amd import distributed as nd

    from nemo_amd.workloads import twelve_class_workload
    wl = twelve_class_workload(args.images, seed=1000 + rank, device=dev)
    solver, eng = wl.solver, wl.solver.engine
    eng.reserve(args.images, int(solver.grid_theta.shape[1]))  # set-up: the timed calls allocate nothing in the library

    def step():
        res = solver.solve_all(wl.feats, wl.clutter, wl.mesh_idx)
        nd.gather_records(res.packed(), world * args.images)  # the path's one collective: [B, 8] records over RCCL
        return res

    for _ in range(args.warmup):
        step()
    nd.fence(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    nd.fence(dev)
    dt = nd.max_over_ranks(time.perf_counter() - t0, dev)

    # ---- the solve timed alone, on the launch stream.  Three forms (include/nemo_hip.h "TABLES / STREAMING"):
    #  2 tables on demand (default): ONE kernel, solve_kernel<TAB, LAZY> -- its waves compute the (32 vertices x 32
    #    pixels) blocks of feature-row x bank-row dot products their pixels need, on the matrix cores, the first time
    #    they need them, and every step reads three floats per covered pixel;
    #  1 whole tables (NEMO_SOLVE_LAZY=0): bank_table_kernel computes every block of every job, then solve_kernel<TAB>;
    #  0 streaming (no table workspace): solve_kernel contracts the feature rows itself in every step.
    # The library records HIP events around its kernels (nemo_solve_last_timing); torch events bracket the whole call.
    clu = eng.clutter_score(wl.feats, wl.clutter)
    C0, th0, _, _ = eng.grid_search(wl.feats, clu, solver.grid_C, solver.grid_theta, mesh_idx=wl.mesh_idx,
                                    grid_idx=wl.mesh_idx)
    torch.cuda.synchronize(dev)
    eng.counter_reset()
    evs, tabs = [], []
    reps = max(1, args.steps)
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.solve(wl.feats, clu, C0, th0, mesh_idx=wl.mesh_idx)
        b.record()
        evs.append((a, b))
        tabs.append(eng.solve_last_timing())   # (table_ms, loop_ms) or None; synchronises
    torch.cuda.synchronize(dev)
    solve_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    cov_steps = eng.counter_read() / reps  # covered pixel-steps of one solve launch
    form, blocks = eng.solve_table_stats()
    blocks /= reps                         # table blocks computed on demand per launch (form 2)
    stream_bytes = cov_steps * eng.bytes_per_covered_pixel()   # n_cov (4 D + 4) per job-step
    working_set = float(min(args.images, 256)) * stream_bytes / max(1, args.images * solver.cfg.total_epochs)
    extra = {}
    hw = eng.h * eng.w
    verts = np.asarray([int(v.shape[0]) for v, _, _ in eng._keep], dtype=np.float64)
    whole_flops = float((2.0 * verts[wl.mesh_idx.cpu().numpy()] * hw * eng.D).sum())   # 2 V hw D per job
    if form in (1, 2) and all(t is not None for t in tabs):
        table_ms = float(np.mean([t[0] for t in tabs])) if form == 1 else 0.0
        loop_ms = float(np.mean([t[1] for t in tabs]))
        loop_bytes = cov_steps * TABLE_BYTES_PER_COVERED_PIXEL_STEP
        # a block on demand: 32 bank rows and 32 feature rows read, 32 x 32 dot products written
        block_bytes = blocks * (64.0 * eng.D * 4 + 32 * 32 * 4)
        block_flops = blocks * 2.0 * 32 * 32 * eng.D
        pmc_file = PMC_LAZY_FILE if form == 2 else PMC_TABLES_FILE
        per_loop = _pmc(pmc_file, 'solve_kernel', 'hbm_bytes_per_covered_pixel_step')
        alg = loop_bytes + (block_bytes if form == 2 else 0.0)
        roof = {'bound': 'hbm', 'achieved': alg / (loop_ms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': alg / (loop_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                'traffic': None if per_loop is None else per_loop * cov_steps,
                'kernel': 'solve_kernel<TAB, LAZY>' if form == 2 else 'solve_kernel<TAB>', 'kernel_ms': loop_ms,
                'algorithmic_bytes_per_launch': alg,
                'bytes_per_covered_pixel_step': TABLE_BYTES_PER_COVERED_PIXEL_STEP,
                'covered_pixel_steps_per_launch': cov_steps,
                'traffic_source': f'scaled from the committed PMC pass {pmc_file} (not read during this run)',
                'note': 'the dominant kernel by time.  It is NOT bound by HBM (nor by the matrix cores): its 300 steps project, '
                        'rasterise, compact and score a job per step on the vector ALUs and the LDS (SQ counters: '
                        'profiles/r03_sq_solve_lazy.txt); per covered pixel and step it reads 16 bytes where the streaming '
                        'form read 516'
                        + ('; algorithmic bytes = those 16 x covered pixel-steps + 36 KB per table block computed on demand '
                           '(32 bank rows + 32 feature rows read, 4 KB written)' if form == 2 else '')
                        + '.  `traffic` above `algorithmic bytes`: a 4-byte table read fetches a cache line'}
        if form == 2:
            extra['roofline_table'] = {
                'bound': 'mfma', 'achieved': None, 'peak': MFMA_FP32_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': None,
                'kernel': 'solve_kernel<TAB, LAZY> (its matrix-core part; not timed apart from the kernel)',
                'blocks_per_launch': blocks, 'blocks_per_job': blocks / max(1, args.images),
                'flops_per_launch': block_flops, 'whole_table_flops_per_launch': whole_flops,
                'fraction_of_whole_table': block_flops / whole_flops, 'dtype': 'f32',
                'note': 'tables on demand: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains) on (32 vertices x 32 pixels) blocks, '
                        'computed by the wave that needs them; NEMO_SOLVE_LAZY=0 computes every block of every job in '
                        'bank_table_kernel instead (2 V h w D flops per job; 22 ms = 0.65 of the fp32 matrix peak, '
                        'profiles/r03_bench_whole_tables.json)'}
        else:
            per_tab = _pmc(PMC_TABLES_FILE, 'bank_table_kernel', 'hbm_bytes_per_job')
            tf = whole_flops / (table_ms * 1e-3) / 1e12
            extra['roofline_table'] = {
                'bound': 'mfma', 'achieved': tf, 'peak': MFMA_FP32_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': tf / MFMA_FP32_PEAK_TFLOPS,
                'traffic': None if per_tab is None else per_tab * args.images, 'kernel': 'bank_table_kernel', 'kernel_ms': table_ms,
                'flops_per_launch': whole_flops, 'dtype': 'f32',
                'note': 'second kernel of the solve: [V x D] . [D x h w] per job on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains), '
                        '2 V h w D flops per job; its traffic is the table it writes (V h w 4 bytes per job) plus the maps it reads once'}
        extra['roofline_streaming_equivalent'] = {
            'bound': 'hbm', 'achieved': stream_bytes / ((table_ms + loop_ms) * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': stream_bytes / ((table_ms + loop_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
            'kernel': 'solve_kernel<TAB, LAZY>' if form == 2 else 'bank_table_kernel + solve_kernel<TAB>',
            'kernel_ms': table_ms + loop_ms, 'algorithmic_bytes_per_launch': stream_bytes,
            'note': 'continuity with rounds 1-2 (frac 0.63-0.65): SURVEY 8(d)\'s per-unit figure, n_cov (4 D + 4) bytes per job-step '
                    '-- what the streaming formulation of the solve must move -- divided by the time of the kernel(s) that '
                    'replace it.  Above 1: those bytes are no longer moved; the contraction they fed is done once per '
                    '(vertex, pixel) block instead of once per step'}
    else:
        achieved = stream_bytes / (solve_ms * 1e-3) / 1e9
        roof = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS, 'traffic': pmc_traffic(cov_steps), 'kernel': 'solve_kernel',
                'traffic_source': f'scaled from the committed PMC pass {PMC_FILE} (not read during this run)',
                'kernel_ms': solve_ms, 'algorithmic_bytes_per_launch': stream_bytes,
                'covered_pixel_steps_per_launch': cov_steps,
                # what the resident jobs (one workgroup per CU) re-read every Adam step, against the 256 MiB
                # Infinity Cache: FETCH_SIZE counts L2 misses whether the Infinity Cache or HBM serves them,
                # so `frac` (algorithmic bytes / time) is an upper bound on DRAM utilisation
                'working_set_bytes': working_set, 'infinity_cache_bytes': 256 * 2 ** 20}
    roof['solve_form'] = {0: 'streaming', 1: 'whole tables', 2: 'tables on demand'}.get(form, str(form))
    roof['solve_ms'] = solve_ms

    if rank == 0:
        ok = torch.isfinite(res.campos).all(dim=1).float().mean().item()
        # sanity: median geodesic error of the solves against the synthetic ground truth
        from nemo_amd.camera_params import pose_error
        th, el, az, ds = (t.cpu().numpy() for t in (res.theta, res.elevation, res.azimuth, res.distance))
        errs = [pose_error((th[i], el[i], az[i], ds[i]), tuple(wl.gt[i])) for i in range(0, args.images, max(1, args.images // 64))]
        out = {
            'metric': 'pose solves/sec (300 Adam iters, whole node)',
            'value': world * args.images * args.steps / dt,
            'unit': 'solves/s',
            'n_gpus': world if not args.share_gpu else 1,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': 1e3 * dt / args.steps,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': wl.name, 'images_per_gpu': args.images, 'feature_map': [eng.h, eng.w, eng.D],
                       'raster': eng.S, 'grid_poses': int(solver.grid_theta.shape[1]),
                       'adam_steps': solver.cfg.total_epochs, 'parallelism': f'images sharded over {world} GPU(s)'
                       + (f' [REHEARSAL: {world} ranks share one GPU, gloo]' if args.share_gpu else '')},
            'roofline': roof,
            'quality': {'finite_fraction': ok, 'median_err_deg_vs_synthetic_gt': float(np.degrees(np.median(errs)))},
        }
        out.update(extra)
        if world == 1:
            out['roofline_mfma'] = roofline_mfma(eng, wl.feats)
        if world == 1 and args.from_files_images > 0:
            del res
            ff = from_files_leg(args.from_files_images, dev, args.from_files_batch)
            out['from_files'] = ff
            out['h2d_gbs'] = ff['h2d_gbs']
            out['from_files_solves_per_s'] = ff['solves_per_s']
        if world == 1 and not args.no_cpu_baseline:
            n_sample = args.cpu_sample or 3 * min(16, os.cpu_count() or 1)
            out['cpu_baseline'] = cpu_baseline(wl, n_sample)
            out['cpu_baseline_autograd'] = cpu_autograd
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
