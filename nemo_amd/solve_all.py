#!/usr/bin/env python3
"""Batched stand-in for `python code/MeshPoseSolveAll.py` (one category per call).

Same flags where they mean the same thing (code/MeshPoseSolveAll.py:8-28), same inputs (the feature .npz
of ExtractFeatures.py, the category's CAD .off meshes, per-image annotation .npz) and the same outputs
(`--save_accuracy` record for CalAccuracy.py, or the Thr / Med lines on stdout, :273-287).  The per-image
Python loop (:217-271) is replaced by clutter score -> pre-rendered grid search -> fused solve over all
images of the category at once; under `python -m torch.distributed.run` the images are sharded over ranks
and the [n, 10] pose records come back through ONE all-gather (nemo_amd.distributed.gather_records; names go by
index), which replaces InferenceNeMo.sh's per-category processes + file merge; rank 0 scores them.

`python -m nemo_amd.solve_all --accuracy_table DIR` is code/CalAccuracy.py.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch

from . import records
from . import synthetic as syn
from .mesh_utils import load_off
from .pose_solver import BatchPoseSolver, SolverConfig


def _flag(s):
    """The reference declares its switches `type=bool` (any non-empty string is True); here the text is read."""
    return str(s).strip().lower() in ('1', 'true', 'yes', 'y', 'on')


def build_parser():
    p = argparse.ArgumentParser(description='NeMo pose estimation (MI355X batched path)')
    p.add_argument('--type_', default='car', type=str)
    p.add_argument('--mesh_d', default='build', type=str)
    p.add_argument('--turn_off_clutter', default=False, type=_flag)
    p.add_argument('--data_pendix', default='', type=str)
    p.add_argument('--feature_path', default='saved_features', type=str)
    p.add_argument('--feature_name', default='3D512_points1saved_model_%s_799.pth', type=str)
    p.add_argument('--record_file', default='', type=str, help='explicit feature .npz (overrides feature_path/name)')
    p.add_argument('--save_accuracy', default='', type=str)
    p.add_argument('--mesh_path', default='../PASCAL3D/PASCAL3D+_release1.1/CAD_%s/%s/', type=str)
    p.add_argument('--anno_path', default='../data/PASCAL3D_NeMo/annotations/%s/', type=str)
    p.add_argument('--total_epochs', default=300, type=int)
    p.add_argument('--lr', default=5e-2, type=float)
    p.add_argument('--adam_beta_0', default=0.4, type=float)
    p.add_argument('--adam_beta_1', default=0.6, type=float)
    p.add_argument('--image_size', default='', type=str, help='H,W override of the category table (:142-145)')
    p.add_argument('--feat_dim', default=128, type=int)
    p.add_argument('--six_dof', action='store_true', help='PredUnlabeledDataset.py variant; writes --save_final_pred')
    p.add_argument('--save_final_pred', default='', type=str)
    p.add_argument('--no_reload', action='store_true',
                   help='with --save_final_pred: solve every image again instead of keeping the entries the file already holds')
    p.add_argument('--accuracy_table', default='', type=str, help='directory of per-category records: print CalAccuracy table')
    p.add_argument('--batch', default=256, type=int, help='images per solve launch; the next batch is read and uploaded meanwhile')
    p.add_argument('--streams', default=2, type=int,
                   help='solver contexts, each on a stream of its own, taking the batches in turn: the next batch starts on the '
                        'compute units the previous one leaves, instead of behind its slowest job')
    # flags of code/MeshPoseSolveAll.py:12-14,20 kept so that its command lines run unchanged
    p.add_argument('--pre_render', default=True, type=_flag,
                   help=':14,209-212 -- False: no stored grid renders, every (image, grid pose) is rendered on the fly (get_init_pos)')
    p.add_argument('--objectnet', default=False, type=_flag, help=':12 -- parsed and never read by the reference; accepted, ignored')
    p.add_argument('--record_pendix', default='', type=str, help=':13 -- parsed and never read by the reference; accepted, ignored')
    p.add_argument('--mesh_path_ref', default='../PASCAL3D/PASCAL3D+_release1.1/CAD/%s/', type=str,
                   help=':20,131,176 -- the reference only counts its entries (one error list per CAD model); here the count '
                        'bounds the cad_index of the annotations when the directory exists')
    p.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                   help='process group under torch.distributed.run: nccl = RCCL, one GPU per rank; gloo = rehearsal on fewer GPUs')
    p.add_argument('--share_gpu', action='store_true', help='rehearsal only (needs --backend gloo): every rank uses GPU 0')
    return p


def record_path(args):
    """code/MeshPoseSolveAll.py:156-167."""
    if args.record_file:
        return args.record_file
    name = args.feature_name % (args.type_, args.mesh_d) if args.feature_name.count('%s') == 2 else args.feature_name
    if len(args.data_pendix) > 0:
        return os.path.join(args.feature_path, args.type_ + '_occ', args.data_pendix + '_' + name)
    return os.path.join(args.feature_path, args.type_, name)


RECORD_WIDTH = 10  # SolveResult.packed() (dist, elev, azim, theta, loss, init, status, pad) + trans_2d (6-DoF, else 0)


def category_jobs(rec, subtypes):
    """Image names and their subtype index, in record order (code/MeshPoseSolveAll.py:199,217)."""
    names, mesh_of = [], []
    for k, st in enumerate(subtypes):
        for n in rec.names(st):
            names.append(n)
            mesh_of.append(k)
    return names, mesh_of


def build_solver(args, rec, subtypes, device):
    """Cameras, rasterizer settings, one mesh + bank + pose grid + clutter vector per subtype (:180-207)."""
    cate = args.type_
    image_size = tuple(int(v) for v in args.image_size.split(',')) if args.image_size else syn.IMAGE_SIZES[cate]
    mesh_dir = args.mesh_path % (args.mesh_d, cate) if args.mesh_path.count('%s') == 2 else args.mesh_path
    kw = dict(image_size=image_size, feat_dim=args.feat_dim, total_epochs=args.total_epochs, lr=args.lr,
              adam_beta_0=args.adam_beta_0, adam_beta_1=args.adam_beta_1, turn_off_clutter=args.turn_off_clutter,
              pre_render=bool(getattr(args, 'pre_render', True)))
    cfg = SolverConfig.unlabeled_6dof(**kw) if args.six_dof else SolverConfig(**kw)
    solver = BatchPoseSolver(cfg, device)
    clutter = []
    for k, st in enumerate(subtypes):                                              # :199-207
        xvert, xface = load_off(os.path.join(mesh_dir, '%02d.off' % (k + 1)))
        solver.add_mesh(xvert, xface, rec.memory(st), syn.DISTANCE_RENDER[cate])
        # the clutter vector is per subtype: `clutter_<subtype>` is read inside the subtype loop (:203,207)
        clutter.append(solver.clutter_vector(torch.from_numpy(rec.clutter(st)).to(solver.device)))
    return solver, clutter


def batch_plan(n, batch):
    """[(lo, hi)] of the launches: two quarter batches and a half batch first, so that the compute units have work a
    quarter batch after the first byte is read instead of a whole one, then full batches."""
    B = max(1, int(batch))
    sizes, lo, plan = [max(1, B // 4), max(1, B // 4), max(1, B // 2)], 0, []
    while lo < n:
        step = sizes.pop(0) if sizes else B
        plan.append((lo, min(n, lo + step)))
        lo = plan[-1][1]
    return plan


def solve_shard(args, rec, subtypes, names, mesh_of, device, stats=None):
    """Clutter score (per subtype) -> grid init -> fused solve for the listed images: [len(names), 10] records.

    The images go through in batches (batch_plan): while earlier batches are solved, a helper thread reads the next one
    from the record into pinned memory and uploads it on a side stream (records.upload_pixel_major; DMA copies only,
    the one transpose kernel of a batch runs when compute units are free).  Batches alternate between --streams solver
    contexts, each with its own stream and workspaces: a solve is one workgroup per job for all its steps, so a launch
    ends with its slowest job while most compute units idle; the next batch's launches, queued on the other stream,
    take those units as they come free.  Batches do not change results (every job is independent)."""
    from concurrent.futures import ThreadPoolExecutor
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    n = len(names)
    plan = batch_plan(n, getattr(args, 'batch', 256) or 256)
    n_ctx = max(1, min(int(getattr(args, 'streams', 2) or 1), len(plan)))
    side = torch.cuda.Stream(dev)

    timed = stats is not None   # optional timeline (tools/runner_timeline.py): device times of every batch's phases
    main = torch.cuda.current_stream(dev)
    marks = []
    if timed:
        t_host0 = time.perf_counter()
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record(main)

    # the reader runs ahead of the launches by up to AHEAD batches (device memory: AHEAD + 2 batches of maps at most)
    import threading
    AHEAD = 3
    slots = threading.Semaphore(AHEAD)

    def load(lo, hi):
        slots.acquire()
        torch.cuda.set_device(dev)
        with torch.cuda.stream(side):
            feats = rec.upload_pixel_major(names[lo:hi], dev, stats=stats)
            ready = torch.cuda.Event(enable_timing=timed)
            ready.record(side)
        if timed:
            marks.append(('maps %d..%d queued (host)' % (lo, hi), time.perf_counter() - t_host0, None))
        return feats, ready

    out, clus, inits = [], [], []
    with ThreadPoolExecutor(max_workers=1) as helper:
        loads = [helper.submit(load, lo, hi) for lo, hi in plan]  # read from now on, while the meshes are registered and pre-rendered
        biggest = max(hi - lo for lo, hi in plan)
        ctxs = []
        for _ in range(n_ctx):
            solver, clutter = build_solver(args, rec, subtypes, dev)
            solver.engine.reserve(biggest, int(solver.grid_theta.shape[1]))
            solver.prerender()
            ctxs.append((solver, clutter, torch.cuda.Stream(dev) if n_ctx > 1 else main))
        for st in ctxs:
            st[2].wait_stream(main)      # set-up (registration, pre-render) ran on the current stream
        for bi, (lo, hi) in enumerate(plan):
            feats, ready = loads[bi].result()
            loads[bi] = None
            solver, clutter, stream = ctxs[bi % n_ctx]
            with torch.cuda.stream(stream):
                stream.wait_event(ready)
                feats.record_stream(stream)
                if timed:
                    b0 = torch.cuda.Event(enable_timing=True)
                    b0.record(stream)
                clu = None if solver.cfg.turn_off_clutter else solver.clutter_scores_by_group(feats, mesh_of[lo:hi], clutter)
                mesh_idx = torch.tensor(mesh_of[lo:hi], dtype=torch.int32, device=dev)
                res = solver.solve_all(feats, None, mesh_idx, clutter_scores=clu)
                t2 = res.trans_2d if res.trans_2d is not None else torch.zeros((hi - lo, 2), device=dev)
                out.append(torch.cat([res.packed(), t2], dim=1))
                clus.append(clu)
                inits.append(res.init_index)
                del feats
                slots.release()
                if timed:
                    b1 = torch.cuda.Event(enable_timing=True)
                    b1.record(stream)
                    marks.append(('batch %d..%d on stream %d' % (lo, hi, bi % n_ctx), time.perf_counter() - t_host0, (ready, b0, b1)))
        for st in ctxs:
            main.wait_stream(st[2])
    for t in out + inits + [c for c in clus if c is not None]:
        t.record_stream(main)
    if timed:
        torch.cuda.synchronize(dev)
        stats['timeline'] = [(label, 1e3 * th) + (tuple(ev0.elapsed_time(e) for e in evs) if evs else ())
                             for label, th, evs in marks]
    info = dict(solver=ctxs[0][0], init_index=torch.cat(inits), clu=None if clus[0] is None else torch.cat(clus))
    return torch.cat(out), info


def run_sharded(n_items, solve_fn, device):
    """The multi-GPU layout of the path (InferenceNeMo.sh:42-50,67-81 replaced): this rank's contiguous
    shard is solved by solve_fn(start, stop) -> [stop-start, W] records, one all-gather of the records returns
    all of them, in item order, on every rank.  No collective on the data path."""
    import torch.distributed as dist
    from .distributed import gather_records, shard_range
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    lo, hi = shard_range(n_items, rank, world)
    local = solve_fn(lo, hi) if hi > lo else torch.zeros((0, RECORD_WIDTH), dtype=torch.float32, device=device)
    return gather_records(local.to(device), n_items)


def load_final_pred(path):
    """Entries of an earlier final_pred.npz (code/PredUnlabeledDataset.py:297-298), read without unpickling."""
    with np.load(path, allow_pickle=False) as z:
        return {k: np.asarray(z[k], dtype=np.float64) for k in z.files}


def solve_category(args, device='cuda:0', skip=()):
    """-> (names, mesh index per name, [n, 10] records on the host) of the category's images, those named in `skip`
    left out (resume: code/PredUnlabeledDataset.py:301-303)."""
    cate = args.type_
    mesh_dir = args.mesh_path % (args.mesh_d, cate) if args.mesh_path.count('%s') == 2 else args.mesh_path
    n_mesh = len([f for f in os.listdir(mesh_dir) if f.endswith('.off')])
    subtypes = ['mesh%02d' % i for i in range(1, 1 + n_mesh)]                      # :172
    with records.FeatureRecord(record_path(args)) as rec:
        names, mesh_of = category_jobs(rec, subtypes)
        if skip:
            keep = [i for i, n in enumerate(names) if n not in skip]
            names, mesh_of = [names[i] for i in keep], [mesh_of[i] for i in keep]
        if not names:
            return names, mesh_of, np.zeros((0, RECORD_WIDTH), np.float32)
        recs = run_sharded(len(names), lambda lo, hi: solve_shard(args, rec, subtypes, names[lo:hi], mesh_of[lo:hi], device)[0],
                           device)
    return names, mesh_of, recs.cpu().numpy()


def errors_from_records(names, recs, anno_dir):
    """code/MeshPoseSolveAll.py:256-271 from the gathered records (columns: dist, elev, azim, theta)."""
    errors, cads = [], []
    if anno_dir and os.path.isdir(anno_dir):
        for i, n in enumerate(names):
            a = records.load_annotation(os.path.join(anno_dir, n + '.npz'))
            errors.append(records.rotation_error(a, recs[i, 3], recs[i, 1], recs[i, 2], recs[i, 0]))
            cads.append(a.cad_index)
    return errors, cads


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.accuracy_table:
        _, text = records.accuracy_table_from_dir(args.accuracy_table, args.data_pendix)
        print(text)
        return 0
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if args.share_gpu:
        if world > 1 and args.backend != 'gloo':
            print('solve_all: --share_gpu needs --backend gloo (RCCL refuses two ranks on one GPU)', file=sys.stderr)
            return 2
        local = 0
    if not torch.cuda.is_available():
        print('solve_all (rank %d): no GPU visible; the pose-solve path has no CPU fallback' % rank, file=sys.stderr)
        return 3
    torch.cuda.set_device(local)             # before any collective: RCCL binds the communicator to the current device
    if world > 1:
        import torch.distributed as dist
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group('gloo')
    done = {}
    if args.save_final_pred and not args.no_reload and os.path.exists(args.save_final_pred):  # PredUnlabeledDataset.py:297-303
        done = load_final_pred(args.save_final_pred)
    names, _, recs = solve_category(args, 'cuda:%d' % local, skip=set(done))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return 0
    anno_dir = args.anno_path % args.type_ if '%s' in args.anno_path else args.anno_path
    if args.save_final_pred:                                                       # PredUnlabeledDataset.py:344-352
        os.makedirs(os.path.dirname(os.path.abspath(args.save_final_pred)), exist_ok=True)
        preds = recs[:, [0, 3, 1, 2, 8, 9]].astype(np.float64)                     # distance, theta, elevation, azimuth, t0, t1
        if done:
            # resumed run: the accuracy below is the CATEGORY's, so the entries kept from the file are scored as
            # well -- their rows rebuilt from the stored predictions (loss / init are not in the file: NaN / 0)
            kept = [k for k in done if k not in set(names)]
            rows = np.zeros((len(kept), RECORD_WIDTH), dtype=np.float64)
            for i, k in enumerate(kept):
                d = np.asarray(done[k], dtype=np.float64).reshape(-1)
                rows[i, [0, 3, 1, 2]] = d[:4]
                rows[i, 4], rows[i, 6] = np.nan, float(np.all(np.isfinite(d[:4])))
                rows[i, 8:10] = d[4:6] if d.size >= 6 else 0.0
            names, recs = kept + list(names), np.concatenate([rows, recs.astype(np.float64)], 0)
        done.update({n: preds[i] for i, n in enumerate(names[len(names) - len(preds):])})
        np.savez(args.save_final_pred, **done)
    n_cad = None
    ref_dir = args.mesh_path_ref % args.type_ if '%s' in args.mesh_path_ref else args.mesh_path_ref
    if ref_dir and os.path.isdir(ref_dir):
        n_cad = len(os.listdir(ref_dir))                                           # :176: one error list per CAD model
    total_error, cads = errors_from_records(names, recs, anno_dir)
    if n_cad is not None and any(c < 1 or c > n_cad for c in cads):
        print('solve_all: an annotation names a cad_index outside 1..%d (%s)' % (n_cad, ref_dir), file=sys.stderr)
        return 4
    if args.save_accuracy:                                                         # :273-274
        records.save_category_errors(args.save_accuracy, total_error, cads)
    elif total_error:                                                              # :275-287
        te = np.asarray(total_error)
        for thr in (np.pi / 6, np.pi / 18):
            print('Thr: ', thr)
            print('Average\t%d\t%.3f' % (len(te), float(np.mean(te < thr))))
        print('Med')
        print('Average\t%d\t%.3f' % (len(te), float(180 / np.pi * np.median(te))))
    return 0


if __name__ == '__main__':
    sys.exit(main())
