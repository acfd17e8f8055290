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

import numpy as np
import torch

from . import records
from . import synthetic as syn
from .mesh_utils import load_off
from .pose_solver import BatchPoseSolver, SolverConfig


def build_parser():
    p = argparse.ArgumentParser(description='NeMo pose estimation (MI355X batched path)')
    p.add_argument('--type_', default='car', type=str)
    p.add_argument('--mesh_d', default='build', type=str)
    p.add_argument('--turn_off_clutter', default=False, type=lambda s: str(s).lower() in ('1', 'true', 'yes'))
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
              adam_beta_0=args.adam_beta_0, adam_beta_1=args.adam_beta_1, turn_off_clutter=args.turn_off_clutter)
    cfg = SolverConfig.unlabeled_6dof(**kw) if args.six_dof else SolverConfig(**kw)
    solver = BatchPoseSolver(cfg, device)
    clutter = []
    for k, st in enumerate(subtypes):                                              # :199-207
        xvert, xface = load_off(os.path.join(mesh_dir, '%02d.off' % (k + 1)))
        solver.add_mesh(xvert, xface, rec.memory(st), syn.DISTANCE_RENDER[cate])
        # the clutter vector is per subtype: `clutter_<subtype>` is read inside the subtype loop (:203,207)
        clutter.append(solver.clutter_vector(torch.from_numpy(rec.clutter(st)).to(solver.device)))
    return solver, clutter


def solve_shard(args, rec, subtypes, names, mesh_of, device, stats=None):
    """Clutter score (per subtype) -> grid init -> fused solve for the listed images: [len(names), 10] records.

    The images go through in batches of --batch: while batch k is solved, a helper thread reads batch k + 1 from
    the record into pinned memory and uploads it on a side stream (records.upload_pixel_major; DMA copies only, the
    one transpose kernel of a batch runs when the solve before it has left the CUs), so the file path costs
    max(read + upload, solve) per batch instead of their sum; the first batch comes in under mesh registration and
    the grid pre-render.  Batches do not change results (every job is independent)."""
    from concurrent.futures import ThreadPoolExecutor
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    n, B = len(names), max(1, int(getattr(args, 'batch', 256) or 256))
    side = torch.cuda.Stream(dev)

    def load(lo, hi):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(side):
            feats = rec.upload_pixel_major(names[lo:hi], dev, stats=stats)
            ready = torch.cuda.Event()
            ready.record(side)
        return feats, ready

    out, clus, inits = [], [], []
    with ThreadPoolExecutor(max_workers=1) as helper:
        nxt = helper.submit(load, 0, min(B, n))  # the first batch comes in while the meshes are registered and pre-rendered
        solver, clutter = build_solver(args, rec, subtypes, dev)
        solver.engine.reserve(min(B, n), int(solver.grid_theta.shape[1]))
        solver.prerender()
        for lo in range(0, n, B):
            hi = min(lo + B, n)
            feats, ready = nxt.result()
            if hi < n:
                nxt = helper.submit(load, hi, min(hi + B, n))
            main = torch.cuda.current_stream(dev)
            main.wait_event(ready)
            feats.record_stream(main)
            clu = None if solver.cfg.turn_off_clutter else solver.clutter_scores_by_group(feats, mesh_of[lo:hi], clutter)
            mesh_idx = torch.tensor(mesh_of[lo:hi], dtype=torch.int32, device=dev)
            res = solver.solve_all(feats, None, mesh_idx, clutter_scores=clu)
            t2 = res.trans_2d if res.trans_2d is not None else torch.zeros((hi - lo, 2), device=dev)
            out.append(torch.cat([res.packed(), t2], dim=1))
            clus.append(clu)
            inits.append(res.init_index)
    info = dict(solver=solver, init_index=torch.cat(inits), clu=None if clus[0] is None else torch.cat(clus))
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
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl' if torch.cuda.is_available() else 'gloo')
    done = {}
    if args.save_final_pred and not args.no_reload and os.path.exists(args.save_final_pred):  # PredUnlabeledDataset.py:297-303
        done = load_final_pred(args.save_final_pred)
    names, _, recs = solve_category(args, 'cuda:%d' % local, skip=set(done))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    if rank != 0:
        return 0
    anno_dir = args.anno_path % args.type_ if '%s' in args.anno_path else args.anno_path
    if args.save_final_pred:                                                       # PredUnlabeledDataset.py:344-352
        os.makedirs(os.path.dirname(os.path.abspath(args.save_final_pred)), exist_ok=True)
        preds = recs[:, [0, 3, 1, 2, 8, 9]]                                        # distance, theta, elevation, azimuth, t0, t1
        done.update({n: preds[i] for i, n in enumerate(names)})
        np.savez(args.save_final_pred, **done)
    total_error, cads = errors_from_records(names, recs, anno_dir)
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
