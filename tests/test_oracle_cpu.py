"""CPU tier (no GPU): the oracle against its pins.

* metric: PINNED by the reference (vectors produced by importing the reference's own
  lib/ProcessCameraParameters.py, tests/golden/make_golden.py).
* pose path: PARITY UNPINNED by the reference (no fixtures exist; PyTorch3D not runnable) --
  the hand-derived backward is checked against torch.autograd on the same op sequence and
  against central finite differences; Adam against torch.optim.Adam; everything against the
  committed oracle fixture so the oracle cannot drift silently.
"""
import math
import os

import numpy as np
import pytest

from nemo_amd import synthetic as syn
from nemo_amd.camera_params import cal_err, cal_rotation_matrix, get_transformation_matrix
from tests.scenes import make_scene, random_poses

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def test_metric_matches_reference_vectors():
    g = np.load(os.path.join(GOLD, 'metric_vectors.npz'))
    for i in range(len(g['azimuth'])):
        m = get_transformation_matrix(g['azimuth'][i], g['elevation'][i], g['distance'][i])
        assert np.allclose(m, g['matrices'][i], rtol=0, atol=1e-12)
    for k, i in enumerate(range(0, len(g['azimuth']), 2)):
        a = cal_rotation_matrix(g['theta'][i], g['elevation'][i], g['azimuth'][i], g['distance'][i])
        b = cal_rotation_matrix(g['theta'][i + 1], g['elevation'][i + 1], g['azimuth'][i + 1], g['distance'][i + 1])
        assert abs(np.real(cal_err(a, b)) - g['pair_errors'][k]) < 1e-9


def test_sincos_close_to_libm(oracle):
    xs = np.concatenate([np.linspace(-7, 7, 4001), np.random.default_rng(0).uniform(-0.6, 0.6, 4000)]).astype(np.float32)
    for x in xs[::7]:
        s, c = oracle.sincos(x)
        assert abs(s - math.sin(float(x))) <= 2.5e-7 and abs(c - math.cos(float(x))) <= 2.5e-7


def test_camera_matches_torch_restatement(oracle):
    import torch
    from oracle import torch_ref
    rng = np.random.default_rng(1)
    C, th = random_poses(rng, 32, 5.0, 0.5)
    R64, T64 = torch_ref.campos_to_R_T(torch.tensor(C, dtype=torch.float64), torch.tensor(th, dtype=torch.float64))
    for i in range(32):
        R, T = oracle.campos_to_R_T(C[i], th[i])
        assert np.abs(R - R64[i].numpy()).max() < 2e-6 and np.abs(T - T64[i].numpy()).max() < 5e-6
        assert np.abs(R @ R.T - np.eye(3)).max() < 1e-5


def test_raster_invariants(oracle):
    sc = make_scene(n_img=1)
    rng = np.random.default_rng(2)
    C, th = random_poses(rng, 6, sc.distance)
    for i in range(6):
        R, T = oracle.campos_to_R_T(C[i], th[i])
        scr = oracle.project(sc.verts, R, T, sc.cfg.ndc_scale)
        p2f, bary, zb = oracle.rasterize(sc.cfg, scr, sc.faces)
        cov = p2f >= 0
        assert cov.sum() > 50
        assert np.all(bary[cov] > 0) and np.abs(bary[cov].sum(-1) - 1).max() < 1e-4
        assert np.all(bary[~cov] == -1) and np.all(p2f[~cov] == -1)
        assert np.all(zb[cov] > 0)
        # crop equivalence: the window of the full raster is the cropped output
        pf, bf, _ = oracle.rasterize(sc.cfg, scr, sc.faces, crop=False)
        g = sc.geo
        assert np.array_equal(pf[g.r0:g.r0 + g.h, g.c0:g.c0 + g.w], p2f)
        # interpolation of a constant bank is constant on covered pixels
        ones = np.ones_like(sc.bank)
        out = oracle.interpolate(sc.cfg, p2f, bary, sc.faces, ones)
        assert np.abs(out[cov] - 1).max() < 1e-4 and np.all(out[~cov] == 0)


def _window(sc, a):
    from oracle import raster_np
    return np.ascontiguousarray(raster_np.crop(a, sc.cfg.S, sc.cfg.h, sc.cfg.w, sc.cfg.r0, sc.cfg.c0))


K_EPS = [1e-8, 1e-30]  # PyTorch3D's kEpsilon: 1e-8 in current sources, 1e-30 in the 0.2.0 era


@pytest.mark.parametrize('k_eps', K_EPS)
@pytest.mark.parametrize('persp', [False, True])
@pytest.mark.parametrize('cate,hw,nv', [('car', (256, 384), 300), ('bottle', (384, 256), 200), ('aeroplane', (640, 800), 1000)])
def test_two_independent_rasterizers_agree(oracle, cate, hw, nv, persp, k_eps):
    """The discrete half of the path has two lineages: the C oracle's per-pixel loop and oracle/raster_np.py
    (float32 array operations over all pixels x all faces, written from SURVEY 8a row P4, sharing no code with the
    C file).  pix_to_face, barycentrics and z of the full S x S image agree bit for bit, and so does the crop."""
    from oracle import raster_np
    sc = make_scene(cate, hw, nv, n_img=1, perspective_correct=persp, k_eps=k_eps)
    rng = np.random.default_rng(12)
    C, th = random_poses(rng, 4, sc.distance, spread=0.3)
    for i in range(4):
        R, T = oracle.campos_to_R_T(C[i], th[i])
        scr = oracle.project(sc.verts, R, T, sc.cfg.ndc_scale)
        pf, bf, zf = oracle.rasterize(sc.cfg, scr, sc.faces, crop=False)
        p2, b2, z2 = raster_np.rasterize(scr, sc.faces, sc.cfg.S, persp, sc.cfg.k_eps)
        assert (pf >= 0).sum() > 50
        assert np.array_equal(pf, p2) and np.array_equal(bf, b2) and np.array_equal(zf, z2)
        pw, bw, _ = oracle.rasterize(sc.cfg, scr, sc.faces)
        assert np.array_equal(pw, _window(sc, p2)) and np.array_equal(bw, _window(sc, b2))


@pytest.mark.parametrize('k_eps', K_EPS)
@pytest.mark.parametrize('persp', [False, True])
def test_two_independent_rasterizers_agree_on_triangle_soups(oracle, persp, k_eps):
    """Same, on the inputs that exercise the decision rules: duplicated faces (ties -> lowest index), repeated
    vertices and collinear faces (kEps skip), a triangle larger than the view in both windings, interpenetrating
    slivers, vertices behind the camera (pz < 0)."""
    from oracle import raster_np
    S, h, w, D = 48, 32, 48, 128
    s = syn.ndc_scale(12.0)
    cfg = oracle.make_cfg(S, h, w, D, s, perspective_correct=persp, k_eps=k_eps)
    ties = 0
    for seed in range(6):
        rng = np.random.default_rng(100 + seed)
        V = 120
        verts = (rng.uniform(-0.5, 0.5, (V, 3)) * np.array([1.0, 0.6, 1.0])).astype(np.float32)
        faces = rng.integers(0, V, (260, 3)).astype(np.int32)
        faces[40:60] = faces[0:20]
        faces[60:70, 2] = faces[60:70, 0]
        verts[100:104] = np.array([[-3, -3, 0.2], [3, -3, 0.2], [0, 4, 0.2], [0, 0, -0.3]], np.float32) * (1 + 0.1 * seed)
        faces[70] = (100, 101, 102)
        faces[71] = (100, 102, 101)
        faces[72] = (100, 101, 103)
        verts[104:107] = verts[10] + np.outer(np.arange(3), [0.1, 0.05, 0.0]).astype(np.float32)
        faces[73] = (104, 105, 106)
        C, th = random_poses(rng, 4, 5.0, spread=0.3)
        C[3] = C[3] * 0.08
        for i in range(4):
            R, T = oracle.campos_to_R_T(C[i], th[i])
            scr = oracle.project(verts, R, T, s)
            pf, bf, zf = oracle.rasterize(cfg, scr, faces, crop=False)
            p2, b2, z2 = raster_np.rasterize(scr, faces, S, persp, cfg.k_eps)
            assert np.array_equal(pf, p2), (seed, i, int((pf != p2).sum()))
            cov = pf >= 0
            assert np.array_equal(bf[cov], b2[cov]) and np.array_equal(zf[cov], z2[cov])
            ties += int(np.isin(pf[cov], np.arange(0, 20)).sum())
            assert not np.isin(pf[cov], np.arange(40, 60)).any()     # a duplicate never beats the original
    assert ties > 0


@pytest.mark.parametrize('persp,clutter', [(False, True), (True, True), (False, False)])
def test_backward_vs_autograd_and_finite_differences(oracle, persp, clutter):
    from oracle import torch_ref
    sc = make_scene(n_img=1, perspective_correct=persp, clutter=clutter)
    rng = np.random.default_rng(3)
    C, th = random_poses(rng, 3, sc.distance)
    az, el, t, d = sc.gt[0]
    C[0] = syn.spherical_to_campos(d * 1.02, el + 0.03, az - 0.04); th[0] = t + 0.02
    clu = sc.clus[0] if clutter else None
    from oracle import raster_np
    for i in range(3):
        e = oracle.evaluate(sc.cfg, sc.verts, sc.faces, sc.bank, sc.feats[0], clu, C[i], th[i], want_maps=True)
        # coverage for the autograd restatement comes from the independent array rasterizer, not from the C oracle
        R, T = oracle.campos_to_R_T(C[i], th[i])
        p2f = _window(sc, raster_np.rasterize(oracle.project(sc.verts, R, T, sc.cfg.ndc_scale), sc.faces, sc.cfg.S, persp, sc.cfg.k_eps)[0])
        assert np.array_equal(p2f, e['pix_to_face'])
        a = torch_ref.loss_and_grads(sc.cfg, sc.verts, sc.faces, sc.bank, sc.feats[0], clu, C[i], th[i], p2f)
        assert abs(e['loss'] - a['loss']) < 1e-6
        ga = np.r_[a['gC'], a['gtheta']]; go = np.r_[e['gC'], e['gtheta']]
        assert np.abs(go - ga).max() <= 1e-4 * np.abs(ga).max()
        # finite differences of the (float64) torch loss with coverage frozen
        eps = 1e-6
        for k in range(4):
            dp = np.zeros(4); dp[k] = eps
            lp = torch_ref.loss_and_grads(sc.cfg, sc.verts, sc.faces, sc.bank, sc.feats[0], clu, C[i] + dp[:3], th[i] + dp[3], p2f)['loss']
            lm = torch_ref.loss_and_grads(sc.cfg, sc.verts, sc.faces, sc.bank, sc.feats[0], clu, C[i] - dp[:3], th[i] - dp[3], p2f)['loss']
            fd = (lp - lm) / (2 * eps)
            assert abs(fd - ga[k]) <= 2e-3 * np.abs(ga).max() + 1e-9


def test_extra_trans_gradient(oracle):
    from oracle import torch_ref
    sc = make_scene(n_img=1)
    C, th = random_poses(np.random.default_rng(4), 1, sc.distance)
    ex = np.array([0.05, -0.03, 0.0], np.float32)
    e = oracle.evaluate(sc.cfg, sc.verts, sc.faces, sc.bank, sc.feats[0], sc.clus[0], C[0], th[0], extra=ex, want_maps=True)
    a = torch_ref.loss_and_grads(sc.cfg, sc.verts, sc.faces, sc.bank, sc.feats[0], sc.clus[0], C[0], th[0], e['pix_to_face'], extra=ex)
    assert np.abs(e['gextra'] - a['gextra']).max() <= 1e-4 * np.abs(a['gextra']).max()


def test_adam_matches_torch_optim(oracle):
    from oracle import torch_ref
    rng = np.random.default_rng(5)
    grads = rng.normal(0, 1e-2, (300, 4)).astype(np.float32)
    p0 = np.array([1.0, 2.0, 3.0, 0.1], np.float32)
    ref = torch_ref.adam_reference(p0, grads)
    tab = syn.adam_table(300)
    p, m, v = p0.copy(), np.zeros(4, np.float32), np.zeros(4, np.float32)
    for t in range(300):
        p, m, v = oracle.adam_step(p, m, v, grads[t], 0.4, 0.6, tab[t, 0], tab[t, 1])
        assert np.abs(p - ref[t]).max() <= 2e-6
    # lr x0.2 after steps 100 and 200 (code/MeshPoseSolveAll.py:249-250)
    assert abs(tab[100, 0] * (1 - 0.4 ** 101) / (tab[99, 0] * (1 - 0.4 ** 100)) - 0.2) < 1e-6


def test_oracle_matches_committed_fixture(oracle):
    g = np.load(os.path.join(GOLD, 'path_small.npz'))
    geo = syn.Geometry.from_image_size(tuple(g['image_hw']))
    cfg = oracle.make_cfg(geo.S, geo.h, geo.w, 128, syn.ndc_scale(12.0), geo.r0, geo.c0)
    for i in range(4):
        e = oracle.evaluate(cfg, g['verts'], g['faces'], g['bank'], g['feat'], g['clu'], g['poses_C'][i], g['poses_theta'][i], want_maps=True)
        assert np.array_equal(e['pix_to_face'], g['pix_to_face'][i]) and np.array_equal(e['bary'], g['bary'][i])
        assert e['loss'] == g['loss'][i] and e['ncov'] == g['ncov'][i]
        assert np.array_equal(np.r_[e['gC'], e['gtheta']].astype(np.float32), g['grad'][i])
    gC, gth = syn.pose_grid(float(g['distance']))
    best, losses = oracle.grid_search(cfg, g['verts'], g['faces'], g['bank'], g['feat'], g['clu'], gC, gth)
    assert best == int(g['grid_best']) and np.array_equal(losses, g['grid_losses'])
    oC, oth, ol, _ = oracle.solve(cfg, g['verts'], g['faces'], g['bank'], g['feat'], g['clu'], gC[best], gth[best], g['adam_tab'])
    assert np.array_equal(oC, g['final_C']) and np.float32(oth) == g['final_theta']


def test_solve_recovers_synthetic_pose(oracle):
    from nemo_amd.camera_params import pose_error
    g = np.load(os.path.join(GOLD, 'path_small.npz'))
    C = g['final_C'].astype(np.float64)
    dist = np.sqrt((C ** 2).sum()); azim = np.arctan(C[0] / C[2]) % np.pi + np.pi * (C[0] < 0); elev = np.arcsin(C[1] / dist)
    az, el, th, d = g['gt']
    assert np.degrees(pose_error((float(g['final_theta']), elev, azim, dist), (th, el, az, d))) < 3.0


def test_solve6_oracle_gradients_and_descent():
    """6-DoF restatement (code/PredUnlabeledDataset.py:318-339): the gradient w.r.t. trans_2d is dL/dT's first
    two components (checked against torch autograd through extra_trans), and with trans_2d frozen at zero
    gradient weight the first iterate equals the 4-DoF one."""
    from oracle import oracle as orc, torch_ref
    sc = make_scene(cate='car', image_hw=(128, 192), number_vertices=150, n_img=1, seed=5)
    az, el, th, d = sc.gt[0]
    C0 = syn.spherical_to_campos(d * 1.04, el + 0.05, az - 0.06)
    ex = np.array([0.03, -0.02, 0.0], np.float32)
    e = orc.evaluate(sc.cfg, sc.verts, sc.faces, sc.bank, sc.feats[0], sc.clus[0], C0, th + 0.02, extra=ex, want_maps=True)
    a = torch_ref.loss_and_grads(sc.cfg, sc.verts, sc.faces, sc.bank, sc.feats[0], sc.clus[0], C0, th + 0.02,
                                 e['pix_to_face'].reshape(sc.cfg.h, sc.cfg.w), extra=ex)
    assert abs(e['loss'] - a['loss']) < 1e-5
    assert np.abs(e['gextra'] - a['gextra']).max() <= 1e-4 * max(1e-3, np.abs(a['gextra']).max())
    tab = syn.adam_table(40)
    p6, l6, tr6 = orc.solve6(sc.cfg, sc.verts, sc.faces, sc.bank, sc.feats[0], sc.clus[0], C0, th + 0.02, tab, want_trace=True)
    _, _, _, tr4 = orc.solve(sc.cfg, sc.verts, sc.faces, sc.bank, sc.feats[0], sc.clus[0], C0, th + 0.02, tab, want_trace=True)
    assert np.array_equal(tr6[0, 0:4], tr4[0, 0:4]) and tr6[0, 6] == tr4[0, 4]
    assert np.array_equal(tr6[0, 7:11], tr4[0, 5:9])          # same first gradient in (C, theta)
    assert tr6[-1, 6] < tr6[0, 6]                               # it descends


def test_rasterizer_lineages_agree_on_random_soups_hypothesis(oracle):
    """Property test (hypothesis): for random small triangle soups -- random vertices, random faces, random raster
    size, either perspective mode, the vertices squeezed towards the view so that faces overlap, interpenetrate and
    cross the image border -- the C oracle and the independent array rasterizer give the same owner for every
    pixel, the same barycentrics and the same depth."""
    from hypothesis import given, settings, strategies as st
    from oracle import raster_np

    @settings(max_examples=40, deadline=None, derandomize=True)
    @given(seed=st.integers(0, 2 ** 31 - 1), S=st.sampled_from([8, 17, 32]), persp=st.booleans(), nf=st.integers(1, 40))
    def check(seed, S, persp, nf):
        rng = np.random.default_rng(seed)
        V = 24
        scr = np.concatenate([rng.uniform(-1.3, 1.3, (V, 2)), rng.uniform(-0.5, 6.0, (V, 1))], 1).astype(np.float32)
        scr[rng.integers(0, V)] = scr[rng.integers(0, V)]                     # a repeated vertex position now and then
        faces = rng.integers(0, V, (nf, 3)).astype(np.int32)
        if nf > 3:
            faces[-1] = faces[0]                                               # an exact duplicate: ties
        cfg = oracle.make_cfg(S, S, S, 128, 9.5, perspective_correct=persp)
        pf, bf, zf = oracle.rasterize(cfg, scr, faces, crop=False)
        p2, b2, z2 = raster_np.rasterize(scr, faces, S, persp, cfg.k_eps)
        assert np.array_equal(pf, p2)
        cov = pf >= 0
        assert np.array_equal(bf[cov], b2[cov]) and np.array_equal(zf[cov], z2[cov])
        assert np.all(bf[~cov] == -1) and np.all(b2[~cov] == -1)

    check()


def test_reference_op_sequence_baseline_matches_the_oracle(oracle):
    """oracle/ref_sequence.py (the CPU baseline of SURVEY 8d: the reference's op sequence with torch.autograd and
    torch.optim.Adam) computes what the C oracle computes: same rendered map, same loss, same dL/dC, dL/dtheta at a
    pose, and its Adam steps follow the oracle's solve."""
    import torch
    from oracle.ref_sequence import ReferenceSequence
    sc = make_scene(n_img=1, seed=33)
    seq = ReferenceSequence(sc)
    C, th = random_poses(np.random.default_rng(4), 1, sc.distance)
    e = oracle.evaluate(sc.cfg, sc.verts, sc.faces, sc.bank, sc.feats[0], sc.clus[0], C[0], th[0], want_maps=True)
    Ct = torch.tensor(C[:1], requires_grad=True)
    tht = torch.tensor(th[:1], requires_grad=True)
    out = seq.render(Ct, tht)                                            # [D,h,w]
    ref = oracle.interpolate(sc.cfg, e['pix_to_face'], e['bary'], sc.faces, sc.bank)
    assert np.abs(out.detach().permute(1, 2, 0).numpy() - ref).max() <= 2e-6
    loss = seq.loss_fun(torch.sum(out * seq.predicted_map, dim=0))
    loss.backward()
    assert abs(loss.item() - e['loss']) <= 1e-5 * abs(e['loss'])
    g = np.r_[Ct.grad[0].numpy(), tht.grad.numpy()]
    want = np.r_[e['gC'], e['gtheta']]
    assert np.abs(g - want).max() <= 2e-4 * np.abs(want).max()
    tab = syn.adam_table(4)
    oC, oth, _, _ = oracle.solve(sc.cfg, sc.verts, sc.faces, sc.bank, sc.feats[0], sc.clus[0], C[0], th[0], tab)
    rC, rth, losses = seq.adam_steps(C[0], th[0], 4)
    assert np.abs(rC - oC).max() <= 1e-4 and abs(rth - oth) <= 1e-4 and len(losses) == 4
