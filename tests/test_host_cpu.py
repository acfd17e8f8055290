"""CPU tier: host logic and the C-ABI surface (no compute calls without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

from nemo_amd import synthetic as syn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cuboid_mesh_sizes_and_closedness():
    v, f = syn.class_mesh('car', 1000)
    assert v.shape == (1088, 3) and f.shape == (1844, 3)      # V=1088, F=1844
    assert f.min() == 0 and f.max() == len(v) - 1
    ex = np.asarray(syn.EXTENTS['car'])
    assert np.allclose(v.max(0) - v.min(0), ex, atol=1e-6)
    v2, f2 = syn.class_mesh('car', 10000)
    assert abs(len(v2) - 10466) < 400 and len(f2) > 19000      # subdivided cuboid of BASELINE config 5


def test_cuboid_mesh_matches_reference_construction():
    """Vertex / face order of tools/CreateCuboidMesh.py:33-108, restated as plain loops."""
    xr, yr, zr = (-0.2, 0.2), (-0.47, 0.47), (-0.15, 0.15)
    v, f = syn.cuboid_mesh(xr, yr, zr, 200)
    w, h, d = 0.4, 0.94, 0.3
    edge = (((w * h + h * d + w * d) * 2 / (200 * 2)) * 2) ** .5
    xs = xr[0] + np.linspace(0, w, int(w / edge + 1)); ys = yr[0] + np.linspace(0, h, int(h / edge + 1))
    zs = zr[0] + np.linspace(0, d, int(d / edge + 1))
    verts, faces, base = [], [], 0
    def side(outer, inner, mk):
        nonlocal base
        for o in outer:
            for i in inner:
                verts.append(mk(o, i))
        ni = len(inner)
        for m in range(len(outer) - 1):
            for n in range(ni - 1):
                faces.append((base + m * ni + n, base + m * ni + n + 1, base + (m + 1) * ni + n))
                faces.append((base + (m + 1) * ni + n + 1, base + m * ni + n + 1, base + (m + 1) * ni + n))
        base += len(outer) * ni
    side(ys, xs, lambda o, i: (i, o, zs[0])); side(ys, xs, lambda o, i: (i, o, zs[-1]))
    side(zs, xs, lambda o, i: (i, ys[0], o)); side(zs, xs, lambda o, i: (i, ys[-1], o))
    side(zs, ys, lambda o, i: (xs[0], i, o)); side(zs, ys, lambda o, i: (xs[-1], i, o))
    assert np.allclose(v, np.array(verts)) and np.array_equal(f, np.array(faces))


def test_pose_grid_order_and_geometry():
    C, th = syn.pose_grid(5.0)
    assert C.shape == (144, 3) and th.shape == (144,)
    # azimuth-major, then elevation, then theta (code/MeshPoseSolveAll.py:66)
    assert np.allclose(th[:3], np.linspace(-np.pi / 6, np.pi / 6, 3))
    assert np.allclose(np.linalg.norm(C, axis=1), 5.0, atol=1e-5)
    assert np.allclose(C[0], [0, 5 * np.sin(-np.pi / 6), 5 * np.cos(-np.pi / 6)], atol=1e-6)
    g = syn.Geometry.from_image_size((640, 800))
    assert (g.S, g.h, g.w, g.r0, g.c0) == (100, 80, 100, 10, 0)
    g = syn.Geometry.from_image_size((256, 672))
    assert (g.S, g.h, g.w, g.r0, g.c0) == (84, 32, 84, 26, 0)    # car: rows 26..57
    assert abs(syn.ndc_scale(12.0) - 9.51436) < 1e-4


def test_spherical_round_trip():
    import torch
    from nemo_amd.pose_solver import camera_position_from_spherical_angles, camera_position_to_spherical_angle
    rng = np.random.default_rng(0)
    d = rng.uniform(3, 8, 50); e = rng.uniform(-1.2, 1.2, 50); a = rng.uniform(0, 2 * np.pi, 50)
    C = camera_position_from_spherical_angles(d, e, a)
    assert np.allclose(C.numpy(), syn.spherical_to_campos(d, e, a), atol=1e-6)
    d2, e2, a2 = camera_position_to_spherical_angle(C)
    assert np.allclose(d2, d, atol=1e-5) and np.allclose(e2, e, atol=1e-5)
    assert np.allclose(np.mod(a2.numpy() - a + np.pi, 2 * np.pi) - np.pi, 0, atol=1e-5)


def test_loss_functions_match_formula():
    import torch
    from nemo_amd.pose_solver import loss_fun, loss_fun0, standard_loss_func_with_clutter
    o, c = torch.rand(8, 9), torch.rand(8, 9)
    assert torch.allclose(loss_fun(o, c), 1 - (torch.maximum(o, c).mean() - c.mean()))
    assert torch.allclose(loss_fun0(o), 1 - o.mean())
    ck = torch.rand(1, 5, 8, 9)
    assert torch.allclose(standard_loss_func_with_clutter(o[None], ck), loss_fun(o[None], ck.max(1)[0]))


def test_abi_library_exports_every_declared_symbol():
    from nemo_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'nemo_hip.h')).read()
    declared = sorted(set(re.findall(r'\b(nemo_[A-Za-z_0-9]+)\s*\(', hdr)))
    assert set(declared) == set(_lib.EXPORTS), (declared, _lib.EXPORTS)
    assert os.path.exists(_lib.LIB_PATH), 'build with nemo_amd/csrc/build.sh'
    lib = ctypes.CDLL(_lib.LIB_PATH)  # loads without a GPU; no compute call is made
    for name in declared:
        assert hasattr(lib, name), name
    assert ctypes.sizeof(_lib.NemoConfig) == 48


def test_product_path_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'nemo_amd')):
        for fn in files:
            if fn.endswith(('.py', '.hip', '.h')):
                txt = open(os.path.join(dirpath, fn)).read()
                assert 'from oracle' not in txt and 'import oracle' not in txt and 'libnemo_oracle' not in txt, fn


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from nemo_amd import _lib
    from nemo_amd.engine import PoseEngine
    with pytest.raises((_lib.NemoHipError, RuntimeError, AssertionError)):
        PoseEngine(40, (30, 40), 128)
    with pytest.raises(_lib.NemoHipError):
        PoseEngine(40, (30, 40), 128, device='cpu')


def test_pixel_range_estimates_are_within_three_probes():
    """The kernels settle 'first / last pixel centre inside [vmin, vmax]' with three probes of the
    PixToNdc table around a closed-form estimate instead of a search loop (nemo_amd/csrc/eval_pose.hpp, exact_range).
    That is exact iff the float32 estimates floor(f), ceil(f), f = ((v + 1) * S - 1) / 2, are within
    [0, +2] resp. [-2, 0] of the true indices: checked here over many raster sizes, including values
    one ulp either side of every pixel centre."""
    f32 = np.float32
    rng = np.random.default_rng(0)
    for S in [16, 30, 32, 48, 64, 75, 96, 100, 112, 125, 128, 160, 200, 224, 256, 333, 512, 1000]:
        Sf = f32(S)
        idx = np.arange(-4, S + 4)
        ndc = (f32(-1.0) + (f32(2.0) * idx.astype(f32) + f32(1.0)) / Sf).astype(f32)  # PixToNdc
        assert np.all(np.diff(ndc) > 0)
        c = ndc[2:-2]
        v = np.concatenate([rng.uniform(-1.0 - 1.5 / S, 1.0 + 1.5 / S, 200000).astype(f32), c,
                            np.nextafter(c, f32(9)), np.nextafter(c, f32(-9))]).astype(f32)
        f = ((np.clip(v, f32(-4), f32(4)) + f32(1.0)) * Sf - f32(1.0)) * f32(0.5)
        assert f.dtype == np.float32
        lo_true = np.searchsorted(ndc, v, side='left') - 4       # smallest i with ndc[i] >= v
        hi_true = np.searchsorted(ndc, v, side='right') - 1 - 4  # largest i with ndc[i] <= v
        dl, dh = lo_true - np.floor(f).astype(int), hi_true - np.ceil(f).astype(int)
        assert dl.min() >= 0 and dl.max() <= 2, (S, dl.min(), dl.max())
        assert dh.min() >= -2 and dh.max() <= 0, (S, dh.min(), dh.max())


def test_feature_record_round_trip_and_upload(tmp_path):
    """ExtractFeatures.py npz format (:137-139,166-168,223): writer -> lazy reader -> pixel-major batch."""
    import torch
    from nemo_amd import records
    rng = np.random.default_rng(3)
    D, h, w = 8, 5, 6
    names = {'mesh01': ['n02814533_1', 'n02814533_2'], 'mesh02': ['n03770679_9']}
    feats = {n: rng.normal(size=(D, h, w)).astype(np.float32) for st in names for n in names[st]}
    mem = {st: rng.normal(size=(7, D)).astype(np.float32) for st in names}
    clu = {st: rng.normal(size=(11, D)).astype(np.float32) for st in names}
    path = str(tmp_path / 'car' / 'rec.npz')
    records.write_feature_record(path, mem, clu, names, feats)
    with records.FeatureRecord(path) as rec:
        assert rec.subtypes == ['mesh01', 'mesh02']
        assert rec.names('mesh01') == names['mesh01'] and sorted(rec.image_keys) == sorted(feats)
        assert np.array_equal(rec.memory('mesh02'), mem['mesh02']) and np.array_equal(rec.clutter('mesh01'), clu['mesh01'])
        order = names['mesh02'] + names['mesh01']
        batch = rec.upload_pixel_major(order, 'cpu', chunk=2)
    assert batch.shape == (3, h, w, D) and batch.is_contiguous()
    for i, n in enumerate(order):
        assert torch.equal(batch[i], torch.from_numpy(feats[n]).permute(1, 2, 0))


def test_accuracy_table_matches_cal_accuracy_arithmetic(tmp_path):
    """code/CalAccuracy.py:15-44 on per-category error records written by save_category_errors."""
    from nemo_amd import records
    rng = np.random.default_rng(4)
    errs = {}
    for i, c in enumerate(records.CATEGORIES):
        e = np.abs(rng.normal(scale=0.3, size=20 + 3 * i))
        errs[c] = e
        records.save_category_errors(str(tmp_path / ('%s.npz' % c)), e, cad_index=np.ones(e.size, int))
    rows, text = records.accuracy_table_from_dir(str(tmp_path))
    n = np.array([errs[c].size for c in records.CATEGORIES], float)
    pi6 = np.array([np.mean(errs[c] < np.pi / 6) * 100 for c in records.CATEGORIES])
    med = np.array([np.degrees(np.median(errs[c])) for c in records.CATEGORIES])
    assert np.allclose(rows[0, :-1], pi6) and np.allclose(rows[2, :-1], med) and np.allclose(rows[3, :-1], n)
    assert np.isclose(rows[0, -1], (n * pi6).sum() / n.sum()) and np.isclose(rows[2, -1], (n * med).sum() / n.sum())
    lines = text.split('\n')
    assert lines[0].startswith('Metric | plane\tbike\t') and lines[0].endswith('tv\tMean')
    assert lines[1].startswith('Pi/6   | ') and lines[2].startswith('Pi/18  | ') and lines[3].startswith('MedErr | ')
    assert lines[1].split('\t')[-1] == '%.1f' % rows[0, -1]


def test_annotation_round_trip_and_rotation_error(tmp_path):
    from nemo_amd import records
    a = records.Annotation(theta=0.1, elevation=0.2, azimuth=1.3, distance=5.0, cad_index=2)
    p = str(tmp_path / 'anno' / 'img.npz')
    records.write_annotation(p, a)
    b = records.load_annotation(p)
    assert b == a
    assert records.rotation_error(a, 0.1, 0.2, 1.3, 5.0) < 1e-6
    assert abs(records.rotation_error(a, 0.1 + 0.05, 0.2, 1.3, 5.0) - 0.05) < 1e-6   # in-plane rotation only
    assert records.rotation_error(a, float('nan'), 0.2, 1.3, 5.0) == np.pi / 2          # :262-263


def test_fastdiv_chain_is_correctly_rounded():
    """csrc/fastdiv.h: r = fma(fma(-d, r0, 1), r0, r0); q0 = n r; q1 = fma(fma(-d, q0, n), r, q0);
    q = fma(fma(-d, q1, n), r, q1) is the correctly rounded n / d when r0 is the correctly rounded
    reciprocal, for operands in the range the kernel declares safe (exact rational arithmetic stands in
    for the fused multiply-adds; corner mantissas included).  An arbitrary one-ulp-off r0 is NOT enough
    (d = 2^k (1 - 2^-24) fails when r0 = 2^-k), so the guarantee on the device rests on v_rcp_f32 itself:
    tools/fastdiv_check.hip compares the chain on the MI355X with the compiler's IEEE division and with the
    quotient rounded from double precision, over 3.4e10 operand pairs."""
    from fractions import Fraction

    def rnd(x):  # Fraction -> nearest float32 (ties to even), normal range
        if x == 0:
            return np.float32(0.0)
        sgn, ax = (-1, -x) if x < 0 else (1, x)
        e = ax.numerator.bit_length() - ax.denominator.bit_length()
        if Fraction(2) ** e > ax:
            e -= 1
        scaled = ax / Fraction(2) ** (e - 23)           # in [2^23, 2^24)
        n, rem = divmod(scaled.numerator, scaled.denominator)
        twice = 2 * rem
        if twice > scaled.denominator or (twice == scaled.denominator and (n & 1)):
            n += 1
        return np.float32(sgn * float(n) * 2.0 ** (e - 23))

    def fma(a, b, c):
        return rnd(Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c)))

    rng = np.random.default_rng(11)
    corner = [0, 0x7fffff, 0x400000, 0x3fffff, 0x400001, 1]

    def make(k):
        man = rng.integers(0, 1 << 23, k)
        pick = rng.integers(0, 4, k) == 0
        man[pick] = rng.choice(corner, pick.sum())
        ex = rng.integers(127 - 40, 127 + 40, k)
        sg = rng.integers(0, 2, k)
        return ((sg << 31) | (ex << 23) | man).astype(np.uint32).view(np.float32)

    bad = 0
    for d in make(600):
        r0 = rnd(Fraction(1) / Fraction(float(d)))
        r = fma(fma(-d, r0, np.float32(1.0)), r0, r0)
        for n in make(8):
            q0 = rnd(Fraction(float(n)) * Fraction(float(r)))
            q1 = fma(fma(-d, q0, n), r, q0)
            q = fma(fma(-d, q1, n), r, q1)
            bad += int(q != rnd(Fraction(float(n)) / Fraction(float(d))))
    assert bad == 0


def test_accuracy_table_matches_reference_stdout(tmp_path):
    """PINNED BY THE REFERENCE: tests/golden/accuracy_table.txt is the stdout of the reference's own
    code/CalAccuracy.py, run by tests/golden/make_golden.py on records holding the arrays of
    tests/golden/accuracy_errors.npz; records.accuracy_table_from_dir prints the same text for the same records."""
    import os
    from nemo_amd import records
    gold = os.path.join(os.path.dirname(__file__), 'golden')
    with np.load(os.path.join(gold, 'accuracy_errors.npz')) as z:
        errs = {k: z[k] for k in z.files}
    assert sorted(errs) == sorted(records.CATEGORIES)
    for c, e in errs.items():
        records.save_category_errors(str(tmp_path / (c + '_occ1.npz')), e)
    rows, text = records.accuracy_table_from_dir(str(tmp_path), '_occ1')
    with open(os.path.join(gold, 'accuracy_table.txt')) as f:
        want = f.read()
    assert text + '\n' == want
    assert rows[3, -1] == sum(len(e) for e in errs.values())


def test_feature_record_direct_reads(tmp_path):
    """records.FeatureRecord: members of an uncompressed .npz (what np.savez / ExtractFeatures.py:223 writes) are read
    with pread straight from the file, a compressed record goes through the zip reader; both give the maps bit for bit,
    pixel-major, through the double-buffered upload (chunks smaller than the record, several reader threads), and a
    map of the wrong shape is refused."""
    import torch
    from nemo_amd import records
    rng = np.random.default_rng(0)
    feats = {'img%d' % i: rng.standard_normal((16, 5, 7)).astype(np.float32) for i in range(10)}
    meta = ({'mesh01': np.zeros((3, 16), np.float32)}, {'mesh01': np.zeros((2, 16), np.float32)}, {'mesh01': list(feats)})
    ref = np.stack([feats[n].transpose(1, 2, 0) for n in feats])
    for compressed in (False, True):
        p = str(tmp_path / ('r%d.npz' % compressed))
        records.write_feature_record(p, *meta, feats)
        if compressed:
            with np.load(p) as z:
                np.savez_compressed(p, **{k: z[k] for k in z.files})
        with records.FeatureRecord(p) as r:
            lay = r._layout('img0')
            assert (lay is None) == compressed and r.feature_shape('img3') == (16, 5, 7)
            for chunk, workers in ((3, 4), (64, 1)):
                out = r.upload_pixel_major(list(feats), 'cpu', chunk=chunk, workers=workers)
                assert out.shape == (10, 5, 7, 16) and np.array_equal(out.numpy(), ref)
            dst = np.empty((16, 5, 7), np.float32)
            r.read_feature_into('img7', dst)
            assert np.array_equal(dst, feats['img7'])
            with pytest.raises(ValueError):
                r.read_feature_into('img7', np.empty((16, 7, 5), np.float32))
            assert r.names('mesh01') == list(feats)
