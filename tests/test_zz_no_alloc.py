"""nemo_solve_static (the entry point that allocates nothing: launch-order workspace in a file-scope device array)
against nemo_solve, through the C ABI, bit for bit -- two calls in a row, the second with a smaller input, so a
stale tail of the first call's workspace contents would show."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


def test_solve_static_matches_solve_twice():
    from nemo_amd.pose_solver import SolverConfig
    from nemo_amd.workloads import twelve_class_workload
    wl = twelve_class_workload(200, seed=17, cfg=SolverConfig(total_epochs=25))
    s, eng = wl.solver, wl.solver.engine
    clu = eng.clutter_score(wl.feats, wl.clutter)
    C0, th0, _, _ = eng.grid_search(wl.feats, clu, s.grid_C, s.grid_theta, mesh_idx=wl.mesh_idx, grid_idx=wl.mesh_idx)
    eng.reserve(200)
    for n in (200, 90):  # both above 64 jobs: the launch-order workspace is in use; the second call is smaller
        idx = torch.arange(n, dtype=torch.int32, device='cuda')
        mesh = wl.mesh_idx[:n].contiguous()
        c0, t0 = C0[:n].contiguous(), th0[:n].contiguous()
        pose_ref, loss_ref, trace_ref = eng.solve(wl.feats, clu, c0, t0, img_idx=idx, mesh_idx=mesh, want_trace=True)
        out = (torch.full((n, 4), float('nan'), device='cuda'), torch.full((n,), float('nan'), device='cuda'))
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        pose, loss = eng.solve(wl.feats, clu, c0, t0, img_idx=idx, mesh_idx=mesh, out=out, static=True)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free0            # nothing was allocated, by anyone
        assert torch.equal(pose, pose_ref) and torch.equal(loss, loss_ref)
        assert torch.isfinite(pose).all()
        _, _, trace = eng.solve(wl.feats, clu, c0, t0, img_idx=idx, mesh_idx=mesh, want_trace=True, static=True)
        assert torch.equal(trace, trace_ref)


def test_solve_static_limits():
    from nemo_amd import _lib
    from nemo_amd.pose_solver import SolverConfig
    from nemo_amd.workloads import twelve_class_workload
    wl = twelve_class_workload(4, seed=2, cfg=SolverConfig(total_epochs=5), categories=['car'])
    eng = wl.solver.engine
    clu = eng.clutter_score(wl.feats, wl.clutter)
    n = 16385                                                    # one more job than the static workspace holds
    idx = (torch.arange(n, device='cuda') % 4).to(torch.int32)
    C0 = torch.tensor([[0.0, 0.0, 5.0]], device='cuda').repeat(n, 1).contiguous()
    with pytest.raises(_lib.NemoHipError, match='static workspace holds 16384'):
        eng.solve(wl.feats, clu, C0, torch.zeros(n, device='cuda'), img_idx=idx, static=True)
    pose, _ = eng.solve(wl.feats, clu, C0[:4].contiguous(), torch.zeros(4, device='cuda'), static=True)  # still usable
    assert torch.isfinite(pose).all()
