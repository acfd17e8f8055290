import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nemo_amd import synthetic as syn
from nemo_amd.engine import PoseEngine
g = np.load('tests/golden/path_small.npz')
geo = syn.Geometry.from_image_size(tuple(g['image_hw']))
eng = PoseEngine(geo.S, (geo.h, geo.w), 128); eng.register_mesh(g['verts'], g['faces'], g['bank'])
dev = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).cuda().contiguous()
feat, clu = dev(g['feat'][None]), dev(g['clu'][None])
gC, gth = syn.pose_grid(float(g['distance']))
C0, th0, best, _ = eng.grid_search(feat, clu, dev(gC[None]), dev(gth[None]))
eng.set_schedule(300)
pose, loss, tr = eng.solve(feat, clu, C0, th0, want_trace=True)
tr = tr.cpu().numpy()[0]; ot = g['trace']
d = np.abs(tr[:, :4] - ot[:, :4]).max(1)
for t in [0, 1, 2, 5, 10, 20, 50, 99, 100, 150, 199, 200, 250, 299]:
    print(t, 'dpose %.2e' % d[t], 'loss gpu %.6f orc %.6f' % (tr[t, 4], ot[t, 4]), 'g gpu', tr[t, 5:9], 'orc', ot[t, 5:9])
print('final gpu', pose.cpu().numpy(), 'orc', g['final_C'], g['final_theta'])
