import sys, time, torch
sys.path.insert(0, '/root/repo')
from nemo_amd.engine import PoseEngine
eng = PoseEngine(200, (160, 200), 256)
feats = torch.randn(128, 160, 200, 256, device='cuda')
for K in (1, 24, 64, 512):
    cb = torch.nn.functional.normalize(torch.randn(K, 256, device='cuda'), dim=1).contiguous()
    eng.clutter_score(feats, cb); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): out = eng.clutter_score(feats, cb)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f'D=256 K={K}: {dt*1e3:.3f} ms, {feats.numel()*4/dt/1e12:.2f} TB/s, {2.0*feats.numel()*K/dt/1e12:.1f} TFLOP/s', flush=True)
