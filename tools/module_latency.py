"""Host-side cost of the drop-in modules at the reference's test-time shapes (B = 1)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import cases, tempo_amd
from tempo_amd import synthetic as S
from tempo_amd.camera import pack_cameras

dev = torch.device('cuda:0')
for B in (1, 8):
    spec = dict(cfg='panoptic', B=B, cube=[80, 80, 20], hm='noise', seed=1)
    case = cases.to_device(cases.build_root(None, full=spec), dev)
    layer = tempo_amd.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    layer.materialize_grids = False
    def call():
        with torch.no_grad():
            return layer(*case['args'])
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        call()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    t0 = time.perf_counter()
    for _ in range(n):
        pack_cameras(*case['args'][4:], case['image_size'])
    pk = (time.perf_counter() - t0) / n * 1e3
    print(f'root B={B}: module forward {ms:.3f} ms per call, of which pack_cameras {pk:.3f} ms', flush=True)
spec = dict(cfg='panoptic', B=10, sub_cube=[64, 64, 64], hm='noise', seed=2, invalid_every=0)
case = cases.to_device(cases.build_window(None, full=spec), dev)
layer = tempo_amd.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
def wcall():
    with torch.no_grad():
        return layer(*case['args'])
for _ in range(5):
    wcall()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    wcall()
torch.cuda.synchronize()
print(f'window B=10: module forward {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per call', flush=True)
