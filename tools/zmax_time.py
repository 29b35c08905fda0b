"""Cost of the fused root z-max: chunked step with and without it, vs torch.max afterwards."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tempo_amd import ops
sys.argv = ['x', '--no-cpu-baseline']
a = bench.parse()
dev = torch.device('cuda:0')
wl = bench.RootWorkload(a, dev, seed=0)
args = (wl.fms, wl.cam, wl.axes, wl.cfg['heatmap_size'], wl.cfg['image_size'])
z = torch.empty(wl.B, wl.J, 80, 80, device=dev)
def t(fn):
    for _ in range(20): fn()
    best = 1e9
    for _ in range(5):  # (best of 5 blocks: the GPU's clock dips now and then)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(100): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 100)
    return best
print('step (python chunk loop)        %.4f ms' % t(lambda: ops.root_step(*args, out=wl.out, in_library=False)))
print('step (library chunk loop)       %.4f ms' % t(lambda: ops.root_step(*args, out=wl.out)))
print('step + fused z-max (python)     %.4f ms' % t(lambda: ops.root_step(*args, out=wl.out, zmax_out=z, in_library=False)))
z.fill_(-7.)
print('step + fused z-max (library)    %.4f ms' % t(lambda: ops.root_step(*args, out=wl.out, zmax_out=z)))
print('step + torch.max(cubes, dim=4)  %.4f ms' % t(lambda: (ops.root_step(*args, out=wl.out, in_library=False), wl.out.max(dim=4))))
print('equal', torch.equal(torch.nan_to_num(z, nan=-1.), torch.nan_to_num(wl.out.max(dim=4)[0], nan=-1.)))
