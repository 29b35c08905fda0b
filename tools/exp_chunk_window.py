"""Window workloads in frame chunks (ops.window_step) vs one launch."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tempo_amd import ops

dev = torch.device('cuda:0')
for argv in (['--workload', 'window_h36m', '--batch', '128', '--timesteps', '1'],
             ['--workload', 'window']):
    sys.argv = ['x', '--no-cpu-baseline'] + argv
    a = bench.parse()
    wl = bench.make_workload(a, dev, seed=0)
    tl, st, en = wl.geo
    P = wl.NW // wl.B
    for chunk in (wl.B, 16, 8, 4, 2, 1):
        if chunk > wl.B:
            continue
        buf = torch.empty(chunk, *wl.hm.shape[1:], device=dev)
        def step():
            ops.window_step(wl.fms, wl.cam, wl.fine, tl, st, en, wl.sub, wl.cfg['heatmap_size'],
                            wl.cfg['image_size'], windows_per_frame=P, out=wl.out, chunk=chunk,
                            hm_buf=buf)
        for _ in range(3):
            step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            step()
        e1.record()
        torch.cuda.synchronize()
        print(f'{wl.name}: chunk {chunk:3d} frames: {e0.elapsed_time(e1) / 10:.4f} ms per step',
              flush=True)
        del buf
    del wl
