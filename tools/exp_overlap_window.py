"""H36M windows: repack of chunk c+1 on a side stream under the window kernel of chunk c."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tempo_amd import ops

dev = torch.device('cuda:0')
sys.argv = ['x', '--no-cpu-baseline', '--workload', 'window_h36m', '--batch', '128', '--timesteps', '1']
a = bench.parse()
wl = bench.make_workload(a, dev, seed=0)
tl, st, en = wl.geo
B = wl.B
ref = torch.empty_like(wl.out)
ops.window_step(wl.fms, wl.cam, wl.fine, tl, st, en, wl.sub, wl.cfg['heatmap_size'],
                wl.cfg['image_size'], windows_per_frame=1, out=ref, chunk=B)
side = torch.cuda.Stream()
for chunk in (64, 32, 16, 8):
    bufs = [torch.empty(chunk, *wl.hm.shape[1:], device=dev) for _ in range(2)]
    def step():
        main = torch.cuda.current_stream()
        ready = [None, None]   # repack of buffer i done
        freed = [None, None]   # kernel reading buffer i done
        chunks = list(range(0, B, chunk))
        def repack(i, f0):
            with torch.cuda.stream(side):
                if freed[i & 1] is not None:
                    side.wait_event(freed[i & 1])
                ops.repack_nhwc([f[f0:f0 + chunk] for f in wl.fms], out=bufs[i & 1])
                ev = torch.cuda.Event(); ev.record(side); ready[i & 1] = ev
        side.wait_stream(main)
        repack(0, chunks[0])
        for i, f0 in enumerate(chunks):
            if i + 1 < len(chunks):
                repack(i + 1, chunks[i + 1])
            main.wait_event(ready[i & 1])
            sl = slice(f0, f0 + chunk)
            ops.unproject_window(bufs[i & 1], wl.cam[sl], wl.fine, tl[sl], st[sl], en[sl], wl.sub,
                                 wl.J, wl.cfg['heatmap_size'], wl.cfg['image_size'], out=wl.out[sl])
            ev = torch.cuda.Event(); ev.record(main); freed[i & 1] = ev
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f'chunk {chunk:3d} overlapped: {ms:.3f} ms per step, equal {torch.equal(wl.out, ref)}', flush=True)
    del bufs
