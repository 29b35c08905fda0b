"""Root step in frame chunks that reuse one small channel-last buffer (does keeping the
repacked maps in the Infinity Cache / L2 help the gather?)."""
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
B = wl.B
for chunk in (32, 16, 12, 10, 8, 7, 6, 5):
    hm = torch.empty(chunk, wl.V, *wl.hm.shape[2:], device=dev)
    def step():
        for c0 in range(0, B, chunk):
            n = min(chunk, B - c0)
            fms = [f[c0:c0 + n] for f in wl.fms]
            ops.repack_nhwc(fms, out=hm[:n])
            ops.unproject_root(hm[:n], wl.cam[c0:c0 + n], wl.axes, wl.J, wl.cfg['heatmap_size'],
                               wl.cfg['image_size'], out=wl.out[c0:c0 + n])
    for _ in range(5):
        step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(30):
        step()
    e1.record()
    torch.cuda.synchronize()
    print(f'chunk {chunk:2d} frames: {e0.elapsed_time(e1) / 30:.4f} ms per 32-frame step', flush=True)

# two streams, double-buffered: repack of chunk c+1 under the unprojection of chunk c
for chunk in (8, 4):
    hms = [torch.empty(chunk, wl.V, *wl.hm.shape[2:], device=dev) for _ in range(2)]
    s_r, s_k = torch.cuda.Stream(), torch.cuda.Stream()
    def step2():
        done_k = [None, None]
        for i, c0 in enumerate(range(0, B, chunk)):
            buf = hms[i & 1]
            with torch.cuda.stream(s_r):
                if done_k[i & 1] is not None:
                    s_r.wait_event(done_k[i & 1])
                ops.repack_nhwc([f[c0:c0 + chunk] for f in wl.fms], out=buf)
                ev = torch.cuda.Event(); ev.record(s_r)
            with torch.cuda.stream(s_k):
                s_k.wait_event(ev)
                ops.unproject_root(buf, wl.cam[c0:c0 + chunk], wl.axes, wl.J,
                                   wl.cfg['heatmap_size'], wl.cfg['image_size'],
                                   out=wl.out[c0:c0 + chunk])
                done_k[i & 1] = torch.cuda.Event(); done_k[i & 1].record(s_k)
    for _ in range(5):
        step2()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(30):
        step2()
    torch.cuda.synchronize()
    print(f'2 streams, chunk {chunk}: {(time.perf_counter() - t0) / 30 * 1e3:.4f} ms per step', flush=True)
