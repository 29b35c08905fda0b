"""Root step as a HIP graph: single-stream chunk loop, and a two-branch version in which the
repack of chunk c+1 may run beside the unprojection of chunk c (double-buffered)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tempo_amd import ops

sys.argv = ['x', '--no-cpu-baseline']
a = bench.parse()
dev = torch.device('cuda:0')
wl = bench.RootWorkload(a, dev, seed=0)
B, chunk = wl.B, wl.chunk
ref = torch.empty_like(wl.out)
ops.root_step(wl.fms, wl.cam, wl.axes, wl.cfg['heatmap_size'], wl.cfg['image_size'], out=ref)
torch.cuda.synchronize()

def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

print(f'eager library loop      {timeit(wl.step):.4f} ms', flush=True)

# (a) single-stream graph
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    wl.step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    wl.step()
wl.out.zero_()
g.replay(); torch.cuda.synchronize()
print(f'graph, one stream       {timeit(g.replay):.4f} ms  equal {torch.equal(wl.out, ref)}', flush=True)

# (b) two branches, double-buffered
bufs = [torch.empty(chunk, *wl.hm.shape[1:], device=dev) for _ in range(2)]
side = torch.cuda.Stream()
def two_branch():
    main = torch.cuda.current_stream()
    ready, freed = [None, None], [None, None]
    chunks = list(range(0, B, chunk))
    def repack(i, c0):
        with torch.cuda.stream(side):
            if freed[i & 1] is not None:
                side.wait_event(freed[i & 1])
            ops.repack_nhwc([f[c0:c0 + chunk] for f in wl.fms], out=bufs[i & 1])
            ev = torch.cuda.Event(); ev.record(side); ready[i & 1] = ev
    side.wait_stream(main)
    repack(0, chunks[0])
    for i, c0 in enumerate(chunks):
        if i + 1 < len(chunks):
            repack(i + 1, chunks[i + 1])
        main.wait_event(ready[i & 1])
        sl = slice(c0, c0 + chunk)
        ops.unproject_root(bufs[i & 1], wl.cam[sl], wl.axes, wl.J, wl.cfg['heatmap_size'],
                           wl.cfg['image_size'], out=wl.out[sl])
        ev = torch.cuda.Event(); ev.record(main); freed[i & 1] = ev
    main.wait_stream(side)
with torch.cuda.stream(s):
    two_branch()
torch.cuda.synchronize()
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2, stream=s):
    two_branch()
wl.out.zero_()
g2.replay(); torch.cuda.synchronize()
print(f'graph, two branches     {timeit(g2.replay):.4f} ms  equal {torch.equal(wl.out, ref)}', flush=True)
