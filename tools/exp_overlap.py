"""Does repack(chunk c+1) overlap unproject(chunk c) on two streams?  (root workload)"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--workload', default='root')
ap.add_argument('--proposals', type=int, default=10)
a = ap.parse_args()
dev = torch.device('cuda:0')
wl = bench.RootWorkload(a, dev, seed=0)
ops = wl.ops
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()

def step_chunks(K):
    B = wl.B
    step = B // K
    evs = []
    main = torch.cuda.current_stream()
    sA.wait_stream(main); sB.wait_stream(main)
    for c in range(K):
        b0, b1 = c * step, (c + 1) * step
        with torch.cuda.stream(sA):
            ops.repack_nhwc([f[b0:b1] for f in wl.fms], out=wl.hm[b0:b1])
            e = torch.cuda.Event(); e.record(sA)
        with torch.cuda.stream(sB):
            sB.wait_event(e)
            ops.unproject_root(wl.hm[b0:b1], wl.cam[b0:b1], wl.axes, wl.J, wl.cfg['heatmap_size'],
                               wl.cfg['image_size'], out=wl.out[b0:b1])
    main.wait_stream(sA); main.wait_stream(sB)

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

wl.step(); ref = wl.out.clone()
print('sequential ms', round(timeit(wl.step), 4))
for K in (1, 2, 4, 8):
    wl.out.zero_(); step_chunks(K); torch.cuda.synchronize()
    print('chunks', K, 'ms', round(timeit(lambda: step_chunks(K)), 4), 'equal', torch.equal(ref, wl.out))
