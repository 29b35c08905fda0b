"""Root step, NCHW input: does repack(chunk c+1) overlap unproject(chunk c) on two streams with
two alternating channel-last buffers?  (Re-measured in round 2 with streaming stores.)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device('cuda:0')
a = bench.parse(['--no-cpu-baseline'])
wl = bench.RootWorkload(a, dev, seed=0)
ops, cfg = wl.ops, wl.cfg
sR, sU = torch.cuda.Stream(), torch.cuda.Stream()
h, w = cfg['heatmap_size'][1], cfg['heatmap_size'][0]
cp = ops.padded_channels(wl.J)


def piped(chunk):
    bufs = [torch.empty(chunk, wl.V, h, w, cp, device=dev) for _ in range(2)]
    done = [None, None]  # event: the unprojection that last read the buffer

    def step():
        main = torch.cuda.current_stream()
        sR.wait_stream(main); sU.wait_stream(main)
        for i, c0 in enumerate(range(0, wl.B, chunk)):
            n = min(chunk, wl.B - c0)
            buf = bufs[i & 1]
            with torch.cuda.stream(sR):
                if done[i & 1] is not None:
                    sR.wait_event(done[i & 1])
                ops.repack_nhwc([f[c0:c0 + n] for f in wl.fms], out=buf[:n])
                e = torch.cuda.Event(); e.record(sR)
            with torch.cuda.stream(sU):
                sU.wait_event(e)
                ops.unproject_root(buf[:n], wl.cam[c0:c0 + n], wl.axes, wl.J, cfg['heatmap_size'],
                                   cfg['image_size'], out=wl.out[c0:c0 + n])
                d = torch.cuda.Event(); d.record(sU)
                done[i & 1] = d
        main.wait_stream(sR); main.wait_stream(sU)
    return step


def timeit(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


wl.step(); ref = wl.out.clone()
print('one stream, library chunk loop (chunk %d): %.4f ms' % (wl.chunk, timeit(wl.step)))
for chunk in (4, 8, 16):
    st = piped(chunk)
    wl.out.zero_(); st(); torch.cuda.synchronize()
    print('two streams, chunk %2d: %.4f ms  equal %s' % (chunk, timeit(st), torch.equal(ref, wl.out)))
