"""Window backward: clock ticks per phase, summed over all waves (library built with
-DTEMPO_BWD_PHASES, TEMPO_LIB pointing at it).  s_memtime ticks at a constant 100 MHz."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tempo_amd import _lib
sys.argv = ['x', '--workload', 'window', '--no-cpu-baseline'] + sys.argv[1:]
a = bench.parse()
dev = torch.device('cuda:0')
wl = bench.make_workload(a, dev, seed=0)
ops = wl.ops
if wl.hm.shape[0] != wl.B:
    wl.hm = torch.empty(wl.B, *wl.hm.shape[1:], device=dev)
ops.repack_nhwc(wl.fms, out=wl.hm)
cp = wl.hm.shape[-1]
cmask = ops.new_clamp_mask(wl.NW, wl.N, cp, dev)
gout = torch.rand(wl.out.shape, device=dev)
hs, im = wl.cfg['heatmap_size'], wl.cfg['image_size']
tl, st, en = wl.geo
ops.unproject_window(wl.hm, wl.cam, wl.fine, tl, st, en, wl.sub, wl.J, hs, im,
                     win_frame=wl.win_frame, out=wl.out, clamp_mask=cmask)
lib = ctypes.CDLL(os.environ['TEMPO_LIB'])
buf = (ctypes.c_ulonglong * 16)()
run = lambda: ops.unproject_window_backward(gout, wl.cam, wl.fine, tl, st, en, cmask,
                                            wl.hm.shape, hs, im, win_frame=wl.win_frame)
run(); lib.tempo_debug_bwd_phases(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
lib.tempo_debug_bwd_phases(buf, 1)
names = ['gradient staged', 'projection', 'box + records', 'accumulation', 'flush issued',
         'loop exit', 'atomics acknowledged']
waves = buf[7]
tot = sum(buf[k] for k in range(7))
print('launch %.3f ms, %d waves, %.1f ticks per wave' % (e0.elapsed_time(e1), waves, tot / waves))
for k, nm in enumerate(names):
    print('%-22s %10.1f ticks per wave  %5.1f %%' % (nm, buf[k] / waves, 100.0 * buf[k] / tot))
