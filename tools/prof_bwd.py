"""Profile driver for the training step: one forward that records the clamp mask, then
--steps backward launches of one bench workload (root / window / window_h36m), nothing else
(for rocprofv3).  Takes bench.py's flags."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
if '--steps' not in sys.argv:
    sys.argv += ['--steps', '3']
a = bench.parse()
dev = torch.device('cuda:0')
wl = bench.make_workload(a, dev, seed=0)
ops = wl.ops
root = a.workload == 'root'
if wl.hm.shape[0] != wl.B:
    wl.hm = torch.empty(wl.B, *wl.hm.shape[1:], device=dev)
ops.repack_nhwc(wl.fms, out=wl.hm)
cp = wl.hm.shape[-1]
cmask = ops.new_clamp_mask(wl.B if root else wl.NW, wl.N, cp, dev)
gout = torch.rand(wl.out.shape, device=dev)
hs, im = wl.cfg['heatmap_size'], wl.cfg['image_size']
if root:
    ops.unproject_root(wl.hm, wl.cam, wl.axes, wl.J, hs, im, out=wl.out, clamp_mask=cmask)
else:
    tl, st, en = wl.geo
    ops.unproject_window(wl.hm, wl.cam, wl.fine, tl, st, en, wl.sub, wl.J, hs, im,
                         win_frame=wl.win_frame, out=wl.out, clamp_mask=cmask)
for _ in range(a.steps):
    if root:
        ops.unproject_root_backward(gout, wl.cam, wl.axes, cmask, wl.hm.shape, hs, im)
    else:
        ops.unproject_window_backward(gout, wl.cam, wl.fine, tl, st, en, cmask, wl.hm.shape,
                                      hs, im, win_frame=wl.win_frame)
torch.cuda.synchronize()
print('done', wl.name)
