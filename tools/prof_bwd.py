"""Profile driver for the training step: one forward that records the clamp mask, then
--steps backward launches of one bench workload (root / window / window_h36m), nothing else
(for rocprofv3).  Takes bench.py's flags; with --planes only: the training step through the
fused planes (planes kernel, claim pass, backward from the planes' gradients)."""
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
hs, im = wl.cfg['heatmap_size'], wl.cfg['image_size']
if a.planes == 'only' and not root:
    # the training step through the fused planes: planes + claim pass + backward per step
    tl, st, en = wl.geo
    gpl = [torch.rand(p.shape, device=dev) for p in wl.planes]
    for _ in range(a.steps):
        ops.unproject_window_planes(wl.hm, wl.cam, wl.fine, tl, st, en, wl.sub, wl.J, hs, im,
                                    win_frame=wl.win_frame, planes_out=wl.planes)
        claims = ops.window_planes_claims(wl.hm, wl.cam, wl.fine, tl, st, en, wl.sub, wl.J,
                                          wl.planes, hs, im, win_frame=wl.win_frame)
        ops.unproject_window_planes_backward(
            gpl, claims, wl.cam, wl.fine, tl, st, en, wl.hm.shape, hs, im,
            win_frame=wl.win_frame,
            tile_texels=ops.backward_tile_texels(wl.cam_host, wl.fine_host, hs, im, planes=True))
    torch.cuda.synchronize()
    print('done', wl.name, 'training through the planes')
    sys.exit(0)
cp = wl.hm.shape[-1]
cmask = ops.new_clamp_mask(wl.B if root else wl.NW, wl.N, cp, dev)
gout = torch.rand(wl.out.shape, device=dev)
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
