"""nhwc in place, the SAME 8-frame chunk unprojected repeatedly (its 79 MB of maps stay in the
Infinity Cache) against the 32-frame launch: is the in-place path slower because its
first-touch misses come from HBM, or because of its layout (V separate tensors)?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

dev = torch.device('cuda:0')
for layout in ('nhwc', 'nchw'):
    args = bench.parse(['--workload', 'root', '--input-layout', layout, '--chunk', '8', '--no-cpu-baseline'])
    wl = bench.RootWorkload(args, dev, 0)
    if layout == 'nchw':
        wl.repack(0, 8)
        print('repacked buffer, same chunk 8 frames, warm: %.4f ms' % timeit(lambda: wl.kernel(0, 8)))
        continue
    print('nhwc same chunk (frames 0-7) warm: %.4f ms per launch' % timeit(lambda: wl.kernel(0, 8)))
    def sweep():
        for c0 in range(0, 32, 8): wl.kernel(c0, 8)
    print('nhwc 4 chunks of 8 (each cold): %.4f ms per launch' % (timeit(sweep) / 4))
    wl.chunk = 32
    print('nhwc one launch of 32: %.4f ms per 8 frames' % (timeit(lambda: wl.kernel(0, 32)) / 4))
    # the same values in ONE tensor [F][V][H][W][CP] (what tempo_conv1x1_nhwc writes)
    hm = torch.stack([f.permute(0, 2, 3, 1) for f in wl.fms], dim=1).contiguous()
    cfg = wl.cfg
    f8 = lambda: wl.ops.unproject_root(hm[:8], wl.cam[:8], wl.axes, wl.J, cfg['heatmap_size'], cfg['image_size'], out=wl.out[:8])
    print('one [F,V,H,W,CP] tensor, frames 0-7 warm: %.4f ms' % timeit(f8))
    f32 = lambda: wl.ops.unproject_root(hm, wl.cam, wl.axes, wl.J, cfg['heatmap_size'], cfg['image_size'], out=wl.out)
    print('one [F,V,H,W,CP] tensor, 32 frames: %.4f ms per 8 frames' % (timeit(f32) / 4))
