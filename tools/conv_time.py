"""process_layer (1x1 conv, 256 -> C) + root unprojection at BASELINE config 2's shape:
torch F.conv2d (NCHW) + tempo_root_step (repack + unproject)  vs
tempo_conv1x1_nhwc (channel-last out) + tempo_unproject_root on that buffer."""
import json, os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tempo_amd import ops

a = bench.parse(['--no-cpu-baseline'])
dev = torch.device('cuda:0')
wl = bench.RootWorkload(a, dev, seed=0)
cfg = wl.cfg
B, V, J = wl.B, wl.V, wl.J
w_, h_ = cfg['heatmap_size']
K = 256
g = torch.Generator(device=dev).manual_seed(1)
x = torch.rand(V * B, K, h_, w_, device=dev, generator=g)
wt = torch.randn(J, K, 1, 1, device=dev, generator=g) / 16
bias = torch.rand(J, device=dev, generator=g)
cp = ops.padded_channels(J)
hm = torch.empty(B, V, h_, w_, cp, device=dev)
out_a = torch.empty_like(wl.out)
out_b = torch.empty_like(wl.out)


def timed(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def ref_conv():
    return F.conv2d(x, wt, bias)


def ref_chain():
    m = F.conv2d(x, wt, bias)
    fms = list(torch.chunk(m, V))
    ops.root_step(fms, wl.cam, wl.axes, cfg['heatmap_size'], cfg['image_size'], out=out_a,
                  chunk=wl.chunk, hm_buf=wl.hm.view(-1))


def hip_conv():
    ops.conv1x1_nhwc(x, wt, bias, num_views=V, out=hm)


def hip_chain():
    ops.conv1x1_nhwc(x, wt, bias, num_views=V, out=hm)
    ops.unproject_root(hm, wl.cam, wl.axes, J, cfg['heatmap_size'], cfg['image_size'], out=out_b)


def hip_unproject_only():
    ops.unproject_root(hm, wl.cam, wl.axes, J, cfg['heatmap_size'], cfg['image_size'], out=out_b)


res = {'torch_conv2d_ms': timed(ref_conv), 'hip_conv1x1_nhwc_ms': timed(hip_conv),
       'torch_conv_plus_root_step_ms': timed(ref_chain), 'hip_conv_plus_unproject_ms': timed(hip_chain),
       'hip_unproject_alone_ms': timed(hip_unproject_only),
       'input_GB': x.numel() * 4 / 1e9}
res['hip_conv_GBps'] = res['input_GB'] / res['hip_conv1x1_nhwc_ms'] * 1e3
res['volume_max_abs_diff'] = float((out_a - out_b).abs().max())
print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in res.items()}))
