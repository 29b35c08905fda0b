"""Config 2 with fp16 heatmaps (a mixed-precision 2-D head): the drop-in module fed the half maps
(widened inside the repack) against `.float()` first, which is what the reference's @force_fp32
does and what this module did before tempo_repack_nhwc_half.  GPU time per call, CUDA events."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import tempo_amd  # noqa: E402


def timed(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / n


def main():
    sys.argv = sys.argv[:1]
    args = bench.parse()
    dev = torch.device('cuda', 0)
    cfg, fms, cam, _ = bench.make_inputs(args, dev, 0)
    centers, scales, cameras = bench.make_inputs.meta
    layer = tempo_amd.ProjectLayer(cfg['image_size'], cfg['heatmap_size'], 'mean')
    layer.materialize_grids = False
    call = lambda maps: layer(maps, cfg['space_size'], [cfg['space_center']], cfg['cube_size'],
                              centers, scales, cameras)[0]
    for dt in (torch.float16, torch.bfloat16):
        half = [m.to(dt) for m in fms]
        with torch.no_grad():
            a = call(half)
            b = call([m.float() for m in half])
            assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))
            t_half = timed(lambda: call(half))
            t_conv = timed(lambda: call([m.float() for m in half]))
            t_f32 = timed(lambda: call(fms))
        print(f'{dt}: half maps in {t_half:.4f} ms | .float() first {t_conv:.4f} ms | '
              f'fp32 maps {t_f32:.4f} ms per 32-frame call')


if __name__ == '__main__':
    main()
