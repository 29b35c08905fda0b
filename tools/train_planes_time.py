"""What the pose stage's training step costs today at config 3's shape, per call of the drop-in
module (32 windows = one proposal of each of 32 frames; tempo_pose.py:152-176 makes 10 such
calls per batch): differentiable volume forward + autograd's three torch.max + their backward
(a 0.5 GB gradient volume) + the HIP window backward -- what the reference's own code does and
what `get_planes` did under autograd before round 3 -- against the trainable fused planes
(planes kernel + claim pass, backward from the planes' gradients: no volume, no gradient
volume), and the inference-only fused planes.  GPU time, CUDA events."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import tempo_amd  # noqa: E402
from tempo_amd import synthetic as S  # noqa: E402


def timed(fn, n=10):
    for _ in range(3):
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
    kind = sys.argv[1] if len(sys.argv) > 1 else 'noise'
    sys.argv = sys.argv[:1]
    args = bench.parse()
    dev = torch.device('cuda', 0)
    cfg, fms, cam, _ = bench.make_inputs(args, dev, 0)
    centers, scales, cameras = bench.make_inputs.meta
    B = len(centers)
    if kind == 'blobs':
        # realistic heatmaps: a few Gaussian blobs per channel, exactly zero elsewhere -- most
        # lines of the volume tie at 0 and their gradient goes to (constant) face voxels
        w, h = cfg['heatmap_size']
        fms = [m.to(dev) for m in S.make_heatmaps(B, cfg['num_views'], cfg['num_joints'], h, w,
                                                  kind='blobs', seed=5)]
    print(f'heatmaps: {kind}')
    layer = tempo_amd.ProjectLayerWithMask(cfg['image_size'], cfg['heatmap_size'], 'mean')
    layer.set_space_size(cfg['space_center'], cfg['space_size'], cfg['sub_space_size'],
                         cfg['sub_cube_size'], dev)
    props = S.make_proposals(B, cfg['space_center'], cfg['space_size'], seed=1, invalid_every=0,
                             margin=0.15).to(dev)
    props[:, 5:7] = 1.0
    g = torch.Generator(device=dev).manual_seed(3)
    wts = [torch.rand(B, cfg['num_joints'], 64, 64, device=dev, generator=g) for _ in range(3)]
    leafs = [f.clone().requires_grad_(True) for f in fms]

    def infer():
        with torch.no_grad():
            layer.get_planes(fms, props, centers, scales, cameras)

    parts = {}

    def train(record=False):
        for f in leafs:
            f.grad = None
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        cubes, _ = layer.get_voxel(leafs, props, centers, scales, cameras)  # (training path)
        ev[1].record()
        planes = [torch.max(cubes, dim=d)[0] for d in (4, 3, 2)]
        loss = sum((p * w).sum() for p, w in zip(planes, wts))
        ev[2].record()
        loss.backward()
        ev[3].record()
        if record:
            ev[3].synchronize()
            parts['volume forward (records the clamp mask)'] = ev[0].elapsed_time(ev[1])
            parts['3 x torch.max + loss'] = ev[1].elapsed_time(ev[2])
            parts['backward (max scatter into a gradient volume + HIP window backward)'] = \
                ev[2].elapsed_time(ev[3])

    fparts = {}

    def fused(record=False):
        for f in leafs:
            f.grad = None
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        pxy, pxz, pyz, _, _ = layer.get_planes(leafs, props, centers, scales, cameras)
        loss = sum((p * w).sum() for p, w in zip((pxy, pxz, pyz), wts))
        ev[1].record()
        loss.backward()
        ev[2].record()
        if record:
            ev[2].synchronize()
            fparts['planes + claims (+ loss)'] = ev[0].elapsed_time(ev[1])
            fparts['backward from the planes\' gradients'] = ev[1].elapsed_time(ev[2])

    t_inf = timed(infer)
    t_train = timed(train)
    train(record=True)
    peak_volume = torch.cuda.max_memory_allocated() / 2**30
    torch.cuda.reset_peak_memory_stats()
    t_fused = timed(fused)
    fused(record=True)
    peak_fused = torch.cuda.max_memory_allocated() / 2**30
    print(f'config 3 shape, one module call = {B} windows of 64^3, {cfg["num_joints"]} joints:')
    print(f'  inference, fused planes (no volume)        {t_inf:8.3f} ms')
    print(f'  training step through the volume           {t_train:8.3f} ms')
    for k, v in parts.items():
        print(f'    {k:<70s} {v:8.3f} ms')
    # the whole pose-stage batch: P = 10 proposals per frame (tempo_pose.py's loop over n)
    P = 10
    cand = S.make_proposals(B * P, cfg['space_center'], cfg['space_size'], seed=2,
                            invalid_every=0, margin=0.15).to(dev)
    cand[:, 5:7] = 1.0
    cand = cand.view(B, P, 7)
    wb = [torch.rand(B, P, 1, cfg['num_joints'], 64, 64, device=dev, generator=g)
          for _ in range(3)]
    geo = (cfg['space_center'], cfg['space_size'], cfg['sub_space_size'], cfg['sub_cube_size'])

    def loops_volume():
        for f in leafs:
            f.grad = None
        loss = 0.0
        for n in range(P):
            cubes, _ = layer.get_voxel(leafs, cand[:, n], centers, scales, cameras)
            for k, d in enumerate((4, 3, 2)):
                loss = loss + (torch.max(cubes, dim=d)[0] * wb[k][:, n, 0]).sum()
        loss.backward()

    def batched_planes():
        for f in leafs:
            f.grad = None
        pl = layer.forward_sequence_planes([leafs], cand, *geo, [centers], [scales], [cameras])
        sum((p * w).sum() for p, w in zip(pl[:3], wb)).backward()

    torch.cuda.reset_peak_memory_stats()
    t_loops = timed(loops_volume, n=3)
    peak_loops = torch.cuda.max_memory_allocated() / 2**30
    torch.cuda.reset_peak_memory_stats()
    t_batched = timed(batched_planes, n=5)
    peak_batched = torch.cuda.max_memory_allocated() / 2**30
    print(f'  training step through the planes (no volume) {t_fused:6.3f} ms')
    for k, v in fparts.items():
        print(f'    {k:<70s} {v:8.3f} ms')
    print(f'  x 10 proposals (320 windows): inference {10 * t_inf:.1f} ms, training through the '
          f'volume {10 * t_train:.1f} ms, through the planes {10 * t_fused:.1f} ms')
    print(f'  peak memory: through the volume {peak_volume:.2f} GiB, through the planes '
          f'{peak_fused:.2f} GiB (inputs and weights included)')
    print(f'whole batch, {B * P} windows (10 proposals x {B} frames), forward + backward:')
    print(f'  the reference\'s loops through the volume (one backward over all)  {t_loops:8.2f} ms, '
          f'peak {peak_loops:.2f} GiB')
    print(f'  forward_sequence_planes under autograd (one call, no volume)      {t_batched:8.2f} ms, '
          f'peak {peak_batched:.2f} GiB')


if __name__ == '__main__':
    main()
