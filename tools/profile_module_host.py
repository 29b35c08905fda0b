"""Where the drop-in modules' host time goes: cProfile over calls of
tempo_amd.ProjectLayer.forward at config 2 (B = 32, lists of dicts, as tempo_detector.py:187
calls it) and of ProjectLayerWithMask.forward at the reference's per-candidate call shape
(B = 1 window, tempo_pose.py:152-162).  Needs a GPU (the calls launch the kernels).

    python tools/profile_module_host.py [calls]
"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import tempo_amd  # noqa: E402
from tempo_amd import synthetic as S  # noqa: E402


def report(name, call, n):
    with torch.no_grad():
        for _ in range(20):
            call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            call()
        host = (time.perf_counter() - t0) / n * 1e3
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(n):
            call()
        pr.disable()
        torch.cuda.synchronize()
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats('tottime').print_stats(22)
    print(f'==== {name}: {host:.4f} ms of host time per call (unprofiled), {n} calls')
    lines = out.getvalue().splitlines()
    print('\n'.join(l for l in lines if l.strip())[:6000])


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    sys.argv = sys.argv[:1]
    args = bench.parse()
    dev = torch.device('cuda', 0)
    cfg, fms, cam, cam_host = bench.make_inputs(args, dev, 0)
    centers, scales, cameras = bench.make_inputs.meta
    layer = tempo_amd.ProjectLayer(cfg['image_size'], cfg['heatmap_size'], 'mean')
    layer.materialize_grids = False
    report('ProjectLayer.forward, config 2 (B = 32)',
           lambda: layer(fms, cfg['space_size'], [cfg['space_center']], cfg['cube_size'],
                         centers, scales, cameras), n)

    wlayer = tempo_amd.ProjectLayerWithMask(cfg['image_size'], cfg['heatmap_size'], 'mean')
    P = S.PANOPTIC
    props = S.make_proposals(10, P['space_center'], P['space_size'], seed=3, invalid_every=0,
                             margin=0.1).to(dev)
    one = [f[:1] for f in fms]
    c1, s1, k1 = centers[:1], scales[:1], cameras[:1]
    report('ProjectLayerWithMask.forward, one candidate (B = 1 window)',
           lambda: wlayer(one, props[:1], P['space_center'], P['space_size'],
                          P['sub_space_size'], P['sub_cube_size'], c1, s1, k1), n)


if __name__ == '__main__':
    main()
