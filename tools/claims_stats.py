"""How sparse is the gradient the planes send back?  Voxels and 4x4x4 bricks of a 64^3 window
that hold at least one unblocked claim, for noise and blob heatmaps (config 3's shape, 32
windows) -- what the planes-gradient backward has to visit."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import tempo_amd  # noqa: E402
from tempo_amd import ops, synthetic as S  # noqa: E402


def main():
    sys.argv = sys.argv[:1]
    args = bench.parse()
    dev = torch.device('cuda', 0)
    cfg, fms, cam, _ = bench.make_inputs(args, dev, 0)
    centers, scales, cameras = bench.make_inputs.meta
    B = len(centers)
    w, h = cfg['heatmap_size']
    layer = tempo_amd.ProjectLayerWithMask(cfg['image_size'], cfg['heatmap_size'], 'mean')
    layer.set_space_size(cfg['space_center'], cfg['space_size'], cfg['sub_space_size'],
                         cfg['sub_cube_size'], dev)
    props = S.make_proposals(B, cfg['space_center'], cfg['space_size'], seed=1, invalid_every=0,
                             margin=0.15).to(dev)
    props[:, 5:7] = 1.0
    for kind in ('noise', 'blobs'):
        maps = fms if kind == 'noise' else [m.to(dev) for m in S.make_heatmaps(
            B, cfg['num_views'], cfg['num_joints'], h, w, kind='blobs', seed=5)]
        _, _, C, _, live, camt, (tl, st, en) = layer._window_inputs(maps, props, centers, scales,
                                                                    cameras)
        hm = ops.repack_nhwc(maps)
        out = ops.unproject_window_planes(hm, camt, layer._fine_axes, tl, st, en,
                                          layer._sub_cube, C, layer.heatmap_size,
                                          layer.image_size, live=live)
        claims = ops.window_planes_claims(hm, camt, layer._fine_axes, tl, st, en,
                                          layer._sub_cube, C, out[1:], layer.heatmap_size,
                                          layer.image_size, live=live)
        hot = torch.zeros(B, 64, 64, 64, dtype=torch.bool, device=dev)
        a = torch.arange(64, device=dev)
        nw = torch.arange(B, device=dev).view(B, 1, 1, 1).expand(B, C, 64, 64)
        p = a.view(1, 1, 64, 1).expand(B, C, 64, 64)
        q = a.view(1, 1, 1, 64).expand(B, C, 64, 64)
        blocked = 0
        for i, cl in enumerate(claims):
            ok = (cl & 1) == 0
            blocked += int((~ok).sum())
            idx = (cl >> 1).long().clamp(0, 63)
            sel = [nw[ok], None, None, None]
            if i == 0:
                sel[1:] = [p[ok], q[ok], idx[ok]]
            elif i == 1:
                sel[1:] = [p[ok], idx[ok], q[ok]]
            else:
                sel[1:] = [idx[ok], p[ok], q[ok]]
            hot[sel[0], sel[1], sel[2], sel[3]] = True
        bricks = hot.view(B, 16, 4, 16, 4, 16, 4).any(dim=6).any(dim=4).any(dim=2)
        print(f'{kind}: hot voxels {float(hot.float().mean()) * 100:.1f} % of a window, bricks with '
              f'a hot voxel {float(bricks.float().mean()) * 100:.1f} % '
              f'({float(bricks.float().sum(dim=(1, 2, 3)).mean()):.0f} of 4096), blocked claims '
              f'{blocked}')


if __name__ == '__main__':
    main()
