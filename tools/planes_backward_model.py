"""Pricing a TRAINABLE fused-planes path (VERDICT r2 item 7 ii) before building it.

The pose regressor trains through ``torch.max(cubes, dim)`` over each axis of the 64^3 window
volume (tempo_pose.py:172-176).  Autograd sends a plane's gradient to ONE voxel per plane entry
and channel -- the argmax, first index on ties (CPU torch.max) -- so of a window's 262144 x C
volume gradients at most 3 x 4096 x C are non-zero.  A backward that starts from the three
plane gradients (never materialising the 5 GB gradient volume) has to
  * know the argmax voxels (the forward would have to emit them: 64-bit (value, index) maxima),
  * project those voxels into every view and scatter 4 texels x (their non-zero channels).
This script counts that work on real projections: the CPU oracle's window volume for Panoptic-
like ring cameras, blob heatmaps (realistic: mostly zero, so whole lines tie at 0 and their
argmax is index 0, a face of the window) and noise heatmaps (worst case: argmaxes anywhere).

Per window it reports: voxels with a non-zero gradient in any channel, 4x4x4 bricks holding
one (the unit of the shipped window backward, whose cost is per brick), 64-byte texel lines
touched per view with per-voxel scatter and after per-brick aggregation, and the time these
imply at config 3 (320 windows, 5 views) from the measured rates: 16 G texel-line atomics/s
chip-wide for scattered lines, 18.7 ms per 320 x 4096 bricks for the brick
kernel.

    python tools/planes_backward_model.py          # a minute of CPU
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import cases  # noqa: E402
from oracle import project_layer_ref as O  # noqa: E402
from tempo_amd import synthetic as S  # noqa: E402

ATOMIC_LINES_PER_S = 16e9      # scattered 64-byte float-atomic requests, chip-wide (measured)
BRICK_KERNEL_MS = 18.7         # shipped window backward, config 3: 320 windows x 4096 bricks
WINDOWS, VIEWS = 320, 5


def window_volume(kind, seed):
    spec = dict(cfg='panoptic', B=1, sub_cube=[64, 64, 64], hm=kind, seed=seed, invalid_every=0)
    case = cases.build_window(None, full=spec)
    args = list(case['args'])
    args[1][:, 5:7] = 1.0  # the whole window is sampled (bench setting)
    layer = O.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    cubes, _ = layer(*args)
    return cubes[0], case, args


def project_texels(case, args, vox_idx, layer_geom):
    """top-left texel (x0, y0) of every listed voxel in every view (oracle coordinate chain)"""
    fms, props, center, space, sub, sub_cube, centers, scales, cameras = args
    ref = O.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    ref.set_space_size(center, space, sub, sub_cube, 'cpu')
    tl = torch.round(props[:, :3] * ref.scale + ref.bias).int()[0]
    fine = [torch.linspace(-space[i] / 2, space[i] / 2, int(ref.fine_voxels_per_axis[i])) +
            center[i] for i in range(3)]
    X = torch.stack([fine[i][(tl[i] + vox_idx[:, i]).clamp(0, len(fine[i]) - 1)]
                     for i in range(3)], dim=1)
    w, h = case['heatmap_size']
    out = []
    for v, cam in enumerate(cameras[0]):
        xy = O.SimpleCameraTorch(cam, 'cpu').world_to_pixel(X)
        width, height = centers[0][v] * 2
        trans = torch.as_tensor(O.get_affine_transform(centers[0][v], scales[0][v] / 200.0, 0,
                                                       case['image_size']), dtype=torch.float)
        g = torch.clamp(O._sample_coords(xy, trans, w, h, case['image_size'], width, height,
                                         'cpu'), -1.1, 1.1)
        ix = ((g[:, 0] + 1) * w - 1) / 2
        iy = ((g[:, 1] + 1) * h - 1) / 2
        out.append((torch.floor(ix).long(), torch.floor(iy).long()))
    return out, (w, h)


def count(kind, seed):
    cubes, case, args = window_volume(kind, seed)
    C = cubes.shape[0]
    hot = torch.zeros(64, 64, 64, dtype=torch.bool)
    pairs = 0
    for dim in (3, 2, 1):  # max over z, y, x of [C, X, Y, Z]
        idx = cubes.max(dim=dim)[1]  # argmax, first index on ties
        a, b_ = torch.meshgrid(torch.arange(64), torch.arange(64), indexing='ij')
        for ch in range(C):
            if dim == 3:
                hot[a, b_, idx[ch]] = True
            elif dim == 2:
                hot[a, idx[ch], b_] = True
            else:
                hot[idx[ch], a, b_] = True
            pairs += 4096
    vox = hot.nonzero()
    bricks = torch.unique((vox // 4) @ torch.tensor([256, 16, 1]))
    tex, (w, h) = project_texels(case, args, vox, None)
    lines_scatter, lines_brick = 0, 0
    bid = (vox // 4) @ torch.tensor([256, 16, 1])
    for x0, y0 in tex:
        keys = []
        for i in (0, 1):
            for j in (0, 1):
                ok = (x0 + i >= 0) & (x0 + i < w) & (y0 + j >= 0) & (y0 + j < h)
                lines_scatter += int(ok.sum())
                keys.append((bid * (1 << 24) + (y0 + j) * 4096 + (x0 + i))[ok])
        lines_brick += len(torch.unique(torch.cat(keys)))
    return dict(voxels=len(vox), pairs=pairs, bricks=len(bricks),
                lines_scatter=lines_scatter / VIEWS, lines_brick=lines_brick / VIEWS,
                zero_frac=float((cubes == 0).float().mean()))


def main():
    print('%-7s %9s %8s %8s %14s %14s | %s' % ('maps', 'hot vox', 'of 262k', 'bricks',
                                               'lines/view', 'after bricks',
                                               'config 3: scatter ms, brick-kernel ms'))
    for kind in ('blobs', 'noise'):
        rows = [count(kind, 900 + s) for s in range(2)]
        m = {k: float(np.mean([r[k] for r in rows])) for k in rows[0]}
        scatter_ms = WINDOWS * VIEWS * m['lines_scatter'] / ATOMIC_LINES_PER_S * 1e3
        brick_ms = BRICK_KERNEL_MS * m['bricks'] / 4096
        print('%-7s %9.0f %7.1f%% %8.0f %14.0f %14.0f | %6.1f ms %6.1f ms   (volume %.0f%% zero)'
              % (kind, m['voxels'], 100 * m['voxels'] / 262144, m['bricks'], m['lines_scatter'],
                 m['lines_brick'], scatter_ms, brick_ms, 100 * m['zero_frac']))


if __name__ == '__main__':
    main()
