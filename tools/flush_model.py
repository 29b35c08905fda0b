"""Window backward: how many texels does a region of voxels touch per view?  (= 64-byte atomic
requests of its flush.)  Pinhole model of the bench's ring cameras (no distortion), 64^3 windows
of 2000 mm at a few places of the Panoptic space; regions of different shapes, counted per 64
voxels so that shapes of different volume compare."""
import itertools, math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tempo_amd import synthetic as S

cfg = S.PANOPTIC
cams = S.ring_cameras(5, seed=0, target=cfg['space_center'])
scale = 0.125  # raw 1920 x 1080 -> 240 x 135 (the crop to 128 rows only shifts)
pitch = cfg['sub_space_size'][0] / (cfg['sub_cube_size'][0] - 1)


def project(cam, X):
    R = np.array(cam['R']); pos = np.array(cam['T']).reshape(3)
    K = np.array(cam['K'])
    xc = (X - pos) @ R
    u = K[0, 0] * xc[:, 0] / xc[:, 2] + K[0, 2]
    v = K[1, 1] * xc[:, 1] / xc[:, 2] + K[1, 2]
    return u * scale - 0.5, v * scale - 0.5


def touched(shape, centre):
    sx, sy, sz = shape
    n = 64
    idx = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing='ij'), -1).reshape(-1, 3)
    X = (idx - (n - 1) / 2) * pitch + np.array(centre)
    region = (idx[:, 0] // sx) * 10000 + (idx[:, 1] // sy) * 100 + idx[:, 2] // sz
    nreg = len(np.unique(region))
    tot = 0
    box = 0
    for cam in cams:
        hx, hy = project(cam, X)
        x0 = np.floor(hx).astype(np.int64); y0 = np.floor(hy).astype(np.int64)
        keys = []
        W, H = cfg['heatmap_size']
        for i, j in itertools.product((0, 1), (0, 1)):
            ok = (x0 + i >= 0) & (x0 + i < W) & (y0 + j >= 0) & (y0 + j < H)  # only texels of the map
            keys.append((region * (1 << 24) + (y0 + j) * 4096 + (x0 + i))[ok])
        tot += len(np.unique(np.concatenate(keys)))
        sample = np.unique(region)[:: max(1, nreg // 64)]
        for r0 in sample:
            m = region == r0
            bx = np.clip(x0[m], -1, W); by = np.clip(y0[m], -1, H)
            box += (bx.max() - bx.min() + 2) * (by.max() - by.min() + 2) * (nreg / len(sample))
    vox = sx * sy * sz
    return tot / (nreg * len(cams)) * 64 / vox, box / (nreg * len(cams))


shapes = [(4, 4, 4), (8, 8, 1), (16, 4, 1), (8, 4, 2), (2, 2, 16), (1, 1, 64), (16, 16, 1), (8, 8, 4),
          (32, 8, 1), (16, 16, 2), (8, 8, 8), (32, 32, 1), (16, 16, 4), (64, 64, 1), (32, 32, 2), (16, 16, 16)]
centres = [(0, -500, 800), (1000, 500, 800), (-1200, -1000, 900)]
print('%-12s %8s %12s %12s' % ('region', 'voxels', 'texels/64vox', 'mean box'))
for sh in shapes:
    r = [touched(sh, c) for c in centres]
    print('%-12s %8d %12.1f %12.0f' % ('x'.join(map(str, sh)), sh[0] * sh[1] * sh[2],
                                       np.mean([a for a, _ in r]), np.mean([b for _, b in r])))
