"""Seeded test cases shared by the CPU tests, the GPU parity tests and
tests/golden/check_against_reference.py.  Inputs are always regenerated from the seed;
golden files hold only expected outputs (plus input checksums)."""
import copy

import numpy as np
import torch

from tempo_amd import synthetic as S

# name -> dict(kind, cfg, B, cube/sub sizes, heatmap kind, seed)
ROOT_CASES = {
    # reduced Panoptic-like: 5 ring cameras, 15 joints
    'root_panoptic_small': dict(cfg='panoptic', B=2, cube=[20, 20, 8], hm='blobs', seed=11),
    'root_panoptic_noise': dict(cfg='panoptic', B=1, cube=[16, 12, 10], hm='noise', seed=12),
    # H36M calibrations, 17 joints (channel padding 17 -> 20), square maps
    'root_h36m_small': dict(cfg='h36m', B=2, cube=[12, 12, 6], hm='noise', seed=13,
                            hm_size=[96, 96], img_size=[96, 96]),
    # per-batch grid centres with one skipped item (len-4 centre, [3] < 0)
    'root_percenter_skip': dict(cfg='panoptic', B=3, cube=[10, 10, 6], hm='blobs', seed=14,
                                centers='per_batch'),
}
WINDOW_CASES = {
    # space 2000 / sub-space 1000 / 16^3 windows -> 31^3 fine grid
    'win_panoptic_small': dict(cfg='panoptic', B=4, sub_cube=[16, 16, 16], hm='blobs',
                               seed=21, space=[2000, 2000, 2000], sub=[1000, 1000, 1000]),
    'win_h36m_noise': dict(cfg='h36m', B=3, sub_cube=[12, 12, 12], hm='noise', seed=22,
                           space=[2400, 2400, 1800], sub=[1200, 1200, 900],
                           hm_size=[96, 96], img_size=[96, 96]),
}


def _cfg(name):
    return copy.deepcopy(S.PANOPTIC if name == 'panoptic' else S.H36M)


def _cams(cfg_name, cfg, seed):
    if cfg_name == 'h36m':
        return S.h36m_cameras()
    return S.ring_cameras(cfg['num_views'], seed=seed, target=cfg['space_center'])


def build_root(name, device='cpu', full=None):
    """Returns dict(args=(feature_maps, grid_size, grid_center, cube_size, centers,
    scales, cameras), image_size, heatmap_size)."""
    spec = dict(ROOT_CASES[name]) if full is None else dict(full)
    cfg = _cfg(spec['cfg'])
    if 'hm_size' in spec:
        cfg['heatmap_size'], cfg['image_size'] = spec['hm_size'], spec['img_size']
    for key in ('num_views', 'num_joints'):  # ring-camera configs only
        if key in spec:
            cfg[key] = spec[key]
    B, seed = spec['B'], spec['seed']
    cams = _cams(spec['cfg'], cfg, seed)
    centers, scales, cameras = S.frame_meta(cams, B, cfg['raw_size'], cfg['image_size'])
    w, h = cfg['heatmap_size']
    fms = S.make_heatmaps(B, len(cams), cfg['num_joints'], h, w, kind=spec['hm'],
                          seed=seed, device=device)
    if spec.get('centers') == 'per_batch':
        rng = np.random.RandomState(seed)
        grid_center = []
        for b in range(B):
            c = [cfg['space_center'][i] + float(rng.uniform(-300, 300)) for i in range(3)]
            grid_center.append(c + [-1.0 if b == 1 else float(b)])
    else:
        grid_center = [cfg['space_center']]
    args = (fms, cfg['space_size'], grid_center, spec['cube'], centers, scales, cameras)
    return dict(args=args, image_size=cfg['image_size'], heatmap_size=cfg['heatmap_size'])


def build_window(name, device='cpu', full=None):
    """Returns dict(args=(feature_maps, grid_center, space_center, space_size,
    sub_space_size, sub_cube_size, centers, scales, cameras), ...)."""
    spec = dict(WINDOW_CASES[name]) if full is None else dict(full)
    cfg = _cfg(spec['cfg'])
    if 'hm_size' in spec:
        cfg['heatmap_size'], cfg['image_size'] = spec['hm_size'], spec['img_size']
    for key in ('num_views', 'num_joints'):  # ring-camera configs only
        if key in spec:
            cfg[key] = spec[key]
    B, seed = spec['B'], spec['seed']
    space = spec.get('space', cfg['space_size'])
    sub = spec.get('sub', cfg['sub_space_size'])
    cams = _cams(spec['cfg'], cfg, seed)
    centers, scales, cameras = S.frame_meta(cams, B, cfg['raw_size'], cfg['image_size'])
    w, h = cfg['heatmap_size']
    fms = S.make_heatmaps(B, len(cams), cfg['num_joints'], h, w, kind=spec['hm'],
                          seed=seed, device=device)
    props = S.make_proposals(B, cfg['space_center'], space, seed=seed,
                             invalid_every=spec.get('invalid_every', 3), margin=0.0)
    args = (fms, props.to(device), cfg['space_center'], space, sub, spec['sub_cube'],
            centers, scales, cameras)
    return dict(args=args, image_size=cfg['image_size'], heatmap_size=cfg['heatmap_size'])


def to_device(case, device):
    a = list(case['args'])
    a[0] = [t.to(device) for t in a[0]]
    if isinstance(a[1], torch.Tensor):
        a[1] = a[1].to(device)
    return dict(case, args=tuple(a))
