"""The HIP path against the two full-size fixtures the REFERENCE module produced
(tests/golden/config1_full.npz, config3_windows.npz: outputs of the reference's own
tempo/models/project_layer.py, see tests/golden/check_against_reference.py), with the matmul
flavour pinned from the file -- no oracle run in between -- and the full batch shapes of
BASELINE configs 3 and 4 through size-independent properties.  Bars: 1e-5 max-abs on the
volume, mask / offset / center_grid bit-exact."""
import hashlib
import os

import numpy as np
import pytest
import torch

import cases

pytestmark = pytest.mark.gpu
TOL = 1e-5
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _gold_mode(gold):
    return dict(zip(('R', 'K', 'A'), (str(x) for x in gold['mm_mode'])))


def _same(a, b):
    return torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0))


def test_hip_root_vs_reference_fixture_config1(gpu_device):
    """BASELINE config 1 at full size (80x80x20, one frame): HIP volume vs the strided sample
    of the reference module's volume, valid-view mask bit for bit."""
    import tempo_amd
    from test_oracle import _full_case
    case = _full_case('root')
    gold = np.load(os.path.join(GOLD, 'config1_full.npz'))
    layer = tempo_amd.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    layer.keep_mask = True
    layer.mm_mode = _gold_mode(gold)
    with torch.no_grad():
        cubes, _ = layer(*cases.to_device(case, gpu_device)['args'])
    got = cubes.cpu().numpy()
    s = int(gold['stride'])
    ref = gold['cubes_strided']
    sub = got[:, :, ::s, ::s, :]
    assert np.array_equal(np.isnan(sub), np.isnan(ref))
    err = float(np.nanmax(np.abs(sub - ref)))
    sha = hashlib.sha256(got.tobytes()).hexdigest() == str(gold['cubes_sha256'])
    print(f'\n[config1 fixture] max-abs {err:.3e}; whole volume sha256 equal: {sha}')
    assert err <= TOL
    assert np.array_equal(layer.last_mask.cpu().numpy().astype(np.uint32), gold['mask'])


def test_hip_window_vs_reference_fixture_config3(gpu_device):
    """BASELINE config 3's geometry (253x253x64 lattice, 64^3 windows): HIP volume vs the
    strided sample of the reference module's, offset and center_grid bit for bit."""
    import tempo_amd
    from test_oracle import _full_case
    case = _full_case('window')
    gold = np.load(os.path.join(GOLD, 'config3_windows.npz'))
    layer = tempo_amd.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    layer.mm_mode = _gold_mode(gold)
    with torch.no_grad():
        cubes, offset = layer(*cases.to_device(case, gpu_device)['args'])
    got = cubes.cpu().numpy()
    s = int(gold['stride'])
    err = float(np.abs(got[:, :, ::s, ::s, ::s] - gold['cubes_strided']).max())
    sha = hashlib.sha256(got.tobytes()).hexdigest() == str(gold['cubes_sha256'])
    print(f'\n[config3 fixture] max-abs {err:.3e}; whole volume sha256 equal: {sha}')
    assert err <= TOL
    assert abs(float(got.astype(np.float64).sum()) - float(gold['cubes_sum'])) <= 1e-3 * got.size * TOL
    assert np.array_equal(offset.cpu().numpy(), gold['offset'])
    assert np.array_equal(layer.center_grid.cpu().numpy(), gold['center_grid'])


def _window_batch(dev, cfg_name, frames, per_frame, seed, invalid_every):
    """Full-size window batch built on the device: (layer, fms, cam, geometry, live, props)."""
    import tempo_amd
    from tempo_amd import synthetic as S
    from tempo_amd.camera import pack_cameras
    cfg = S.PANOPTIC if cfg_name == 'panoptic' else S.H36M
    cams = S.h36m_cameras() if cfg_name == 'h36m' else S.ring_cameras(
        cfg['num_views'], seed=seed, target=cfg['space_center'])
    centers, scales, cameras = S.frame_meta(cams, frames, cfg['raw_size'], cfg['image_size'])
    cam = pack_cameras(centers, scales, cameras, cfg['image_size']).to(dev)
    w, h = cfg['heatmap_size']
    g = torch.Generator(device=dev).manual_seed(seed)
    fms = [torch.rand(frames, cfg['num_joints'], h, w, device=dev, generator=g)
           for _ in range(len(cams))]
    layer = tempo_amd.ProjectLayerWithMask(cfg['image_size'], cfg['heatmap_size'], 'mean')
    layer.set_space_size(cfg['space_center'], cfg['space_size'], cfg['sub_space_size'],
                         cfg['sub_cube_size'], dev)
    props = S.make_proposals(frames * per_frame, cfg['space_center'], cfg['space_size'],
                             seed=seed, invalid_every=invalid_every, margin=0.1).to(dev)
    tl, _, st, en = layer.window_geometry(props)
    layer._ensure_grids(dev)
    geo = [t.to(torch.int32).contiguous() for t in (tl, st, en)]
    live = (props[:, 3] >= 0).to(torch.uint8).contiguous()
    return cfg, layer, fms, cam, geo, live, props, (centers, scales, cameras)


@pytest.mark.parametrize('cfg_name,frames,per_frame', [('panoptic', 32, 10), ('h36m', 128, 1)])
def test_full_window_batches_properties(cfg_name, frames, per_frame, gpu_device):
    """BASELINE config 3 at its full 320 windows (10 proposals x 32 frames) and config 4's
    128-window slice (H36M, 384^2 maps, 17 joints), through the library's chunked step: every
    sampled window equals the same window computed alone through the drop-in module (its own
    launch shape, its own repack), skipped windows are exactly zero, values lie in [0, 1]."""
    from tempo_amd import ops
    dev = gpu_device
    cfg, layer, fms, cam, geo, live, props, meta = _window_batch(
        dev, cfg_name, frames, per_frame, seed=700 + frames, invalid_every=7)
    NW = frames * per_frame
    out, _ = ops.window_step(fms, cam, layer._fine_axes, *geo, layer._sub_cube,
                             cfg['heatmap_size'], cfg['image_size'],
                             windows_per_frame=per_frame, live=live)
    torch.cuda.synchronize()
    assert out.shape == (NW, cfg['num_joints'], 64, 64, 64)
    dead = (live == 0).nonzero().flatten().tolist()
    assert dead and len(dead) < NW
    for i in dead:
        assert not out[i].any(), f'skipped window {i} is not zero'
    # (a voxel on a camera's principal plane is NaN, here as in the reference: a handful)
    assert int(out.isnan().sum()) <= NW * cfg['num_joints'] * 8
    fin = torch.nan_to_num(out, nan=0.5)
    assert float(fin.min()) >= 0.0 and float(fin.max()) <= 1.0 and float(fin.max()) > 0.0
    del fin
    centers, scales, cameras = meta
    picks = sorted({0, 1, NW // 3, NW // 2 + 1, NW - 2, NW - 1, dead[0]})
    for i in picks:
        f = i // per_frame
        with torch.no_grad():
            one, _ = layer([m[f:f + 1] for m in fms], props[i:i + 1], cfg['space_center'],
                           cfg['space_size'], cfg['sub_space_size'], cfg['sub_cube_size'],
                           centers[f:f + 1], scales[f:f + 1], cameras[f:f + 1])
        assert _same(one[0], out[i]), f'window {i} differs from the same window computed alone'
    # the fused planes of the same batch (volume never written) equal torch.max of the volume
    _, planes = ops.window_step(fms, cam, layer._fine_axes, *geo, layer._sub_cube,
                                cfg['heatmap_size'], cfg['image_size'],
                                windows_per_frame=per_frame, live=live, planes=True,
                                want_cubes=False)
    torch.cuda.synchronize()
    for p, dim in zip(planes, (4, 3, 2)):
        for i0 in range(0, NW, 64):  # slice by slice: torch.max over 5 GB at once is slow
            assert _same(p[i0:i0 + 64], out[i0:i0 + 64].max(dim=dim)[0])
