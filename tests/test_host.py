"""Host-side mirror of the reference interface: camera packing, geometry,
argument handling.  No GPU."""
import numpy as np
import pytest
import torch

import cases
import tempo_amd
from oracle import project_layer_ref as O
from tempo_amd import camera as C, project_layer as PL, synthetic as S


def _cams():
    return [(S.PANOPTIC, S.ring_cameras(5, seed=2)), (S.H36M, S.h36m_cameras())]


def test_packed_cameras_equal_the_oracles_camera_objects():
    for cfg, cams in _cams():
        centers, scales, cameras = S.frame_meta(cams, 2, cfg['raw_size'], cfg['image_size'])
        table = C.pack_cameras(centers, scales, cameras, cfg['image_size'])
        assert table.shape == (2, len(cams), 32) and table.dtype == torch.float32
        for v, cam in enumerate(cams):
            o = O.SimpleCameraTorch(cam, 'cpu')
            rec = table[1, v]
            assert torch.equal(rec[0:9].view(3, 3), o.R_w2c)
            assert torch.equal(rec[9:12], o.T_w2c.view(3))
            assert torch.equal(rec[12:18].view(2, 3), o.K.T.contiguous())
            assert torch.equal(rec[18:21], o.k[:3]) and torch.equal(rec[21:23], o.p)
            assert rec[23] == 1.0
            t = torch.as_tensor(O.get_affine_transform(centers[1][v], scales[1][v] / 200.0, 0,
                                                       cfg['image_size']), dtype=torch.float)
            assert torch.equal(rec[24:30].view(2, 3), t)
            assert rec[30] == centers[1][v][0] * 2 and rec[31] == centers[1][v][1] * 2


def test_pinned_flavour_matches_torch_on_this_host():
    from tempo_amd import numerics
    here = numerics.probe()['R']
    for cfg, cams in _cams():
        centers, scales, cameras = S.frame_meta(cams, 1, cfg['raw_size'], cfg['image_size'])
        auto = C.pack_cameras(centers, scales, cameras, cfg['image_size'], 'auto')
        pinned = C.pack_cameras(centers, scales, cameras, cfg['image_size'], here)
        assert torch.equal(auto, pinned)


def test_camera_without_distortion_and_bad_cameras():
    cam = dict(S.ring_cameras(1, seed=0, distortion=False)[0])
    centers, scales, cameras = S.frame_meta([cam], 1, (1920, 1080), [960, 512])
    rec = C.pack_cameras(centers, scales, cameras, [960, 512])[0, 0]
    assert rec[23] == 0.0 and float(rec[18:23].abs().sum()) == 0.0
    with pytest.raises(NotImplementedError):
        C.pack_cameras(centers, scales, [[dict(cam, intrinsics=1)]], [960, 512])
    bad = {k: v for k, v in cam.items() if k not in ('K', 'Kfull')}
    with pytest.raises(ValueError):
        C.pack_cameras(centers, scales, [[bad]], [960, 512])
    with pytest.raises(ValueError):
        C.pack_cameras(centers, scales, [[dict(cam, R=[[1.0, 0.0], [0.0, 1.0]])]], [960, 512])


def test_modules_keep_the_reference_interface():
    import inspect
    r = inspect.signature(tempo_amd.ProjectLayer.forward)
    assert list(r.parameters) == ['self', 'feature_maps', 'grid_size', 'grid_center',
                                  'cube_size', 'centers', 'scales', 'cameras']
    w = inspect.signature(tempo_amd.ProjectLayerWithMask.forward)
    assert list(w.parameters) == ['self', 'feature_maps', 'grid_center', 'space_center',
                                  'space_size', 'sub_space_size', 'sub_cube_size', 'centers',
                                  'scales', 'cameras']
    layer = tempo_amd.ProjectLayer(384, 96, 'mean')
    assert layer.image_size == [384, 384] and layer.heatmap_size == [96, 96]
    assert len(list(layer.parameters())) == 0
    with pytest.raises(NotImplementedError):
        tempo_amd.ProjectLayer([960, 512], [240, 128], 'learned')
    wl = tempo_amd.ProjectLayerWithMask([960, 512], [240, 128], 'mean')
    assert wl.fine_grid is None and wl.center_grid is None


def test_no_cpu_fallback():
    case = cases.build_root('root_panoptic_noise')
    layer = tempo_amd.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    with pytest.raises(RuntimeError, match='HIP devices only'):
        layer(*case['args'])
    case = cases.build_window('win_h36m_noise')
    wl = tempo_amd.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    with pytest.raises(RuntimeError, match='HIP devices only'):
        wl(*case['args'])


def test_window_geometry_equals_the_oracles():
    for name in cases.WINDOW_CASES:
        case = cases.build_window(name)
        _, gc, sc, ss, sub, cube, *_ = case['args']
        mine = tempo_amd.ProjectLayerWithMask(case['image_size'], case['heatmap_size'])
        ref = O.ProjectLayerWithMask(case['image_size'], case['heatmap_size'])
        mine.set_space_size(sc, ss, sub, cube, 'cpu')
        ref.set_space_size(sc, ss, sub, cube, 'cpu')
        for a in ('fine_voxels_per_axis', 'scale', 'bias', 'voxels_per_axis',
                  'whole_space_size', 'whole_space_center', 'ind_space_size'):
            assert torch.equal(getattr(mine, a), getattr(ref, a)), a
        for x, y in zip(mine.window_geometry(gc), ref.window_geometry(gc)):
            assert torch.equal(x, y)
        assert mine._fine_bins == [int(v) for v in ref.fine_voxels_per_axis]
        mine._ensure_grids('cpu')
        fine = O.compute_grid(ref.whole_space_size, ref.whole_space_center,
                              ref.fine_voxels_per_axis)
        assert torch.equal(mine.fine_grid, fine)


def test_panoptic_fine_grid_size():
    wl = tempo_amd.ProjectLayerWithMask([960, 512], [240, 128])
    p = S.PANOPTIC
    wl.set_space_size(p['space_center'], p['space_size'], p['sub_space_size'],
                      p['sub_cube_size'], 'cpu')
    assert wl._fine_bins == [253, 253, 64]


def test_root_axes_and_skip_flags():
    layer = tempo_amd.ProjectLayer([960, 512], [240, 128])
    axes, cube = layer._axes([8000, 8000, 2000], [[0, -500, 800]], [80, 80, 20], 4, 'cpu')
    g = O.compute_grid([8000, 8000, 2000], [0, -500, 800], [80, 80, 20])
    assert torch.equal(axes[2], g[:20, 2]) and torch.equal(axes[0], g[::1600, 0])
    assert torch.equal(layer.compute_grid([8000, 8000, 2000], [0, -500, 800], [8, 4, 3]),
                       O.compute_grid([8000, 8000, 2000], [0, -500, 800], [8, 4, 3]))
    gc = [[0., 0., 0., 1.], [5., 6., 7., -1.], [1., 2., 3., 0.]]
    assert layer._live(gc, 3, 'cpu').tolist() == [1, 0, 1]
    assert layer._live([[0, 0, 0]], 3, 'cpu') is None
    axes, _ = layer._axes(2000, gc, 4, 3, 'cpu')
    assert axes[0].shape == (3, 4)
    assert torch.equal(axes[1][1], torch.linspace(-1000, 1000, 4) + 6.0)


def test_share_groups_table():
    from tempo_amd import ops
    t = ops.share_groups(['a', 'a', 'a', 'b', 'b', 'a', 'c'])
    assert t.tolist() == [[0, 3], [3, 2], [5, 1], [6, 1]]
    assert ops.share_groups(['a', 'b', 'c']) is None
    assert ops.share_groups(['a'] * 5, max_items=2).tolist() == [[0, 2], [2, 2], [4, 1]]


def test_packed_table_cache_sees_replaced_calibrations():
    """pack_cameras keeps whole tables per call shape (identity of the per-frame lists +
    a fingerprint of every dict): same inputs -> the same table object; a replaced dict,
    centre or scale -> a fresh table equal to an uncached packing."""
    import copy
    from tempo_amd import camera, synthetic as S
    cfg = S.PANOPTIC
    cams = S.ring_cameras(5, seed=3, target=cfg['space_center'])
    c, s, m = S.frame_meta(cams, 4, cfg['raw_size'], cfg['image_size'])
    camera.clear_cache()
    a = camera.pack_cameras(c, s, m, cfg['image_size'])
    b = camera.pack_cameras(list(c), list(s), list(m), cfg['image_size'])  # new outer lists
    assert b is a
    m[2][1] = dict(m[2][1], T=[[10.0], [20.0], [30.0]])
    d = camera.pack_cameras(c, s, m, cfg['image_size'])
    assert d is not a and not torch.equal(d, a)
    assert torch.equal(d[:2], a[:2]) and torch.equal(d[3], a[3])
    c[1][0] = c[1][0] + 1.0
    e = camera.pack_cameras(c, s, m, cfg['image_size'])
    assert e is not d and not torch.equal(e[1, 0], d[1, 0])
    camera.clear_cache()
    f = camera.pack_cameras(copy.deepcopy(c), copy.deepcopy(s), copy.deepcopy(m), cfg['image_size'])
    assert torch.equal(f, e)


def test_packed_table_cache_sees_in_place_edits_and_other_value_types():
    """The fingerprint reads the bytes: a number edited IN PLACE inside a nested list, an
    ndarray field or a centre array changes the table; ndarray / numpy-scalar fields and torch
    centres (the per-item fallback of camera._blobs) are cached and watched just the same."""
    import copy
    import numpy as np
    from tempo_amd import camera, synthetic as S
    cfg = S.PANOPTIC
    cams = S.ring_cameras(5, seed=4, target=cfg['space_center'])
    c, s, m = S.frame_meta(cams, 3, cfg['raw_size'], cfg['image_size'])
    m = copy.deepcopy(m)
    camera.clear_cache()
    a = camera.pack_cameras(c, s, m, cfg['image_size'])
    assert camera.pack_cameras(c, s, m, cfg['image_size']) is a
    m[1][2]['T'][0][0] += 25.0  # nested python list, edited in place
    b = camera.pack_cameras(c, s, m, cfg['image_size'])
    # (the frames of a recording share their calibration objects: view 2 changes in all)
    assert b is not a and not torch.equal(b[1, 2], a[1, 2]) and torch.equal(b[:, 0], a[:, 0])
    c[2][3][0] += 4.0  # ndarray centre, edited in place
    d = camera.pack_cameras(c, s, m, cfg['image_size'])
    assert d is not b and not torch.equal(d[2, 3], b[2, 3])
    # ndarray fields and numpy scalars
    for frame in m:
        for cam in frame:
            for k in list(cam):
                if isinstance(cam[k], list):
                    cam[k] = np.asarray(cam[k], dtype=np.float32)
                elif isinstance(cam[k], float):
                    cam[k] = np.float32(cam[k])
    e = camera.pack_cameras(c, s, m, cfg['image_size'])
    assert camera.pack_cameras(c, s, m, cfg['image_size']) is e
    m[0][0]['R'][1, 1] += 0.125
    f = camera.pack_cameras(c, s, m, cfg['image_size'])
    assert f is not e and not torch.equal(f[0, 0], e[0, 0])
    # torch centres
    ct = [[torch.as_tensor(x) for x in row] for row in c]
    g = camera.pack_cameras(ct, s, m, cfg['image_size'])
    assert torch.equal(g, f) and camera.pack_cameras(ct, s, m, cfg['image_size']) is g
    ct[0][1][1] += 2.0
    h = camera.pack_cameras(ct, s, m, cfg['image_size'])
    assert h is not g and not torch.equal(h[0, 1], g[0, 1])
    camera.clear_cache()
