"""The root detector's path without the volume (root half), through the C
ABI: z-max planes with ``cubes = NULL`` (``center_net.py:230``) and the K z-columns computed
from the heatmaps (``tempo_detector.py:204-206``).  Bars: bit-identical to ``torch.max`` /
``tempo_gather_zcolumns`` on the volume the plain call writes, and <= 1e-5 (in practice 0)
against the detector's own torch expression evaluated on the CPU oracle's volume."""
import pytest
import torch

import cases
from oracle import project_layer_ref as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _canon(t):
    return torch.nan_to_num(t.float(), nan=-7777.0)


def _setup(dev, B=5, cube=(12, 10, 9), joints=15, extra=0, seed=500, batched=True, hw=(14, 24)):
    from tempo_amd import synthetic as S
    from tempo_amd.camera import pack_cameras
    h, w = hw
    cfg = dict(S.PANOPTIC, heatmap_size=[w, h], num_joints=joints)
    cams = S.ring_cameras(5, seed=seed, target=cfg['space_center'])
    centers, scales, cameras = S.frame_meta(cams, B, cfg['raw_size'], cfg['image_size'])
    cam = pack_cameras(centers, scales, cameras, cfg['image_size']).to(dev)
    g = torch.Generator().manual_seed(seed + 1)
    nchw = [torch.rand(B, joints + extra, h, w, generator=g).to(dev) for _ in range(5)]
    shift = (torch.rand(B, 1, generator=g) * 200) if batched else torch.zeros(1)
    axes = [(torch.linspace(-s / 2, s / 2, n)[None, :] + c + shift).contiguous().to(dev)
            if batched else (torch.linspace(-s / 2, s / 2, n) + c).contiguous().to(dev)
            for s, n, c in zip(cfg['space_size'], cube, cfg['space_center'])]
    return cfg, cam, nchw, axes, (centers, scales, cameras)


@pytest.mark.parametrize('z', [5, 9, 24, 64, 70])
def test_zmax_without_the_volume(z, gpu_device):
    """cubes = NULL in tempo_unproject_root_zmax / tempo_root_step_zmax /
    tempo_unproject_root_views_zmax: the z-max planes equal torch.max of the volume the plain
    call writes (whole-column waves, segmented maxima with atomics, columns of a wave and
    longer), a skipped item's planes are zero whatever the buffer held."""
    from tempo_amd import ops
    dev = gpu_device
    B, cube = 5, (6, 5, z)
    cfg, cam, nchw, axes, _ = _setup(dev, B=B, cube=cube, seed=510 + z)
    live = torch.tensor([1, 0, 1, 1, 1], dtype=torch.uint8, device=dev)
    J = cfg['num_joints']
    hm = ops.repack_nhwc(nchw)
    vol, _, _ = ops.unproject_root(hm, cam, axes, J, cfg['heatmap_size'], cfg['image_size'],
                                   live=live)
    ref = vol.max(dim=4)[0]
    assert float(ref.max()) > 0.0 and not ref[1].any()
    new = lambda: torch.full((B, J, cube[0], cube[1]), 7.0, device=dev)
    z1 = new()
    out = ops.unproject_root(hm, cam, axes, J, cfg['heatmap_size'], cfg['image_size'],
                             live=live, zmax_out=z1, want_cubes=False)
    assert out[0] is None
    assert torch.equal(_canon(z1), _canon(ref))
    for chunk, lib in ((2, True), (2, False), (8, True)):
        z2 = new()
        out = ops.root_step(nchw, cam, axes, cfg['heatmap_size'], cfg['image_size'], live=live,
                            chunk=chunk, in_library=lib, zmax_out=z2, want_cubes=False)
        assert out[0] is None
        assert torch.equal(_canon(z2), _canon(ref)), (chunk, lib)
    nhwc = [f.contiguous(memory_format=torch.channels_last) for f in nchw]
    z3 = new()
    ops.unproject_root_views(ops.channels_last_layout(nhwc), cam, axes, J, cfg['heatmap_size'],
                             cfg['image_size'], live=live, zmax_out=z3, want_cubes=False)
    torch.cuda.synchronize()
    assert torch.equal(_canon(z3), _canon(ref))


@pytest.mark.parametrize('joints,extra,batched', [(15, 0, True), (17, 3, False), (32, 0, True)])
def test_zcolumns_from_the_heatmaps(joints, extra, batched, gpu_device):
    """tempo_unproject_root_columns (repacked maps), _columns_views (channel-last in place) and
    tempo_root_step_columns (the reference's NCHW tensors, read plane by plane): bit-identical
    to tempo_gather_zcolumns on the written volume; cells outside the grid and skipped items
    give zeros; a channel slice of wider tensors."""
    from tempo_amd import ops
    dev = gpu_device
    B, cube, K = 4, (9, 7, 11), 6
    cfg, cam, nchw, axes, _ = _setup(dev, B=B, cube=cube, joints=joints, extra=extra,
                                     seed=530 + joints, batched=batched)
    live = torch.tensor([1, 1, 0, 1], dtype=torch.uint8, device=dev)
    hm = ops.repack_nhwc(nchw, extra, joints)
    vol, _, _ = ops.unproject_root(hm, cam, axes, joints, cfg['heatmap_size'],
                                   cfg['image_size'], live=live)
    g = torch.Generator().manual_seed(5)
    index = torch.randint(0, cube[0] * cube[1], (B, K), generator=g)
    index[0, 1], index[3, 2] = -1, cube[0] * cube[1]       # outside the grid: zeros
    index[1, 0], index[1, 5] = 0, cube[0] * cube[1] - 1    # the grid's corners
    index = index.to(dev)
    ref = ops.gather_zcolumns(vol, index)
    assert float(ref.max()) > 0.0 and not ref[2].any() and not ref[0, 1].any()
    kw = dict(live=live)
    got = ops.unproject_root_columns(hm, cam, axes, index, cfg['heatmap_size'],
                                     cfg['image_size'], channels=joints, **kw)
    assert got.shape == ref.shape and torch.equal(_canon(got), _canon(ref)), 'repacked'
    nhwc = [f.contiguous(memory_format=torch.channels_last) for f in nchw]
    got = ops.unproject_root_columns(ops.channels_last_layout(nhwc, extra), cam, axes, index,
                                     cfg['heatmap_size'], cfg['image_size'], channels=joints,
                                     **kw)
    assert torch.equal(_canon(got), _canon(ref)), 'channel-last in place'
    got = ops.unproject_root_columns(nchw, cam, axes, index, cfg['heatmap_size'],
                                     cfg['image_size'], channels=joints, c_begin=extra, **kw)
    torch.cuda.synchronize()
    assert torch.equal(_canon(got), _canon(ref)), 'NCHW planes'


def test_detector_path_without_the_volume_vs_oracle(gpu_device):
    """The detector's sequence (tempo_detector.py:187-206) on the drop-in module --
    forward_zmax, top-K cells, zcolumns -- against the same torch expressions on the CPU
    oracle's volume; Panoptic-like case with a skipped item and per-item centres, and the
    full-size root grid of config 1."""
    import tempo_amd
    for name, full in (('root_percenter_skip', None), ('root_panoptic_small', None),
                       ('root_panoptic_noise', dict(cases.ROOT_CASES['root_panoptic_noise'],
                                                    cube=[80, 80, 20]))):
        case = cases.build_root(name, full=full)
        ref_cubes, _ = O.ProjectLayer(case['image_size'], case['heatmap_size'],
                                      'mean')(*case['args'])
        B, C, X, Y, Z = ref_cubes.shape
        ref_zmax = ref_cubes.max(dim=4)[0]
        K = 10
        index = ref_zmax.sum(1).flatten(1).topk(K)[1]  # some plausible top-K cells
        ref_cols = torch.gather(torch.flatten(ref_cubes, 2, 3).permute(0, 2, 1, 3), dim=1,
                                index=index.view(B, -1, 1, 1).repeat(1, 1, C, Z))
        layer = tempo_amd.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
        dcase = cases.to_device(case, gpu_device)
        with torch.no_grad():
            zmax = layer.forward_zmax(*dcase['args'])
            cols = layer.zcolumns(*dcase['args'], index.to(gpu_device))
            cubes, _ = layer(*dcase['args'])
        torch.cuda.synchronize()
        assert zmax.shape == ref_zmax.shape and cols.shape == ref_cols.shape
        ez = float((_canon(zmax.cpu()) - _canon(ref_zmax)).abs().max())
        ec = float((_canon(cols.cpu()) - _canon(ref_cols)).abs().max())
        print(f'\n[{name}] z-max max-abs {ez:.3e}, z-columns max-abs {ec:.3e}')
        assert ez <= TOL and ec <= TOL
        # and bit-identical to the module's own volume
        assert torch.equal(_canon(zmax), _canon(cubes.max(dim=4)[0]))
        assert torch.equal(_canon(cols), _canon(
            torch.gather(torch.flatten(cubes, 2, 3).permute(0, 2, 1, 3), dim=1,
                         index=index.to(gpu_device).view(B, -1, 1, 1).repeat(1, 1, C, Z))))


def test_fused_consumers_under_autograd(gpu_device):
    """no shortcut may silently cut the autograd graph.  The module-level consumers
    (forward_zmax, zcolumns, get_planes) fall back to forward() + the torch expression when a
    feature map requires a gradient -- gradients equal autograd through the oracle's op chain;
    the raw forward-only ops (gather_zcolumns on a volume that requires grad, PackedHeatmaps
    that require grad) raise."""
    import tempo_amd
    from tempo_amd import ops
    case = cases.build_root('root_panoptic_noise')
    ref_fms = [f.clone().requires_grad_(True) for f in case['args'][0]]
    ref_cubes, _ = O.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')(
        ref_fms, *case['args'][1:])
    B, C, X, Y, Z = ref_cubes.shape
    g = torch.Generator().manual_seed(4)
    wz = torch.randn(B, C, X, Y, generator=g)
    index = torch.randint(0, X * Y, (B, 4), generator=g)
    wc = torch.randn(B, 4, C, Z, generator=g)
    ref_cols = torch.gather(torch.flatten(ref_cubes, 2, 3).permute(0, 2, 1, 3), dim=1,
                            index=index.view(B, -1, 1, 1).repeat(1, 1, C, Z))
    ((ref_cubes.max(dim=4)[0] * wz).sum() + (ref_cols * wc).sum()).backward()

    dcase = cases.to_device(case, gpu_device)
    layer = tempo_amd.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    fms = [f.clone().requires_grad_(True) for f in dcase['args'][0]]
    calls = []
    plain_get_voxel = layer.get_voxel
    layer.get_voxel = lambda *a, **k: (calls.append(1), plain_get_voxel(*a, **k))[1]
    zmax = layer.forward_zmax(fms, *dcase['args'][1:])
    cols = layer.zcolumns(fms, *dcase['args'][1:], index.to(gpu_device))
    assert zmax.requires_grad and cols.requires_grad
    # the detector's two calls on the same inputs share ONE differentiable volume
    assert len(calls) == 1
    other = [f.clone().requires_grad_(True) for f in dcase['args'][0]]
    layer.zcolumns(other, *dcase['args'][1:], index.to(gpu_device))  # other tensors: no reuse
    assert len(calls) == 2
    layer.zcolumns(other, *dcase['args'][1:], index.to(gpu_device))  # unchanged: reused
    assert len(calls) == 2
    with torch.no_grad():
        other[0].add_(0.0)  # edited in place since (its version moved): no reuse
    layer.zcolumns(other, *dcase['args'][1:], index.to(gpu_device))
    assert len(calls) == 3
    layer.get_voxel = plain_get_voxel
    ((zmax * wz.to(gpu_device)).sum() + (cols * wc.to(gpu_device)).sum()).backward()
    for a, b in zip(fms, ref_fms):
        assert float((a.grad.cpu() - b.grad).abs().max()) <= 2e-5 * max(1.0, float(b.grad.abs().max()))
    cubes, _ = layer(fms, *dcase['args'][1:])
    with pytest.raises(NotImplementedError):
        ops.gather_zcolumns(cubes, index.to(gpu_device))
    with torch.no_grad():
        ops.gather_zcolumns(cubes, index.to(gpu_device))
    packed = tempo_amd.PackedHeatmaps(
        ops.repack_nhwc(dcase['args'][0]).requires_grad_(True), 15)
    with pytest.raises(NotImplementedError):
        layer(packed, *dcase['args'][1:])
    with torch.no_grad():
        layer(packed, *dcase['args'][1:])

    # window planes: get_planes under autograd == oracle volume + torch.max
    wcase = cases.build_window('win_h36m_noise')
    rf = [f.clone().requires_grad_(True) for f in wcase['args'][0]]
    rc, _ = O.ProjectLayerWithMask(wcase['image_size'], wcase['heatmap_size'], 'mean')(
        rf, *wcase['args'][1:])
    ws = [torch.randn(rc.max(dim=d)[0].shape, generator=g) for d in (4, 3, 2)]
    sum((rc.max(dim=d)[0] * w).sum() for d, w in zip((4, 3, 2), ws)).backward()
    wl = tempo_amd.ProjectLayerWithMask(wcase['image_size'], wcase['heatmap_size'], 'mean')
    dw = cases.to_device(wcase, gpu_device)
    gf = [f.clone().requires_grad_(True) for f in dw['args'][0]]
    wl.set_space_size(*dw['args'][2:6], gpu_device)
    planes = wl.get_planes(gf, dw['args'][1], *dw['args'][6:])
    sum((p * w.to(gpu_device)).sum() for p, w in zip(planes[:3], ws)).backward()
    for a, b in zip(gf, rf):
        assert float((a.grad.cpu() - b.grad).abs().max()) <= 2e-5 * max(1.0, float(b.grad.abs().max()))
