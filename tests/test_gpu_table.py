"""GPU parity of the footprint-table path (one projection per calibration, then a
gather-only kernel) and of heatmaps read in place from channel-last tensors, through the
C ABI.  Bars: bit-identical to the fused kernels (same operations, same order), and
<= 1e-5 / mask bit-exact against the CPU oracle."""
import pytest
import torch

import cases
from oracle import project_layer_ref as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _bits(bounding):
    V = bounding.shape[-1]
    return (bounding.to(torch.int32) << torch.arange(V, dtype=torch.int32)).sum(-1)


def _canon(t):
    return torch.nan_to_num(t.float(), nan=-7777.0)


def _mixed(seed=31, which=(0, 0, 0, 1, 1, 0, 2), cube=(14, 10, 9), views=5, joints=15,
           hm_kind='noise'):
    """frames from three recordings (calibrations interleaved), distinct maps"""
    from tempo_amd import synthetic as S
    cfg = dict(S.PANOPTIC, num_joints=joints, num_views=views)
    rings = [S.ring_cameras(views, seed=seed + k, target=cfg['space_center']) for k in range(3)]
    metas = [S.frame_meta(rings[k], 1, cfg['raw_size'], cfg['image_size']) for k in which]
    centers = [m[0][0] for m in metas]
    scales = [m[1][0] for m in metas]
    cameras = [m[2][0] for m in metas]
    w, h = cfg['heatmap_size']
    fms = S.make_heatmaps(len(which), views, joints, h, w, kind=hm_kind, seed=seed)
    return cfg, fms, centers, scales, cameras, list(cube)


def _axes(cfg, cube, dev):
    return [(torch.linspace(-s / 2, s / 2, n) + c).contiguous().to(dev)
            for s, n, c in zip(cfg['space_size'], cube, cfg['space_center'])]


@pytest.mark.parametrize('views,joints', [(5, 15), (10, 17), (4, 32), (13, 15), (32, 15)])
def test_root_table_equals_fused_and_oracle(views, joints, gpu_device):
    """calibrations A A A B B A C, a skipped frame: the table path (3 projections instead
    of 7) is bit-identical to the fused kernel -- volume, grids, mask, z-max -- and meets
    the oracle bar."""
    from tempo_amd import ops
    from tempo_amd.camera import pack_cameras
    dev = gpu_device
    cfg, fms, centers, scales, cameras, cube = _mixed(views=views, joints=joints)
    B = len(centers)
    cam_host = pack_cameras(centers, scales, cameras, cfg['image_size'])
    cam_cal, cal_of = ops.calibrations(cam_host)
    assert cam_cal.shape[0] == 3 and cal_of.tolist() == [0, 0, 0, 1, 1, 0, 2]
    axes = _axes(cfg, cube, dev)
    live = torch.tensor([1, 0, 1, 1, 1, 1, 1], dtype=torch.uint8, device=dev)
    hm = ops.repack_nhwc([f.to(dev) for f in fms])
    zref = torch.empty(B, joints, cube[0], cube[1], device=dev)
    ref = ops.unproject_root(hm, cam_host.to(dev), axes, joints, cfg['heatmap_size'],
                             cfg['image_size'], live=live, want_grids=True, want_mask=True,
                             zmax_out=zref)
    lay = ops.packed_layout(hm)
    table = ops.project_footprints(cam_cal.to(dev), axes, lay.H, lay.W, lay.texel_bytes,
                                   cfg['heatmap_size'], cfg['image_size'])
    zgot = torch.empty_like(zref)
    got = ops.gather_root(lay, table, 3, cube, joints, cal_of=cal_of.to(dev), axes=axes,
                          live=live, want_grids=True, want_mask=True, zmax_out=zgot)
    torch.cuda.synchronize()
    for a, b, what in zip(got, ref, ('cubes', 'grids', 'mask')):
        assert torch.equal(_canon(a), _canon(b)), what
    assert torch.equal(_canon(zgot), _canon(zref))
    assert float(got[0][1].abs().max()) == 0.0 and float(got[0][0].abs().max()) > 0.0
    # against the oracle (the skipped frame through a len-4 centre)
    gc = [list(cfg['space_center']) + [(-1.0 if b == 1 else float(b))] for b in range(B)]
    ref_layer = O.ProjectLayer(cfg['image_size'], cfg['heatmap_size'], 'mean')
    ref_cubes, ref_grids = ref_layer(fms, cfg['space_size'], gc, cube, centers, scales, cameras)
    tol = TOL if views < 16 else 1e-4  # V >= 16: torch's view-sum order only to rounding
    assert float((_canon(got[0].cpu()) - _canon(ref_cubes)).abs().max()) <= tol
    assert torch.equal(got[1].cpu(), ref_grids)
    assert torch.equal(got[2].cpu().to(torch.int32), _bits(ref_layer.last_bounding))


def test_root_step_shared_equals_root_step(gpu_device):
    """tempo_root_step_shared (table built inside the call, chunk loop, ragged last
    chunk) vs tempo_root_step."""
    from tempo_amd import ops
    from tempo_amd.camera import pack_cameras
    dev = gpu_device
    cfg, fms, centers, scales, cameras, cube = _mixed(seed=51)
    cam_host = pack_cameras(centers, scales, cameras, cfg['image_size'])
    cam_cal, cal_of = ops.calibrations(cam_host)
    axes = _axes(cfg, cube, dev)
    dfms = [f.to(dev) for f in fms]
    live = torch.tensor([1, 1, 1, 0, 1, 1, 1], dtype=torch.uint8, device=dev)
    ref = ops.root_step(dfms, cam_host.to(dev), axes, cfg['heatmap_size'], cfg['image_size'],
                        live=live, want_grids=True, want_mask=True, chunk=3)
    for chunk in (3, 8, 1):
        got = ops.root_step_shared(dfms, cam_cal.to(dev), cal_of.to(dev), axes,
                                   cfg['heatmap_size'], cfg['image_size'], live=live,
                                   want_grids=True, want_mask=True, chunk=chunk)
        torch.cuda.synchronize()
        for a, b, what in zip(got, ref, ('cubes', 'grids', 'mask')):
            assert torch.equal(_canon(a), _canon(b)), (what, chunk)
    # a calibration index outside the table: that frame reads as skipped
    bad = cal_of.clone()
    bad[2] = 9
    got = ops.root_step_shared(dfms, cam_cal.to(dev), bad.to(dev), axes, cfg['heatmap_size'],
                               cfg['image_size'], chunk=4)
    assert float(got[0][2].abs().max()) == 0.0
    assert torch.equal(_canon(got[0][0]), _canon(ref[0][0]))


@pytest.mark.parametrize('name', list(cases.WINDOW_CASES))
def test_window_table_equals_fused_and_oracle(name, gpu_device):
    """fine-lattice footprint table + gather vs the fused window kernel (bit-identical)
    and vs the oracle; 3 proposals per frame through win_frame."""
    import tempo_amd
    from tempo_amd import ops
    from tempo_amd.camera import pack_cameras
    dev = gpu_device
    case = cases.build_window(name)
    fms, props, space_center, space, sub, sub_cube, centers, scales, cameras = case['args']
    F_ = fms[0].shape[0]
    J = fms[0].shape[1]
    layer = tempo_amd.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    layer.set_space_size(space_center, space, sub, sub_cube, dev)
    tl, _, st, en = layer.window_geometry(props.to(dev))
    layer._ensure_grids(dev)
    geo = [t.to(torch.int32).contiguous() for t in (tl, st, en)]
    live = (props[:, 3] >= 0).to(torch.uint8).to(dev)
    cam_host = pack_cameras(centers, scales, cameras, case['image_size'])
    cam_cal, cal_of = ops.calibrations(cam_host)
    hm = ops.repack_nhwc([f.to(dev) for f in fms])
    ref = ops.unproject_window(hm, cam_host.to(dev), layer._fine_axes, *geo, sub_cube, J,
                               case['heatmap_size'], case['image_size'], live=live)
    lay = ops.packed_layout(hm)
    table = ops.project_footprints(cam_cal.to(dev), layer._fine_axes, lay.H, lay.W,
                                   lay.texel_bytes, case['heatmap_size'], case['image_size'],
                                   window_mode=True)
    fine_dims = [int(a.numel()) for a in layer._fine_axes]
    got = ops.gather_window(lay, table, cam_cal.shape[0], fine_dims, *geo, sub_cube, J,
                            cal_of=cal_of.to(dev), live=live)
    torch.cuda.synchronize()
    assert torch.equal(_canon(got), _canon(ref))
    ref_cubes, _ = O.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')(
        *case['args'])
    assert float((got.cpu() - ref_cubes).abs().max()) <= TOL
    # every window against every frame (win_frame), still bit-identical to the fused kernel
    P = 3
    rep = lambda t: t.repeat(P, 1).contiguous()
    wf = (torch.arange(P * F_, device=dev) % F_).to(torch.int32).contiguous()
    a = ops.unproject_window(hm, cam_host.to(dev), layer._fine_axes, rep(geo[0]), rep(geo[1]),
                             rep(geo[2]), sub_cube, J, case['heatmap_size'], case['image_size'],
                             win_frame=wf, live=live.repeat(P).contiguous())
    b = ops.gather_window(lay, table, cam_cal.shape[0], fine_dims, rep(geo[0]), rep(geo[1]),
                          rep(geo[2]), sub_cube, J, cal_of=cal_of.to(dev), win_frame=wf,
                          live=live.repeat(P).contiguous())
    torch.cuda.synchronize()
    assert torch.equal(_canon(a), _canon(b))


@pytest.mark.parametrize('joints,extra', [(15, 0), (17, 0), (15, 17), (32, 0)])
def test_channels_last_maps_read_in_place(joints, extra, gpu_device):
    """V channels_last tensors read in place (texel = 4*C_total bytes: 60 B at 15 joints,
    not a multiple of the 16-byte fetch) vs the repacked path: bit-identical for the fused
    root and window kernels and for the table path; small maps so that the corner texels
    are sampled.  `extra` leading channels exercise the channel-slice offset."""
    import tempo_amd
    from tempo_amd import ops, synthetic as S
    from tempo_amd.camera import pack_cameras
    dev = gpu_device
    cfg = dict(S.PANOPTIC, heatmap_size=[24, 14], num_joints=joints)
    V, B, cube = 5, 3, [12, 12, 6]
    cams = S.ring_cameras(V, seed=71, target=cfg['space_center'])
    centers, scales, cameras = S.frame_meta(cams, B, cfg['raw_size'], cfg['image_size'])
    cam_host = pack_cameras(centers, scales, cameras, cfg['image_size'])
    cam = cam_host.to(dev)
    C_total = joints + extra
    g = torch.Generator().manual_seed(72)
    nchw = [torch.rand(B, C_total, 14, 24, generator=g).to(dev) for _ in range(V)]
    nhwc = [f.contiguous(memory_format=torch.channels_last) for f in nchw]
    assert all(ops.is_channels_last(f) for f in nhwc)
    axes = _axes(cfg, cube, dev)
    hm = ops.repack_nhwc(nchw, extra, joints)
    ref = ops.unproject_root(hm, cam, axes, joints, cfg['heatmap_size'], cfg['image_size'],
                             want_grids=True, want_mask=True)
    lay = ops.channels_last_layout(nhwc, extra)
    got = ops.unproject_root_views(lay, cam, axes, joints, cfg['heatmap_size'],
                                   cfg['image_size'], want_grids=True, want_mask=True)
    torch.cuda.synchronize()
    for a, b, what in zip(got, ref, ('cubes', 'grids', 'mask')):
        assert torch.equal(_canon(a), _canon(b)), what
    assert float(ref[0].max()) > 0.0
    # the table path on the in-place layout
    cam_cal, cal_of = ops.calibrations(cam_host)
    table = ops.project_footprints(cam_cal.to(dev), axes, lay.H, lay.W, lay.texel_bytes,
                                   cfg['heatmap_size'], cfg['image_size'])
    got2 = ops.gather_root(lay, table, 1, cube, joints, want_mask=True)
    torch.cuda.synchronize()
    assert torch.equal(_canon(got2[0]), _canon(ref[0])) and torch.equal(got2[2], ref[2])
    # windows
    space, sub, sub_cube = [2400, 2400, 1800], [1200, 1200, 900], [12, 12, 12]
    layer = tempo_amd.ProjectLayerWithMask(cfg['image_size'], cfg['heatmap_size'], 'mean')
    layer.set_space_size(cfg['space_center'], space, sub, sub_cube, dev)
    props = S.make_proposals(B, cfg['space_center'], space, seed=73, invalid_every=0, margin=0.0)
    tl, _, st, en = layer.window_geometry(props.to(dev))
    layer._ensure_grids(dev)
    geo = [t.to(torch.int32).contiguous() for t in (tl, st, en)]
    wref = ops.unproject_window(hm, cam, layer._fine_axes, *geo, sub_cube, joints,
                                cfg['heatmap_size'], cfg['image_size'])
    wgot = ops.unproject_window_views(lay, cam, layer._fine_axes, *geo, sub_cube, joints,
                                      cfg['heatmap_size'], cfg['image_size'])
    torch.cuda.synchronize()
    assert torch.equal(_canon(wgot), _canon(wref))
    assert float(wref.max()) > 0.0


def test_root_table_config2_shape(gpu_device):
    """BASELINE config 2 at full size (32 frames, 80x80x20, one calibration): table path
    vs the chunked fused step, bit-identical; one frame against the oracle."""
    from tempo_amd import ops, synthetic as S
    from tempo_amd.camera import pack_cameras
    dev = gpu_device
    cfg = S.PANOPTIC
    B, V, J = 32, cfg['num_views'], cfg['num_joints']
    w, h = cfg['heatmap_size']
    g = torch.Generator(device=dev).manual_seed(5)
    fms = [torch.rand(B, J, h, w, device=dev, generator=g) for _ in range(V)]
    cams = S.ring_cameras(V, seed=5, target=cfg['space_center'])
    centers, scales, cameras = S.frame_meta(cams, B, cfg['raw_size'], cfg['image_size'])
    cam_host = pack_cameras(centers, scales, cameras, cfg['image_size'])
    cam_cal, cal_of = ops.calibrations(cam_host)
    assert cam_cal.shape[0] == 1
    axes = _axes(cfg, cfg['cube_size'], dev)
    ref, _, rmask = ops.root_step(fms, cam_host.to(dev), axes, cfg['heatmap_size'],
                                  cfg['image_size'], want_mask=True)
    got, _, gmask = ops.root_step_shared(fms, cam_cal.to(dev), None, axes, cfg['heatmap_size'],
                                         cfg['image_size'], want_mask=True)
    torch.cuda.synchronize()
    assert torch.equal(_canon(got), _canon(ref)) and torch.equal(gmask, rmask)
    layer = O.ProjectLayer(cfg['image_size'], cfg['heatmap_size'], 'mean')
    b = 17
    oc, _ = layer([f[b:b + 1].cpu() for f in fms], cfg['space_size'], [cfg['space_center']],
                  cfg['cube_size'], [centers[b]], [scales[b]], [cameras[b]])
    assert torch.equal(got[b:b + 1].cpu().isnan(), oc.isnan())
    assert float((_canon(got[b:b + 1].cpu()) - _canon(oc)).abs().max()) <= TOL


def test_gather_zcolumns_equals_the_detectors_torch_gather(gpu_device):
    """tempo_detector.py:204-206, restated with torch ops, vs the kernel (bit-equal)."""
    from tempo_amd import ops
    dev = gpu_device
    g = torch.Generator().manual_seed(3)
    for (B, C, X, Y, Z, K) in ((4, 15, 80, 80, 20, 10), (2, 32, 7, 5, 9, 3), (1, 1, 2, 2, 1, 4)):
        cubes = torch.rand(B, C, X, Y, Z, generator=g).to(dev)
        index = torch.randint(0, X * Y, (B, K), generator=g).to(dev)
        ref = torch.gather(torch.flatten(cubes, 2, 3).permute(0, 2, 1, 3), dim=1,
                           index=index.view(B, -1, 1, 1).repeat(1, 1, C, Z))
        got = ops.gather_zcolumns(cubes, index)
        assert got.shape == ref.shape and torch.equal(got, ref)
    bad = torch.tensor([[0, X * Y, -1, 1]], device=dev)
    got = ops.gather_zcolumns(cubes, bad)
    assert float(got[0, 1].abs().max()) == 0.0 and float(got[0, 2].abs().max()) == 0.0
    assert torch.equal(got[0, 0], cubes[0, :, 0, 0, :]) and torch.equal(got[0, 3], cubes[0, :, 0, 1, :])
