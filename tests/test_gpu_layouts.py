"""GPU parity of heatmaps read in place from channel-last tensors (no repack) and of the
z-column gather, through the C ABI.  Bars: bit-identical to the repacked path (same
operations, same order) / to the detector's torch.gather."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _canon(t):
    return torch.nan_to_num(t.float(), nan=-7777.0)


def _axes(cfg, cube, dev):
    return [(torch.linspace(-s / 2, s / 2, n) + c).contiguous().to(dev)
            for s, n, c in zip(cfg['space_size'], cube, cfg['space_center'])]


@pytest.mark.parametrize('joints,extra', [(15, 0), (17, 0), (15, 17), (32, 0)])
def test_channels_last_maps_read_in_place(joints, extra, gpu_device):
    """V channels_last tensors read in place (texel = 4*C_total bytes: 60 B at 15 joints,
    not a multiple of the 16-byte fetch) vs the repacked path: bit-identical for the fused
    root and window kernels; small maps so that the corner texels
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


@pytest.mark.parametrize('T,V,B,K,C,H,W', [(1, 2, 2, 256, 15, 16, 24), (2, 3, 1, 64, 32, 9, 13),
                                           (1, 1, 2, 100, 17, 8, 8), (1, 5, 1, 256, 15, 128, 240)])
def test_conv1x1_nhwc_producer(T, V, B, K, C, H, W, gpu_device):
    """process_layer (rnn_multiview_pose.py:61-66, :321) with channel-last output: against
    F.conv2d evaluated in float64 (bar: 2e-6 * sum|w||x|-scale, i.e. fp32 FMA-chain
    rounding; torch's own fp32 conv is held to the same bar), layout = what repack_nhwc
    produces from the chunked NCHW result, padding channels zero."""
    import torch.nn.functional as F
    from tempo_amd import ops
    dev = gpu_device
    g = torch.Generator().manual_seed(11)
    x = torch.randn(T * V * B, K, H, W, generator=g).to(dev)
    w = (torch.randn(C, K, 1, 1, generator=g) / K ** 0.5).to(dev)
    bias = torch.randn(C, generator=g).to(dev)
    got = ops.conv1x1_nhwc(x, w, bias, num_views=V, timesteps=T)
    cp = ops.padded_channels(C)
    assert got.shape == (T * B, V, H, W, cp)
    ref64 = F.conv2d(x.double(), w.double(), bias.double())
    ref32 = F.conv2d(x, w, bias)
    # feature_maps[t][v] = chunk(chunk(map, T)[t], V)[v]  ->  [T*B, V, H, W, C]
    def regroup(m):
        return m.view(T, V, B, C, H, W).permute(0, 2, 1, 4, 5, 3).reshape(T * B, V, H, W, C)
    scale = float((w.double().abs().view(1, C, K, 1, 1) * x.double().abs().unsqueeze(1)).sum(2).max())
    err = float((got[..., :C].double() - regroup(ref64)).abs().max())
    err32 = float((regroup(ref32).double() - regroup(ref64)).abs().max())
    print(f'\n[conv1x1 K={K} C={C}] max-abs vs fp64: ours {err:.3e}, torch fp32 {err32:.3e}, scale {scale:.2f}')
    assert err <= 2e-6 * scale
    assert float(got[..., C:].abs().max()) == 0.0 if cp > C else True
    # the same tensor the repack produces from the NCHW feature maps (values to rounding)
    fms = [regroup(ref32)[:B, v].permute(0, 3, 1, 2).contiguous() for v in range(V)]
    hm = ops.repack_nhwc(fms)
    assert hm.shape == got[:B].shape
    assert float((hm - got[:B]).abs().max()) <= 4e-6 * scale


def test_modules_take_channels_last_and_packed_maps(gpu_device):
    """The drop-in modules with (i) NCHW lists (repack), (ii) channels_last lists (read in
    place), (iii) PackedHeatmaps: bit-identical to each other, and <= 1e-5 / mask bit-exact
    against the oracle; root (mask, grids, z-max) and window."""
    import cases
    import tempo_amd
    from oracle import project_layer_ref as O
    dev = gpu_device
    case = cases.build_root('root_panoptic_noise')
    dcase = cases.to_device(case, dev)
    fms = dcase['args'][0]
    rest = dcase['args'][1:]
    ref_layer = O.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    ref, ref_grids = ref_layer(*case['args'])
    from tempo_amd import ops
    variants = {
        'nchw': fms,
        'channels_last': [f.contiguous(memory_format=torch.channels_last) for f in fms],
        'packed': tempo_amd.PackedHeatmaps(ops.repack_nhwc(fms), fms[0].shape[1]),
    }
    outs = {}
    for name, maps in variants.items():
        layer = tempo_amd.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
        layer.keep_mask = True
        layer.keep_zmax = True  # (channel-last maps: tempo_unproject_root_views_zmax)
        with torch.no_grad():
            cubes, grids = layer(maps, *rest)
        outs[name] = (cubes, grids, layer.last_mask)
        if layer.keep_zmax:
            assert torch.equal(_canon(layer.last_zmax), _canon(cubes.max(dim=4)[0]))
    for name in ('channels_last', 'packed'):
        for a, b in zip(outs[name], outs['nchw']):
            assert torch.equal(_canon(a), _canon(b)), name
    assert float((_canon(outs['packed'][0].cpu()) - _canon(ref)).abs().max()) <= 1e-5
    assert torch.equal(outs['packed'][1].cpu(), ref_grids)
    V = ref_layer.last_bounding.shape[-1]
    bits = (ref_layer.last_bounding.to(torch.int32) << torch.arange(V, dtype=torch.int32)).sum(-1)
    assert torch.equal(outs['packed'][2].cpu().to(torch.int32), bits)
    # PackedHeatmaps still reads like the reference's list of [B,C,H,W] tensors
    pk = variants['packed']
    assert len(pk) == len(fms) and all(torch.equal(pk[v], fms[v]) for v in range(len(fms)))

    case = cases.build_window('win_h36m_noise')  # 17 joints: 68-byte texels in place
    dcase = cases.to_device(case, dev)
    fms = dcase['args'][0]
    rest = dcase['args'][1:]
    ref, ref_off = O.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')(
        *case['args'])
    variants = {
        'nchw': fms,
        'channels_last': [f.contiguous(memory_format=torch.channels_last) for f in fms],
        'packed': tempo_amd.PackedHeatmaps(ops.repack_nhwc(fms), fms[0].shape[1]),
    }
    outs = {}
    for name, maps in variants.items():
        layer = tempo_amd.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
        with torch.no_grad():
            outs[name] = layer(maps, *rest)
    for name in ('channels_last', 'packed'):
        assert torch.equal(outs[name][0], outs['nchw'][0]) and torch.equal(outs[name][1], outs['nchw'][1])
    assert float((outs['packed'][0].cpu() - ref).abs().max()) <= 1e-5
    assert torch.equal(outs['packed'][1].cpu(), ref_off)


def test_process_layer_nhwc_feeds_the_project_layer(gpu_device):
    """rnn_multiview_pose.py:321-326 + tempo_detector.py:187: backbone map -> process_layer
    -> feature_maps[t] -> ProjectLayer, with ProcessLayerNHWC + PackedHeatmaps against
    nn.Conv2d + the oracle (the conv's rounding differs from torch's by ~1e-6, so the volume
    is held to 2e-5 here; the unprojection itself is exact, see the other tests)."""
    import tempo_amd
    from oracle import project_layer_ref as O
    from tempo_amd import synthetic as S
    dev = gpu_device
    cfg = dict(S.PANOPTIC, heatmap_size=[60, 32])
    T, V, B, K, C = 2, 5, 2, 64, 15
    torch.manual_seed(21)
    layer2d = tempo_amd.ProcessLayerNHWC(K, C).to(dev)
    ref2d = torch.nn.Conv2d(K, C, 1)
    ref2d.load_state_dict({k: v.cpu() for k, v in layer2d.state_dict().items()})
    x = torch.rand(T * V * B, K, 32, 60) * 0.2
    cams = S.ring_cameras(V, seed=22, target=cfg['space_center'])
    centers, scales, cameras = S.frame_meta(cams, B, cfg['raw_size'], cfg['image_size'])
    cube = [14, 12, 8]
    with torch.no_grad():
        ref_map = ref2d(x)
        ref_fms = [list(torch.chunk(m, V)) for m in torch.chunk(ref_map, T)]
        got_fms = layer2d(x.to(dev), num_views=V, timesteps=T)
    assert len(got_fms) == T and isinstance(got_fms[0], tempo_amd.PackedHeatmaps)
    proj = tempo_amd.ProjectLayer(cfg['image_size'], cfg['heatmap_size'], 'mean')
    oracle = O.ProjectLayer(cfg['image_size'], cfg['heatmap_size'], 'mean')
    for t in range(T):
        for v in range(V):
            assert float((got_fms[t][v].cpu() - ref_fms[t][v]).abs().max()) <= 1e-5
        with torch.no_grad():
            cubes, _ = proj(got_fms[t], cfg['space_size'], [cfg['space_center']], cube, centers,
                            scales, cameras)
        ref, _ = oracle(ref_fms[t], cfg['space_size'], [cfg['space_center']], cube, centers,
                        scales, cameras)
        assert float(ref.max()) > 0.0
        assert float((_canon(cubes.cpu()) - _canon(ref)).abs().max()) <= 2e-5
    # training mode: autograd through F.conv2d, the reference's list format
    xg = x.to(dev).requires_grad_(True)
    lists = layer2d(xg, num_views=V, timesteps=T)
    assert isinstance(lists[0], list) and lists[0][0].shape == (B, C, 32, 60)


@pytest.mark.parametrize('cols', [7, 9])
def test_window_geometry_kernel_equals_the_tensor_ops(cols, gpu_device):
    """tempo_window_geometry against the reference-shaped tensor ops of
    ProjectLayerWithMask.window_geometry (project_layer.py:299-324): integers, offset and
    skip flags bit for bit, including proposals outside the space, half-way rounding cases
    and bounding boxes wider than the window."""
    import tempo_amd
    from tempo_amd import ops, synthetic as S
    cfg = S.PANOPTIC
    layer = tempo_amd.ProjectLayerWithMask(cfg['image_size'], cfg['heatmap_size'], 'mean')
    layer.set_space_size(cfg['space_center'], cfg['space_size'], cfg['sub_space_size'],
                         cfg['sub_cube_size'], gpu_device)
    g = torch.Generator().manual_seed(5)
    B = 4096
    gc = torch.zeros(B, cols)
    span = torch.tensor(cfg['space_size'], dtype=torch.float32)
    gc[:, :3] = (torch.rand(B, 3, generator=g) * 1.6 - 0.8) * span + \
        torch.tensor(cfg['space_center'], dtype=torch.float32)
    # exact half-way cases of the rounding: (k + 0.5 - bias) / scale
    k = torch.arange(0, 200, dtype=torch.float32)
    gc[:200, 0] = (k + 0.5 - layer.bias.cpu()[0]) / layer.scale.cpu()[0]
    if cols >= 7:
        gc[:, 3] = torch.randint(-1, 4, (B,), generator=g).float()
        gc[:, 4] = torch.rand(B, generator=g)
        gc[:, 5:7] = torch.rand(B, 2, generator=g) * 2.4 - 0.7  # also < 0 and > 1
    gc = gc.to(gpu_device)
    tl, off, st, en = layer.window_geometry(gc)
    tl2, off2, st2, en2, live2 = ops.window_geometry(
        gc, layer.scale, layer.bias, layer._wss_f, layer._iss_f, layer._fine_bins,
        layer._sub_cube)
    assert torch.equal(tl.to(torch.int32), tl2) and torch.equal(st.to(torch.int32), st2)
    assert torch.equal(en.to(torch.int32), en2)
    assert torch.equal(off.view(torch.int32), off2.view(torch.int32))
    assert torch.equal((gc[:, 3] >= 0).to(torch.uint8), live2)


@pytest.mark.parametrize('want_cubes', [False, True])
def test_window_planes_on_maps_in_place(want_cubes, gpu_device):
    """tempo_unproject_window_planes_views: the pose regressor's axis-max planes
    (tempo_pose.py:172-176) from channels_last tensors and from PackedHeatmaps, no repack,
    bit-identical to the NCHW path and to torch.max over the HIP volume."""
    import cases
    import tempo_amd
    from tempo_amd import ops
    case = cases.to_device(cases.build_window('win_panoptic_small'), gpu_device)
    fms, gc = case['args'][0], case['args'][1]
    rest = case['args'][6:]
    variants = {
        'nchw': fms,
        'channels_last': [f.contiguous(memory_format=torch.channels_last) for f in fms],
        'packed': tempo_amd.PackedHeatmaps(ops.repack_nhwc(fms), fms[0].shape[1]),
    }
    outs = {}
    for name, maps in variants.items():
        layer = tempo_amd.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
        with torch.no_grad():
            cubes, _ = layer(maps, *case['args'][1:])
            outs[name] = layer.get_planes(maps, gc, *rest, want_cubes=want_cubes)
        torch.cuda.synchronize()
        pxy, pxz, pyz, _, cubes2 = outs[name]
        assert (cubes2 is not None) == want_cubes
        if want_cubes:
            assert torch.equal(_canon(cubes), _canon(cubes2))
        for got, dim in ((pxy, 4), (pxz, 3), (pyz, 2)):
            assert torch.equal(_canon(got), _canon(cubes.max(dim=dim)[0])), (name, dim)
    for name in ('channels_last', 'packed'):
        for a, b in zip(outs[name][:3], outs['nchw'][:3]):
            assert torch.equal(_canon(a), _canon(b)), name


def test_channel_slices_at_the_end_of_an_allocation(gpu_device):
    """a channel slice of a channels_last tensor is handed over as a shifted
    pointer; the fetch lanes cover CP channels, so the last texels of the last frame used to
    be read up to 4 * (CP - C) bytes past the tensor.  The buffer range of a map now ends with
    the slice's last channel (per-dword range check: tools/range_check.hip).  40 channels go
    through the module as slices of 32 + 8 (CP = 16 for the second: 32 bytes that do not
    exist); the tensors sit at the very end of their allocations."""
    import tempo_amd
    from tempo_amd import synthetic as S
    dev = gpu_device
    cfg = dict(S.PANOPTIC, heatmap_size=[24, 14], num_joints=40)
    V, B, C, H, W = 5, 2, 40, 14, 24
    cams = S.ring_cameras(V, seed=81, target=cfg['space_center'])
    centers, scales, cameras = S.frame_meta(cams, B, cfg['raw_size'], cfg['image_size'])
    g = torch.Generator().manual_seed(82)
    nchw = [torch.rand(B, C, H, W, generator=g).to(dev) for _ in range(V)]
    nhwc = []
    for f in nchw:
        raw = torch.empty(B * H * W * C, dtype=torch.float32, device=dev)  # exactly the bytes
        t = torch.as_strided(raw, (B, C, H, W), (H * W * C, 1, W * C, C))
        t.copy_(f)
        assert t.data_ptr() + t.numel() * 4 == raw.data_ptr() + raw.numel() * 4
        nhwc.append(t)
    layer = tempo_amd.ProjectLayer(cfg['image_size'], cfg['heatmap_size'], 'mean')
    args = (cfg['space_size'], [cfg['space_center']], [10, 9, 7], centers, scales, cameras)
    with torch.no_grad():
        ref, _ = layer(nchw, *args)
        got, _ = layer(nhwc, *args)
    torch.cuda.synchronize()
    assert torch.equal(_canon(got), _canon(ref)) and float(ref[:, 32:].max()) > 0.0
