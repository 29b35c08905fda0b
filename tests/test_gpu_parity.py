"""GPU parity: HIP kernels (through the C ABI) vs the CPU oracle and the
committed golden fixtures.  Bars (BASELINE.json north_star): volume max-abs
error <= 1e-5 in fp32, valid-view mask bit-exact.  The kernels reproduce the
oracle's operation order, so on hosts where torch's CPU kernels contract FMAs
the way the probed build does the error is exactly 0; the assertion is the
stated tolerance."""
import os

import numpy as np
import pytest
import torch

import cases
from oracle import project_layer_ref as O

pytestmark = pytest.mark.gpu
TOL = 1e-5
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _bits(bounding):
    V = bounding.shape[-1]
    return (bounding.to(torch.int32) << torch.arange(V, dtype=torch.int32)).sum(-1)


def _gold_mode(gold):
    return dict(zip(('R', 'K', 'A'), (str(x) for x in gold['mm_mode'])))


def _report(name, got, ref):
    d = (got - ref).abs()
    nz = int((got != ref).sum())
    print(f'\n[{name}] max-abs {float(d.max()):.3e}  differing elements {nz}/{ref.numel()}')
    return float(d.max())


def test_library_loaded_from_tree():
    from tempo_amd import _lib
    lib = _lib.load()
    assert lib.tempo_version() == _lib.ABI_VERSION
    assert os.path.dirname(_lib.LIB_PATH).endswith(os.path.join('tempo_amd', 'lib'))


@pytest.mark.parametrize('name', list(cases.ROOT_CASES))
def test_root_vs_oracle_and_golden(name, gpu_device):
    import tempo_amd
    case = cases.build_root(name)
    ref_layer = O.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    ref_cubes, ref_grids = ref_layer(*case['args'])
    ref_mask = _bits(ref_layer.last_bounding)

    layer = tempo_amd.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    layer.keep_mask = True
    dcase = cases.to_device(case, gpu_device)
    with torch.no_grad():
        cubes, grids = layer(*dcase['args'])
    torch.cuda.synchronize()
    assert cubes.shape == ref_cubes.shape and grids.shape == ref_grids.shape
    err = _report(name, cubes.cpu(), ref_cubes)
    assert err <= TOL
    assert torch.equal(grids.cpu(), ref_grids)
    assert torch.equal(layer.last_mask.cpu().to(torch.int32), ref_mask), 'valid-view mask'

    # golden file: produced by the oracle on the build host, whose BLAS flavour
    # it records; ask the kernels for that flavour
    gold = np.load(os.path.join(GOLD, name + '.npz'))
    layer.mm_mode = _gold_mode(gold)
    with torch.no_grad():
        cubes, _ = layer(*dcase['args'])
    gerr = float(np.abs(cubes.cpu().numpy() - gold['cubes']).max())
    print(f'[{name}] vs golden: max-abs {gerr:.3e}')
    assert gerr <= TOL
    assert np.array_equal(layer.last_mask.cpu().numpy().astype(np.uint32), gold['mask'])


@pytest.mark.parametrize('name', list(cases.WINDOW_CASES))
def test_window_vs_oracle_and_golden(name, gpu_device):
    import tempo_amd
    case = cases.build_window(name)
    ref_layer = O.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    ref_cubes, ref_off = ref_layer(*case['args'])

    layer = tempo_amd.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    dcase = cases.to_device(case, gpu_device)
    with torch.no_grad():
        cubes, off = layer(*dcase['args'])
    torch.cuda.synchronize()
    assert cubes.shape == ref_cubes.shape
    err = _report(name, cubes.cpu(), ref_cubes)
    assert err <= TOL
    assert torch.equal(off.cpu(), ref_off)
    assert torch.equal(layer.center_grid.cpu(), ref_layer.center_grid)
    assert torch.equal(layer.fine_grid.cpu(), ref_layer.fine_grid)

    gold = np.load(os.path.join(GOLD, name + '.npz'))
    layer.mm_mode = _gold_mode(gold)
    with torch.no_grad():
        cubes, off = layer(*dcase['args'])
    gerr = float(np.abs(cubes.cpu().numpy() - gold['cubes']).max())
    print(f'[{name}] vs golden: max-abs {gerr:.3e}')
    assert gerr <= TOL
    assert np.array_equal(off.cpu().numpy(), gold['offset'])


def test_projection_chain_bit_exact(gpu_device):
    """pixel coordinates, sample-grid coordinates and the bounds test of the
    kernels equal the oracle's torch ops bit for bit (Panoptic-like + H36M)."""
    from tempo_amd import ops, synthetic as S
    from tempo_amd.camera import pack_cameras
    for cfg, cams in ((S.PANOPTIC, S.ring_cameras(5, seed=3)), (S.H36M, S.h36m_cameras())):
        centers, scales, cameras = S.frame_meta(cams, 1, cfg['raw_size'], cfg['image_size'])
        table = pack_cameras(centers, scales, cameras, cfg['image_size'])[0]
        grid = O.compute_grid(cfg['space_size'], cfg['space_center'], cfg['cube_size'])
        pix, g, inside = ops.project_points(grid.to(gpu_device), table.to(gpu_device),
                                            cfg['heatmap_size'], cfg['image_size'])
        w, h = cfg['heatmap_size']
        for v, cam in enumerate(cams):
            xy = O.SimpleCameraTorch(cam, 'cpu').world_to_pixel(grid)
            width, height = centers[0][v] * 2
            ins = (xy[:, 0] >= 0) & (xy[:, 1] >= 0) & (xy[:, 0] < width) & (xy[:, 1] < height)
            trans = torch.as_tensor(
                O.get_affine_transform(centers[0][v], scales[0][v] / 200.0, 0,
                                       cfg['image_size']), dtype=torch.float)
            sg = torch.clamp(O._sample_coords(xy, trans, w, h, cfg['image_size'], width,
                                              height, 'cpu'), -1.1, 1.1)
            assert torch.equal(inside[v].cpu().bool(), ins)
            # bit patterns, with every NaN mapped to one pattern (a voxel on a
            # camera's principal plane projects to NaN in the reference too)
            canon = lambda t: torch.nan_to_num(t, nan=-7777.0).view(torch.int32)
            assert torch.equal(canon(pix[v].cpu()), canon(xy))
            assert torch.equal(canon(g[v].cpu()), canon(sg))


def test_repack_matches_permute(gpu_device):
    from tempo_amd import ops
    torch.manual_seed(5)
    for (B, V, C, H, W) in ((2, 3, 15, 7, 37), (1, 2, 17, 16, 64), (2, 2, 32, 5, 13)):
        fms = [torch.rand(B, C, H, W, device=gpu_device) for _ in range(V)]
        out = ops.repack_nhwc(fms)
        cp = out.shape[-1]
        ref = torch.zeros(B, V, H, W, cp, device=gpu_device)
        ref[..., :C] = torch.stack(fms, 1).permute(0, 1, 3, 4, 2)
        assert torch.equal(out, ref)


def _mixed_calibration_case(seed=31, B=7, cube=(14, 10, 9)):
    """B frames from three recordings: calibrations A A A B B A C, distinct noise maps."""
    from tempo_amd import synthetic as S
    cfg = S.PANOPTIC
    rings = [S.ring_cameras(5, seed=seed + k, target=cfg['space_center']) for k in range(3)]
    which = [0, 0, 0, 1, 1, 0, 2][:B]
    metas = [S.frame_meta(rings[k], 1, cfg['raw_size'], cfg['image_size']) for k in which]
    centers = [m[0][0] for m in metas]
    scales = [m[1][0] for m in metas]
    cameras = [m[2][0] for m in metas]
    w, h = cfg['heatmap_size']
    fms = S.make_heatmaps(B, 5, cfg['num_joints'], h, w, kind='noise', seed=seed)
    return cfg, fms, centers, scales, cameras, list(cube)


def test_root_shared_projection_equals_per_item(gpu_device):
    """Grouped launch (one projection per run of frames with one calibration) vs the
    per-item launch: bit-identical volume, grids and mask; and vs the oracle."""
    import tempo_amd
    cfg, fms, centers, scales, cameras, cube = _mixed_calibration_case()
    # len-4 centres: item 1 (inside the first run) is skipped by the rule of :81
    gc = [list(cfg['space_center']) + [(-1.0 if b == 1 else float(b))] for b in range(len(centers))]
    ref_layer = O.ProjectLayer(cfg['image_size'], cfg['heatmap_size'], 'mean')
    ref_cubes, ref_grids = ref_layer(fms, cfg['space_size'], gc, cube, centers, scales, cameras)
    dfms = [t.to(gpu_device) for t in fms]
    outs = []
    for share in (True, False):
        layer = tempo_amd.ProjectLayer(cfg['image_size'], cfg['heatmap_size'], 'mean')
        layer.keep_mask = True
        layer.share_projection = share
        with torch.no_grad():
            cubes, grids = layer(dfms, cfg['space_size'], gc, cube, centers, scales, cameras)
        outs.append((cubes.cpu(), grids.cpu(), layer.last_mask.cpu()))
    canon = lambda t: torch.nan_to_num(t, nan=-7777.0).view(torch.int32)
    assert torch.equal(canon(outs[0][0]), canon(outs[1][0])), 'volume: grouped vs per-item'
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    assert float((outs[0][0][1]).abs().max()) == 0.0  # the skipped item stays zero
    err = _report('root_shared', outs[0][0], ref_cubes)
    assert err <= TOL
    assert torch.equal(outs[0][1], ref_grids)
    assert torch.equal(outs[0][2].to(torch.int32), _bits(ref_layer.last_bounding))


@pytest.mark.parametrize('cfgname', ['panoptic', 'h36m'])
def test_window_shared_projection_equals_per_item(cfgname, gpu_device):
    """T timesteps of one proposal (same window, same calibration, different maps):
    grouped launch vs per-item launch, bit-identical; both vs the oracle run per timestep."""
    import tempo_amd
    from tempo_amd import ops, synthetic as S
    from tempo_amd.camera import pack_cameras
    cfg = S.PANOPTIC if cfgname == 'panoptic' else dict(S.H36M, heatmap_size=[96, 96],
                                                        image_size=[96, 96])
    cams = (S.ring_cameras(5, seed=41, target=cfg['space_center']) if cfgname == 'panoptic'
            else S.h36m_cameras())
    T, P = 3, 3  # timesteps, proposals; windows ordered (proposal, timestep)
    V, J = len(cams), cfg['num_joints']
    w, h = cfg['heatmap_size']
    space, sub, sub_cube = [2400, 2400, 1800], [1200, 1200, 900], [12, 12, 12]
    fms = S.make_heatmaps(T, V, J, h, w, kind='noise', seed=42)  # frame = timestep
    centers, scales, cameras = S.frame_meta(cams, T, cfg['raw_size'], cfg['image_size'])
    props = S.make_proposals(P, cfg['space_center'], space, seed=43, invalid_every=3, margin=0.0)
    layer = tempo_amd.ProjectLayerWithMask(cfg['image_size'], cfg['heatmap_size'], 'mean')
    layer.set_space_size(cfg['space_center'], space, sub, sub_cube, gpu_device)
    tl, _, st, en = layer.window_geometry(props.to(gpu_device))
    layer._ensure_grids(gpu_device)
    rep = lambda t: t.to(torch.int32).repeat_interleave(T, 0).contiguous()
    win_frame = torch.arange(T, dtype=torch.int32, device=gpu_device).repeat(P).contiguous()
    live = (props[:, 3] >= 0).to(torch.uint8).repeat_interleave(T).to(gpu_device)
    cam = pack_cameras(centers, scales, cameras, cfg['image_size']).to(gpu_device)
    hm = ops.repack_nhwc([t.to(gpu_device) for t in fms])
    groups = torch.tensor([[p * T, T] for p in range(P)], dtype=torch.int32, device=gpu_device)
    kw = dict(win_frame=win_frame, live=live)
    a = ops.unproject_window(hm, cam, layer._fine_axes, rep(tl), rep(st), rep(en), sub_cube, J,
                             cfg['heatmap_size'], cfg['image_size'], groups=groups, **kw)
    b = ops.unproject_window(hm, cam, layer._fine_axes, rep(tl), rep(st), rep(en), sub_cube, J,
                             cfg['heatmap_size'], cfg['image_size'], **kw)
    torch.cuda.synchronize()
    assert torch.equal(a, b), 'window volume: grouped vs per-item'
    ref_layer = O.ProjectLayerWithMask(cfg['image_size'], cfg['heatmap_size'], 'mean')
    for t in range(T):
        # the reference's call for timestep t: every proposal against frame t
        f_t = [fm[t:t + 1].expand(P, -1, -1, -1).contiguous() for fm in fms]
        c3 = lambda x: [x[t]] * P
        ref, _ = ref_layer(f_t, props, cfg['space_center'], space, sub, sub_cube,
                           c3(centers), c3(scales), c3(cameras))
        got = a.view(P, T, *a.shape[1:])[:, t].cpu()
        assert float((got - ref).abs().max()) <= TOL


def _planes_ref(cubes):
    return cubes.max(dim=4)[0], cubes.max(dim=3)[0], cubes.max(dim=2)[0]


@pytest.mark.parametrize('name', list(cases.WINDOW_CASES))
@pytest.mark.parametrize('want_cubes', [False, True])
def test_window_planes_equal_axis_max(name, want_cubes, gpu_device):
    """Fused epilogue (tempo_pose.py:172-176): the three planes are bit-identical to
    torch.max over the volume -- the HIP volume and, within TOL, the oracle's."""
    import tempo_amd
    case = cases.build_window(name)
    dcase = cases.to_device(case, gpu_device)
    layer = tempo_amd.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    with torch.no_grad():
        cubes, off = layer(*dcase['args'])
        fms, gc = dcase['args'][0], dcase['args'][1]
        pxy, pxz, pyz, off2, cubes2 = layer.get_planes(fms, gc, *dcase['args'][6:],
                                                       want_cubes=want_cubes)
    torch.cuda.synchronize()
    assert torch.equal(off, off2)
    assert (cubes2 is not None) == want_cubes
    if want_cubes:
        assert torch.equal(cubes, cubes2)
    for got, ref in zip((pxy, pxz, pyz), _planes_ref(cubes)):
        assert got.shape == ref.shape
        assert torch.equal(torch.nan_to_num(got, nan=-1.0), torch.nan_to_num(ref, nan=-1.0))
    ref_layer = O.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    ref_cubes, _ = ref_layer(*case['args'])
    for got, ref in zip((pxy, pxz, pyz), _planes_ref(ref_cubes)):
        assert float((got.cpu() - ref).abs().max()) <= TOL


def test_root_forward_sequence_equals_loop(gpu_device):
    """tempo_detector.py:175-191: the loop over time_idx in one launch."""
    import tempo_amd
    T = 3
    cs = [cases.build_root('root_panoptic_small', full=dict(cases.ROOT_CASES['root_panoptic_small'],
                                                            seed=50 + t)) for t in range(T)]
    ds = [cases.to_device(c, gpu_device) for c in cs]
    layer = tempo_amd.ProjectLayer(cs[0]['image_size'], cs[0]['heatmap_size'], 'mean')
    with torch.no_grad():
        loop = torch.cat([layer(*d['args'])[0] for d in ds], dim=0)
        seq = layer.forward_sequence([d['args'][0] for d in ds], ds[0]['args'][1],
                                     [d['args'][2] for d in ds], ds[0]['args'][3],
                                     [d['args'][4] for d in ds], [d['args'][5] for d in ds],
                                     [d['args'][6] for d in ds])
    canon = lambda t: torch.nan_to_num(t, nan=-1.0)
    assert seq.shape == loop.shape and torch.equal(canon(seq), canon(loop))
    # timesteps already channel-last (packed by the producer / channels_last tensors) are read
    # in place, NCHW ones repacked: same volume
    from tempo_amd import ops
    mixed = [tempo_amd.PackedHeatmaps(ops.repack_nhwc(ds[0]['args'][0]), ds[0]['args'][0][0].shape[1]),
             [f.contiguous(memory_format=torch.channels_last) for f in ds[1]['args'][0]],
             ds[2]['args'][0]]
    with torch.no_grad():
        seq2 = layer.forward_sequence(mixed, ds[0]['args'][1], [d['args'][2] for d in ds],
                                      ds[0]['args'][3], [d['args'][4] for d in ds],
                                      [d['args'][5] for d in ds], [d['args'][6] for d in ds])
    assert torch.equal(canon(seq2), canon(seq))
    # and against the oracle's own loop over time_idx (tempo_detector.py:175-191)
    ref_layer = O.ProjectLayer(cs[0]['image_size'], cs[0]['heatmap_size'], 'mean')
    ref = torch.cat([ref_layer(*c['args'])[0] for c in cs], dim=0)
    assert torch.equal(seq.cpu().isnan(), ref.isnan())
    assert float((canon(seq.cpu()) - canon(ref)).abs().max()) <= TOL


def test_window_step_ten_proposals_per_frame_vs_oracle(gpu_device):
    """BASELINE config 3's batched shape: P proposals per frame (window j looks at frame
    j // P), frames sent through in chunks (ops.window_step) -- against the oracle called
    the reference's way, one proposal per batch item (tempo_pose.py:266 loops over n)."""
    import tempo_amd
    from tempo_amd import ops, synthetic as S
    from tempo_amd.camera import pack_cameras
    cfg = S.PANOPTIC
    F_, P = 3, 10
    space, sub, sub_cube = [2400, 2400, 1800], [1200, 1200, 900], [12, 12, 12]
    cams = [S.ring_cameras(5, seed=80 + b, target=cfg['space_center']) for b in range(F_)]
    metas = [S.frame_meta(cams[b], 1, cfg['raw_size'], cfg['image_size']) for b in range(F_)]
    centers = [m[0][0] for m in metas]
    scales = [m[1][0] for m in metas]
    cameras = [m[2][0] for m in metas]
    w, h = cfg['heatmap_size']
    J = cfg['num_joints']
    fms = S.make_heatmaps(F_, 5, J, h, w, kind='noise', seed=81)
    props = S.make_proposals(F_ * P, cfg['space_center'], space, seed=82, invalid_every=7,
                             margin=0.0).view(F_, P, 7)
    layer = tempo_amd.ProjectLayerWithMask(cfg['image_size'], cfg['heatmap_size'], 'mean')
    layer.set_space_size(cfg['space_center'], space, sub, sub_cube, gpu_device)
    flat = props.view(F_ * P, 7).to(gpu_device)  # ordered (frame, proposal)
    tl, _, st, en = layer.window_geometry(flat)
    layer._ensure_grids(gpu_device)
    geo = [t.to(torch.int32).contiguous() for t in (tl, st, en)]
    live = (flat[:, 3] >= 0).to(torch.uint8).contiguous()
    cam = pack_cameras(centers, scales, cameras, cfg['image_size']).to(gpu_device)
    dfms = [f.to(gpu_device) for f in fms]
    ref_layer = O.ProjectLayerWithMask(cfg['image_size'], cfg['heatmap_size'], 'mean')
    refs = [ref_layer(fms, props[:, n], cfg['space_center'], space, sub, sub_cube, centers,
                      scales, cameras)[0] for n in range(P)]
    ref = torch.stack(refs, dim=1)  # [F,P,C,12,12,12]
    for chunk in (1, 2, 3):
        got, _ = ops.window_step(dfms, cam, layer._fine_axes, *geo, sub_cube,
                                 cfg['heatmap_size'], cfg['image_size'], windows_per_frame=P,
                                 live=live, chunk=chunk)
        torch.cuda.synchronize()
        got = got.view(F_, P, *got.shape[1:]).cpu()
        err = _report(f'window_step P={P} chunk={chunk}', got, ref)
        assert err <= TOL
    assert float(ref.abs().max()) > 0.0
    # the fused axis-max on the same batched shape
    got, planes = ops.window_step(dfms, cam, layer._fine_axes, *geo, sub_cube,
                                  cfg['heatmap_size'], cfg['image_size'], windows_per_frame=P,
                                  live=live, planes=True, want_cubes=False)
    assert got is None
    for pl, dim in zip(planes, (5, 4, 3)):
        want = ref.max(dim=dim)[0].view(F_ * P, *pl.shape[1:])
        assert float((pl.cpu() - want).abs().max()) <= TOL


def test_window_forward_sequence_planes_equals_loops(gpu_device):
    """tempo_pose.py:145-176 + :266: candidates x timesteps x axis-max in one launch."""
    import tempo_amd
    from tempo_amd import synthetic as S
    cfg = S.PANOPTIC
    T, B, P = 2, 2, 3
    space, sub, sub_cube = [2000, 2000, 2000], [1000, 1000, 1000], [16, 16, 16]
    cams = [S.ring_cameras(5, seed=60 + b, target=cfg['space_center']) for b in range(B)]
    w, h = cfg['heatmap_size']
    fms_t = [[f.to(gpu_device) for f in S.make_heatmaps(B, 5, cfg['num_joints'], h, w,
                                                        kind='noise', seed=70 + t)]
             for t in range(T)]
    metas = [S.frame_meta(cams[b], 1, cfg['raw_size'], cfg['image_size']) for b in range(B)]
    centers = [m[0][0] for m in metas]
    scales = [m[1][0] for m in metas]
    cameras = [m[2][0] for m in metas]
    cand = S.make_proposals(B * P, cfg['space_center'], space, seed=61, invalid_every=4,
                            margin=0.0).view(B, P, 7).to(gpu_device)
    layer = tempo_amd.ProjectLayerWithMask(cfg['image_size'], cfg['heatmap_size'], 'mean')
    with torch.no_grad():
        pxy, pxz, pyz, off, cubes = layer.forward_sequence_planes(
            fms_t, cand, cfg['space_center'], space, sub, sub_cube, [centers] * T,
            [scales] * T, [cameras] * T, want_cubes=True)
        for n in range(P):
            for t in range(T):
                ref, ref_off = layer(fms_t[t], cand[:, n], cfg['space_center'], space, sub,
                                     sub_cube, centers, scales, cameras)
                assert torch.equal(cubes[:, n, t], ref)
                assert torch.equal(off[:, n], ref_off)
                assert torch.equal(pxy[:, n, t], ref.max(dim=4)[0])
                assert torch.equal(pxz[:, n, t], ref.max(dim=3)[0])
                assert torch.equal(pyz[:, n, t], ref.max(dim=2)[0])
    # the producer's output -- one channel-last buffer, frames ordered (t, b), handed over as
    # one PackedHeatmaps per timestep -- is used where it lies
    from tempo_amd import ops
    from tempo_amd.project_layer import _packed_sequence
    J = cfg['num_joints']
    buf = torch.cat([ops.repack_nhwc(fms_t[t]) for t in range(T)], 0)
    packed_t = [tempo_amd.PackedHeatmaps(buf[t * B:(t + 1) * B], J) for t in range(T)]
    assert _packed_sequence(packed_t) is not None
    assert _packed_sequence(packed_t[::-1]) is None
    with torch.no_grad():
        got = layer.forward_sequence_planes(
            packed_t, cand, cfg['space_center'], space, sub, sub_cube, [centers] * T,
            [scales] * T, [cameras] * T, want_cubes=True)
        got_rev = layer.forward_sequence_planes(  # (not one buffer in order: copied)
            [tempo_amd.PackedHeatmaps(p.hm.clone(), J) for p in packed_t], cand,
            cfg['space_center'], space, sub, sub_cube, [centers] * T, [scales] * T,
            [cameras] * T, want_cubes=False)
    for a, b in zip(got, (pxy, pxz, pyz, off, cubes)):
        assert torch.equal(a, b)
    for a, b in zip(got_rev[:4], (pxy, pxz, pyz, off)):
        assert torch.equal(a, b)
    # and against the oracle's loops over candidates and time_idx (tempo_pose.py:145-176,
    # :266): cubes, offset and the three axis-max planes
    ref_layer = O.ProjectLayerWithMask(cfg['image_size'], cfg['heatmap_size'], 'mean')
    cpu_fms = [[f.cpu() for f in fm] for fm in fms_t]
    for n in range(P):
        for t in range(T):
            ref, ref_off = ref_layer(cpu_fms[t], cand[:, n].cpu(), cfg['space_center'], space,
                                     sub, sub_cube, centers, scales, cameras)
            assert float((cubes[:, n, t].cpu() - ref).abs().max()) <= TOL
            assert torch.equal(off[:, n].cpu(), ref_off)
            for got, dim in ((pxy, 4), (pxz, 3), (pyz, 2)):
                assert float((got[:, n, t].cpu() - ref.max(dim=dim)[0]).abs().max()) <= TOL


def test_prepacked_camera_table_is_accepted(gpu_device):
    """A caller may pack the calibration of its sequence once and hand the table over."""
    import tempo_amd
    from tempo_amd.camera import pack_cameras
    case = cases.to_device(cases.build_root('root_panoptic_small'), gpu_device)
    args = list(case['args'])
    layer = tempo_amd.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    with torch.no_grad():
        ref, _ = layer(*args)
        table = pack_cameras(args[4], args[5], args[6], case['image_size']).to(gpu_device)
        got, _ = layer(*args[:4], None, None, table)
    assert torch.equal(torch.nan_to_num(got, nan=-1.0), torch.nan_to_num(ref, nan=-1.0))


@pytest.mark.parametrize('dims', [(6, 8, 64), (8, 6, 64), (10, 12, 64), (5, 4, 32), (3, 7, 20)])
def test_planes_kernel_dispatch_edges(dims, gpu_device):
    """Window shapes on both sides of the fused-planes kernel's conditions (64-voxel
    columns, VY a multiple of 4; x walks of 8 with a ragged end) against torch.max of the
    volume the plain kernel writes; windows partly outside the lattice, one skipped."""
    from tempo_amd import ops, synthetic as S
    from tempo_amd.camera import pack_cameras
    cfg = S.PANOPTIC
    cams = S.ring_cameras(5, seed=81, target=cfg['space_center'])
    F, NW, J = 2, 5, cfg['num_joints']
    w, h = cfg['heatmap_size']
    fms = [f.to(gpu_device) for f in S.make_heatmaps(F, 5, J, h, w, kind='noise', seed=82)]
    centers, scales, cameras = S.frame_meta(cams, F, cfg['raw_size'], cfg['image_size'])
    cam = pack_cameras(centers, scales, cameras, cfg['image_size']).to(gpu_device)
    hm = ops.repack_nhwc(fms)
    fine = [torch.linspace(-1500, 1500, 97) + c for c in cfg['space_center']]
    fine = [a.to(gpu_device) for a in fine]
    g = torch.Generator().manual_seed(83)
    tl = torch.randint(-10, 80, (NW, 3), generator=g, dtype=torch.int32)
    vox = torch.tensor(dims, dtype=torch.int32)
    lim = torch.tensor([97, 97, 97], dtype=torch.int32)
    start = torch.maximum(tl + torch.randint(0, 3, (NW, 3), generator=g, dtype=torch.int32),
                          torch.zeros_like(tl))
    end = torch.maximum(torch.minimum(tl + vox - torch.randint(0, 3, (NW, 3), generator=g,
                                                               dtype=torch.int32), lim), start + 1)
    dev = lambda t: t.contiguous().to(gpu_device)
    win_frame = dev(torch.tensor([0, 1, 1, 0, 1], dtype=torch.int32))
    live = dev(torch.tensor([1, 1, 0, 1, 1], dtype=torch.uint8))
    args = (hm, cam, fine, dev(tl), dev(start), dev(end), list(dims), J, cfg['heatmap_size'],
            cfg['image_size'])
    vol = ops.unproject_window(*args, win_frame=win_frame, live=live)
    cubes, pxy, pxz, pyz = ops.unproject_window_planes(*args, win_frame=win_frame, live=live,
                                                       want_cubes=True)
    _, qxy, qxz, qyz = ops.unproject_window_planes(*args, win_frame=win_frame, live=live)
    torch.cuda.synchronize()
    canon = lambda t: torch.nan_to_num(t, nan=-1.0)
    assert float(vol[2].abs().max()) == 0.0 and float(vol.abs().max()) > 0.0
    assert torch.equal(canon(cubes), canon(vol))
    for p, q, dim in ((pxy, qxy, 4), (pxz, qxz, 3), (pyz, qyz, 2)):
        ref = vol.max(dim=dim)[0]
        assert torch.equal(canon(p), canon(ref)) and torch.equal(canon(q), canon(ref))


def test_root_step_library_loop_equals_separate_calls(gpu_device):
    """tempo_root_step (chunk loop inside the library) vs repack + unproject on the whole
    batch: volume, grids and mask bit-identical, ragged last chunk, per-item centres, a
    skipped item."""
    from tempo_amd import ops, synthetic as S
    from tempo_amd.camera import pack_cameras
    cfg = S.PANOPTIC
    B, cube = 7, [12, 10, 9]
    cams = S.ring_cameras(5, seed=91, target=cfg['space_center'])
    centers, scales, cameras = S.frame_meta(cams, B, cfg['raw_size'], cfg['image_size'])
    cam = pack_cameras(centers, scales, cameras, cfg['image_size']).to(gpu_device)
    w, h = cfg['heatmap_size']
    fms = [f.to(gpu_device) for f in S.make_heatmaps(B, 5, cfg['num_joints'], h, w,
                                                     kind='noise', seed=92)]
    g = torch.Generator().manual_seed(93)
    axes = [(torch.linspace(-s / 2, s / 2, n)[None, :] + c +
             torch.rand(B, 1, generator=g) * 200).contiguous().to(gpu_device)
            for s, n, c in zip(cfg['space_size'], cube, cfg['space_center'])]
    live = torch.tensor([1, 1, 0, 1, 1, 1, 1], dtype=torch.uint8, device=gpu_device)
    hm = ops.repack_nhwc(fms)
    ref = ops.unproject_root(hm, cam, axes, cfg['num_joints'], cfg['heatmap_size'],
                             cfg['image_size'], live=live, want_grids=True, want_mask=True)
    for chunk, lib in ((3, True), (3, False), (8, True), (1, True)):
        got = ops.root_step(fms, cam, axes, cfg['heatmap_size'], cfg['image_size'], live=live,
                            want_grids=True, want_mask=True, chunk=chunk, in_library=lib)
        torch.cuda.synchronize()
        for a, b in zip(got, ref):
            assert torch.equal(torch.nan_to_num(a.float(), nan=-1.0),
                               torch.nan_to_num(b.float(), nan=-1.0)), (chunk, lib)
        # the same step with the fused z-max (tempo_root_step_zmax / per-chunk calls): every
        # entry is written, the dead item's included, whatever the buffer held before
        zmax = torch.full((B, cfg['num_joints'], cube[0], cube[1]), 7.0, device=gpu_device)
        got = ops.root_step(fms, cam, axes, cfg['heatmap_size'], cfg['image_size'], live=live,
                            chunk=chunk, in_library=lib, zmax_out=zmax)
        torch.cuda.synchronize()
        assert torch.equal(torch.nan_to_num(got[0], nan=-1.0), torch.nan_to_num(ref[0], nan=-1.0))
        assert torch.equal(torch.nan_to_num(zmax, nan=-1.0),
                           torch.nan_to_num(ref[0].max(dim=4)[0], nan=-1.0)), (chunk, lib)
        assert not zmax[2].any()


@pytest.mark.parametrize('name', list(cases.ROOT_CASES))
def test_root_zmax_equals_torch_max(name, gpu_device):
    """center_net.py:230: the z-max plane emitted by the kernel (columns shorter than a
    wave, straddling waves) is bit-identical to torch.max over the volume."""
    import tempo_amd
    case = cases.to_device(cases.build_root(name), gpu_device)
    layer = tempo_amd.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    layer.keep_zmax = True
    with torch.no_grad():
        cubes, _ = layer(*case['args'])
    canon = lambda t: torch.nan_to_num(t, nan=-1.0)
    assert layer.last_zmax.shape == cubes.shape[:4]
    assert torch.equal(canon(layer.last_zmax), canon(cubes.max(dim=4)[0]))


@pytest.mark.parametrize('z', [5, 24, 27, 64, 70])
def test_root_zmax_column_shapes(z, gpu_device):
    """The z-max's three code paths: waves of whole columns (z = 5: 60 of 64 lanes), columns
    that straddle waves with too many idle lanes otherwise (24, 27: segmented maximum + one
    atomic per segment), columns of a whole wave and longer (64, 70)."""
    import tempo_amd
    spec = dict(cases.ROOT_CASES['root_panoptic_noise'], cube=[6, 5, z], B=2, seed=400 + z)
    case = cases.to_device(cases.build_root('root_panoptic_noise', full=spec), gpu_device)
    layer = tempo_amd.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    layer.keep_zmax = True
    with torch.no_grad():
        cubes, _ = layer(*case['args'])
    canon = lambda t: torch.nan_to_num(t, nan=-1.0)
    assert cubes.shape[-1] == z
    assert torch.equal(canon(layer.last_zmax), canon(cubes.max(dim=4)[0]))
    plain = tempo_amd.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    with torch.no_grad():
        ref, _ = plain(*case['args'])
    assert torch.equal(canon(cubes), canon(ref))  # (the wave shape does not change the volume)


@pytest.mark.parametrize('variant', ['root', 'window'])
def test_extreme_heatmap_values_bit_exact(variant, gpu_device):
    """Channels scaled into the fp32 denormal range and up to the overflow threshold, a NaN
    patch and an infinity patch: the masked mean's double-precision division must round like
    the oracle's fp32 division everywhere (denormal quotients are redone with the plain
    division, NaN and inf pass through), bit for bit."""
    import tempo_amd
    from test_contract import _extreme_maps
    if variant == 'root':
        case = _extreme_maps(cases.build_root('root_panoptic_noise'))
        ref, _ = O.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')(*case['args'])
        layer = tempo_amd.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    else:
        case = _extreme_maps(cases.build_window('win_h36m_noise'))
        ref, _ = O.ProjectLayerWithMask(case['image_size'], case['heatmap_size'],
                                        'mean')(*case['args'])
        layer = tempo_amd.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    with torch.no_grad():
        got, _ = layer(*cases.to_device(case, gpu_device)['args'])
    got, ref = got.cpu().numpy(), ref.numpy()
    assert np.isnan(ref).any() and (np.abs(ref[np.isfinite(ref)]) < 1.2e-38).any()
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    same = np.nan_to_num(got, nan=-1.0).view(np.uint32) == np.nan_to_num(ref, nan=-1.0).view(np.uint32)
    print(f'[extreme {variant}] differing elements: {int((~same).sum())} of {same.size}')
    assert same.all()
