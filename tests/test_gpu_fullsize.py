"""GPU parity at BASELINE.json's full sizes and on the shapes the small cases do
not reach: the oracle on whole frames where it finishes in seconds, and
size-independent properties (batch independence, exact scaling, joint
permutation, mask/volume consistency, z-slab reassembly) on the batch-32 bench
configuration.  Bars as in test_gpu_parity.py: 1e-5 max-abs, mask bit-exact."""
import pytest
import torch

import cases
from oracle import project_layer_ref as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _bits(bounding):
    V = bounding.shape[-1]
    return (bounding.to(torch.int32) << torch.arange(V, dtype=torch.int32)).sum(-1)


def _maxabs(got, ref):
    """max-abs error; a voxel on a camera's principal plane is NaN in the reference
    (0/0 in the perspective divide) and must be NaN here too"""
    assert torch.equal(got.isnan(), ref.isnan()), 'NaN positions differ'
    return float((torch.nan_to_num(got) - torch.nan_to_num(ref)).abs().max())


def _same(a, b):
    """bitwise-equal volumes, NaN (see _maxabs) counted as equal to NaN"""
    return torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0))


def _root_pair(spec, dev, keep_mask=True):
    import tempo_amd
    case = cases.build_root(None, full=spec)
    layer = tempo_amd.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    layer.keep_mask = keep_mask
    return case, cases.to_device(case, dev), layer


def _check_root(spec, dev):
    case, dcase, layer = _root_pair(spec, dev)
    ref_layer = O.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    ref, ref_grids = ref_layer(*case['args'])
    with torch.no_grad():
        got, grids = layer(*dcase['args'])
    err = _maxabs(got.cpu(), ref)
    nz = int((torch.nan_to_num(got.cpu()) != torch.nan_to_num(ref)).sum())
    print(f'\n[{spec}] max-abs {err:.3e}, differing {nz}/{ref.numel()}')
    assert err <= TOL
    assert torch.equal(grids.cpu(), ref_grids)
    assert torch.equal(layer.last_mask.cpu().to(torch.int32), _bits(ref_layer.last_bounding))


def test_config1_one_frame_full_root_grid(gpu_device):
    """BASELINE configs[0]: Panoptic cam5, 1 frame, 15 joints, 128x240 maps, 80x80x20."""
    _check_root(dict(cfg='panoptic', B=1, cube=[80, 80, 20], hm='noise', seed=101), gpu_device)


def test_root_zmax_full_grid(gpu_device):
    """80x80x20 (20-voxel columns, 3.2 per wave) and 128^3 z-slab of 16: z-max vs torch.max."""
    for spec in (dict(cfg='panoptic', B=2, cube=[80, 80, 20], hm='noise', seed=110),
                 dict(cfg='panoptic', B=1, cube=[32, 32, 64], hm='noise', seed=111),
                 dict(cfg='panoptic', B=1, cube=[24, 24, 7], hm='blobs', seed=112)):
        case, dcase, layer = _root_pair(spec, gpu_device)
        layer.keep_zmax = True
        with torch.no_grad():
            cubes, _ = layer(*dcase['args'])
        assert _same(layer.last_zmax, cubes.max(dim=4)[0])


@pytest.mark.parametrize('views', [7, 10, 13])
def test_dense_many_views(views, gpu_device):
    """configs[4] reduced: 10 (and 7, 13) views -- the view sum runs in torch's
    vectorised order from 8 views on."""
    _check_root(dict(cfg='panoptic', B=1, cube=[32, 32, 32], hm='noise', seed=102 + views,
                     num_views=views), gpu_device)


@pytest.mark.parametrize('joints', [1, 4, 16, 19, 24, 27, 32])
def test_channel_counts(joints, gpu_device):
    """every padded channel width the kernels are instantiated for (16..32)."""
    _check_root(dict(cfg='panoptic', B=2, cube=[12, 10, 7], hm='noise', seed=200 + joints,
                     num_joints=joints), gpu_device)


def test_window_full_64_cubed(gpu_device):
    """configs[2] geometry on one frame pair: 253x253x64 fine lattice, 64^3 windows."""
    import tempo_amd
    spec = dict(cfg='panoptic', B=2, sub_cube=[64, 64, 64], hm='noise', seed=301,
                invalid_every=0)
    case = cases.build_window(None, full=spec)
    ref_layer = O.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    ref, ref_off = ref_layer(*case['args'])
    layer = tempo_amd.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    dcase = cases.to_device(case, gpu_device)
    with torch.no_grad():
        got, off = layer(*dcase['args'])
    err = _maxabs(got.cpu(), ref)
    print(f'\n[window 64^3] max-abs {err:.3e}, differing {int((got.cpu() != ref).sum())}')
    assert err <= TOL
    assert torch.equal(off.cpu(), ref_off)
    # fused axis-max planes (64-voxel columns: the in-wave reduction path), volume skipped
    with torch.no_grad():
        pxy, pxz, pyz, _, none = layer.get_planes(dcase['args'][0], dcase['args'][1],
                                                  *dcase['args'][6:])
    assert none is None
    for p, dim in ((pxy, 4), (pxz, 3), (pyz, 2)):
        assert torch.equal(torch.nan_to_num(p, nan=-1.0), torch.nan_to_num(got.max(dim=dim)[0], nan=-1.0))
        assert _maxabs(torch.nan_to_num(p.cpu(), nan=-1.0), torch.nan_to_num(ref.max(dim=dim)[0], nan=-1.0)) <= TOL


def test_h36m_window_config4_reduced(gpu_device):
    """configs[3] per GPU slice, reduced: H36M 4 cameras, 17 joints, 64^3 window."""
    import tempo_amd
    spec = dict(cfg='h36m', B=1, sub_cube=[64, 64, 64], hm='noise', seed=302, invalid_every=0,
                hm_size=[96, 96], img_size=[96, 96])
    case = cases.build_window(None, full=spec)
    ref_layer = O.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    ref, _ = ref_layer(*case['args'])
    layer = tempo_amd.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    with torch.no_grad():
        got, _ = layer(*cases.to_device(case, gpu_device)['args'])
    assert _maxabs(got.cpu(), ref) <= TOL


def test_config2_batch32_properties(gpu_device):
    """BASELINE configs[1] (the bench workload) at full size: properties that need
    no oracle run, plus the oracle on two of the 32 frames."""
    spec = dict(cfg='panoptic', B=32, cube=[80, 80, 20], hm='noise', seed=103)
    case, dcase, layer = _root_pair(spec, gpu_device)
    fms, size, center, cube, centers, scales, cameras = dcase['args']
    layer.materialize_grids = False
    with torch.no_grad():
        full, _ = layer(*dcase['args'])
        mask = layer.last_mask.clone()
        # batch independence: a frame alone gives the frame's slice of the batch
        for b in (0, 17, 31):
            one, _ = layer([f[b:b + 1] for f in fms], size, center, cube, centers[b:b + 1],
                           scales[b:b + 1], cameras[b:b + 1])
            assert _same(one[0], full[b])
        # scaling by a power of two is exact (noise maps are in [0,1): no clamp)
        half, _ = layer([f * 0.5 for f in fms], size, center, cube, centers, scales, cameras)
        assert _same(half, full * 0.5)
        # joints are independent channels
        perm = torch.randperm(fms[0].shape[1], generator=torch.Generator().manual_seed(1))
        pj, _ = layer([f[:, perm.to(gpu_device)] for f in fms], size, center, cube, centers,
                      scales, cameras)
        assert _same(pj, full[:, perm.to(gpu_device)])
    # a voxel no view sees is exactly 0; the rest lies in [0, 1]
    seen = (mask.to(torch.int32) != 0).view(32, 1, *cube)
    fin = torch.nan_to_num(full, nan=0.5)
    assert float(fin.min()) >= 0.0 and float(fin.max()) <= 1.0
    assert int(full.isnan().sum()) <= 15 * 32 * 4  # a handful of principal-plane voxels at most
    assert bool((full[(~seen).expand_as(full)] == 0).all())
    assert bool(seen.any()) and bool((~seen).any())
    # oracle on the first and the last frame
    cfm = case['args'][0]
    for b in (0, 31):
        ref_layer = O.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
        ref, _ = ref_layer([f[b:b + 1] for f in cfm], size, center, cube, centers[b:b + 1],
                           scales[b:b + 1], cameras[b:b + 1])
        assert _maxabs(full[b].cpu(), ref[0]) <= TOL
        assert torch.equal(mask[b].cpu().to(torch.int32).view(-1),
                           _bits(ref_layer.last_bounding).view(-1))


def test_zslabs_reassemble(gpu_device):
    """z-slab calls (the multi-GPU dense path, here all on one device) concatenate
    to the unsharded volume bit for bit."""
    spec = dict(cfg='panoptic', B=2, cube=[24, 20, 16], hm='noise', seed=104)
    case, dcase, layer = _root_pair(spec, gpu_device, keep_mask=False)
    with torch.no_grad():
        full, _ = layer(*dcase['args'])
        for world in (2, 4):
            step = 16 // world
            slabs = [layer.get_voxel(*dcase['args'], zslab=(r * step, (r + 1) * step))[0]
                     for r in range(world)]
            assert _same(torch.cat(slabs, dim=-1), full)


def test_config5_dense_128_cubed_10_views(gpu_device):
    """BASELINE configs[4] at full size on one device: 10 views, 15 joints, 128^3 grid;
    the oracle on the whole grid, and the 8 z-slabs of the 8-GPU run reassembled."""
    spec = dict(cfg='panoptic', B=1, cube=[128, 128, 128], hm='noise', seed=105, num_views=10)
    case, dcase, layer = _root_pair(spec, gpu_device)
    ref_layer = O.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    ref, _ = ref_layer(*case['args'])
    layer.materialize_grids = False
    with torch.no_grad():
        full, _ = layer(*dcase['args'])
        assert _maxabs(full.cpu(), ref) <= TOL
        assert torch.equal(layer.last_mask.cpu().to(torch.int32).view(-1),
                           _bits(ref_layer.last_bounding).view(-1))
        slabs = [layer.get_voxel(*dcase['args'], zslab=(16 * r, 16 * r + 16))[0] for r in range(8)]
    assert _same(torch.cat(slabs, dim=-1), full)


def test_config4_h36m_384_maps(gpu_device):
    """BASELINE configs[3] geometry at the reference's map size: H36M 4 cameras, 17 joints,
    384x384 heatmaps and image, 64^3 windows in the 8000x8000x2000 space."""
    import tempo_amd
    spec = dict(cfg='h36m', B=2, sub_cube=[64, 64, 64], hm='noise', seed=303, invalid_every=0,
                hm_size=[384, 384], img_size=[384, 384])
    case = cases.build_window(None, full=spec)
    ref_layer = O.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    ref, ref_off = ref_layer(*case['args'])
    layer = tempo_amd.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    with torch.no_grad():
        got, off = layer(*cases.to_device(case, gpu_device)['args'])
    assert _maxabs(got.cpu(), ref) <= TOL
    assert torch.equal(off.cpu(), ref_off)
