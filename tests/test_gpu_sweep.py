"""Seeded random sweep of call shapes, HIP modules vs the CPU oracle: view counts 1..8, channel
counts 1..40 (above 32 the module slices), odd map sizes, grids smaller than one wave and with
awkward z extents, shared and per-item grid centres with skipped items, blob and noise maps,
NCHW / channels_last / packed inputs.  Every case is small enough for the oracle to finish in
well under a second.  Bars as everywhere: volume <= 1e-5 max-abs, NaN positions equal, mask /
grids / offset bit-exact."""
import numpy as np
import pytest
import torch

import cases
from oracle import project_layer_ref as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _bits(bounding):
    V = bounding.shape[-1]
    return (bounding.to(torch.int32) << torch.arange(V, dtype=torch.int32)).sum(-1)


def _close(got, ref):
    assert got.shape == ref.shape
    assert torch.equal(got.isnan(), ref.isnan())
    return float((torch.nan_to_num(got) - torch.nan_to_num(ref)).abs().max())


def _sizes(rng):
    """(heatmap_size, image_size): the 2-D network's stride is 4 or 8, sizes need not be round"""
    stride = int(rng.choice([4, 8]))
    w, h = int(rng.randint(9, 70)), int(rng.randint(9, 50))
    return [w, h], [w * stride, h * stride]


def _root_spec(i):
    rng = np.random.RandomState(9000 + i)
    hm, img = _sizes(rng)
    spec = dict(cfg='panoptic', B=int(rng.randint(1, 5)),
                cube=[int(rng.randint(1, 22)), int(rng.randint(1, 22)), int(rng.randint(1, 27))],
                hm=str(rng.choice(['blobs', 'noise'])), seed=9000 + i, hm_size=hm, img_size=img,
                num_views=int(rng.randint(1, 9)),
                num_joints=int(rng.choice([1, 3, 15, 16, 17, 20, 24, 31, 32, 33, 40])))
    if rng.rand() < 0.4 and spec['B'] >= 2:
        spec['centers'] = 'per_batch'
    return spec, str(rng.choice(['nchw', 'channels_last', 'packed']))


def _window_spec(i):
    rng = np.random.RandomState(9500 + i)
    hm, img = _sizes(rng)
    # (cubic windows: the reference stacks the three face grids of a window into
    # center_grid, project_layer.py:369-378, which needs equal face sizes)
    sub_cube = [int(rng.randint(2, 19))] * 3
    space = [float(rng.randint(1500, 2600)) for _ in range(3)]
    spec = dict(cfg='panoptic', B=int(rng.randint(1, 6)), sub_cube=sub_cube,
                hm=str(rng.choice(['blobs', 'noise'])), seed=9500 + i, hm_size=hm, img_size=img,
                space=space, sub=[s / 2 for s in space], num_views=int(rng.randint(1, 9)),
                num_joints=int(rng.choice([1, 4, 15, 17, 21, 32, 35])),
                invalid_every=int(rng.choice([0, 2, 3])))
    return spec, str(rng.choice(['nchw', 'channels_last']))


def _as_layout(maps, layout):
    import tempo_amd
    from tempo_amd import ops
    if layout == 'channels_last' and maps[0].shape[1] > 1:
        return [m.contiguous(memory_format=torch.channels_last) for m in maps]
    if layout == 'packed' and maps[0].shape[1] <= 32:
        return tempo_amd.PackedHeatmaps(ops.repack_nhwc(maps), maps[0].shape[1])
    return maps


@pytest.mark.parametrize('i', range(24))
def test_root_sweep(i, gpu_device):
    import tempo_amd
    spec, layout = _root_spec(i)
    case = cases.build_root(None, full=spec)
    ref_layer = O.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    ref_cubes, ref_grids = ref_layer(*case['args'])
    ref_mask = _bits(ref_layer.last_bounding)
    layer = tempo_amd.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    layer.keep_mask = True
    layer.keep_zmax = True
    args = list(cases.to_device(case, gpu_device)['args'])
    args[0] = _as_layout(args[0], layout)
    with torch.no_grad():
        cubes, grids = layer(*args)
    torch.cuda.synchronize()
    err = _close(cubes.cpu(), ref_cubes)
    print(f'\n[root sweep {i}] {spec} {layout}: max-abs {err:.3e}')
    assert err <= TOL
    assert torch.equal(grids.cpu(), ref_grids)
    assert torch.equal(layer.last_mask.cpu().to(torch.int32), ref_mask)
    if layer.last_zmax is not None:
        want = cubes.max(dim=4)[0]
        assert torch.equal(torch.nan_to_num(layer.last_zmax, nan=-7.0),
                           torch.nan_to_num(want, nan=-7.0))


@pytest.mark.parametrize('i', range(16))
def test_window_sweep(i, gpu_device):
    import tempo_amd
    spec, layout = _window_spec(i)
    case = cases.build_window(None, full=spec)
    ref_layer = O.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    ref_cubes, ref_off = ref_layer(*case['args'])
    layer = tempo_amd.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    args = list(cases.to_device(case, gpu_device)['args'])
    args[0] = _as_layout(args[0], layout)
    with torch.no_grad():
        cubes, off = layer(*args)
    torch.cuda.synchronize()
    err = _close(cubes.cpu(), ref_cubes)
    print(f'\n[window sweep {i}] {spec} {layout}: max-abs {err:.3e}')
    assert err <= TOL
    assert torch.equal(off.cpu(), ref_off)
    assert torch.equal(layer.center_grid.cpu(), ref_layer.center_grid)


RTOL_GRAD = 2e-5  # tests/test_gpu_backward.py: fp32 atomics, order not fixed


def _grad_check(got, ref, name):
    scale = max(1.0, float(max(r.abs().max() for r in ref)))
    worst = max(float((g.cpu() - r).abs().max()) for g, r in zip(got, ref))
    print(f'\n[{name}] grad max-abs diff {worst:.3e} (scale {scale:.3e})')
    assert worst <= RTOL_GRAD * scale
    for g, r in zip(got, ref):  # texels no sample touches get exactly zero
        assert int(((r == 0) & (g.cpu() != 0)).sum()) == 0


@pytest.mark.parametrize('i', range(0, 24, 3))
def test_root_backward_sweep(i, gpu_device):
    """the same shapes through autograd: HIP backward vs torch autograd through the oracle"""
    import tempo_amd
    spec, _ = _root_spec(i)
    case = cases.build_root(None, full=spec)
    args = list(case['args'])
    ref_fms = [f.clone().requires_grad_(True) for f in args[0]]
    ref_cubes, _ = O.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')(
        ref_fms, *args[1:])
    weight = torch.randn(ref_cubes.shape, generator=torch.Generator().manual_seed(70 + i))
    (torch.nan_to_num(ref_cubes) * weight).sum().backward()
    layer = tempo_amd.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    fms = [f.clone().to(gpu_device).requires_grad_(True) for f in args[0]]
    cubes, _ = layer(fms, *args[1:])
    assert _close(cubes.detach().cpu(), ref_cubes.detach()) <= TOL
    (torch.nan_to_num(cubes) * weight.to(gpu_device)).sum().backward()
    _grad_check([f.grad for f in fms], [f.grad for f in ref_fms], f'root backward sweep {i}')


@pytest.mark.parametrize('i', range(0, 16, 2))
def test_window_backward_sweep(i, gpu_device):
    import tempo_amd
    spec, _ = _window_spec(i)
    case = cases.build_window(None, full=spec)
    args = list(case['args'])
    ref_fms = [f.clone().requires_grad_(True) for f in args[0]]
    ref_cubes, _ = O.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')(
        ref_fms, *args[1:])
    weight = torch.randn(ref_cubes.shape, generator=torch.Generator().manual_seed(80 + i))
    (torch.nan_to_num(ref_cubes) * weight).sum().backward()
    layer = tempo_amd.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    fms = [f.clone().to(gpu_device).requires_grad_(True) for f in args[0]]
    cubes, _ = layer(fms, args[1].to(gpu_device), *args[2:])
    assert _close(cubes.detach().cpu(), ref_cubes.detach()) <= TOL
    (torch.nan_to_num(cubes) * weight.to(gpu_device)).sum().backward()
    _grad_check([f.grad for f in fms], [f.grad for f in ref_fms], f'window backward sweep {i}')


def _eq(a, b):
    return torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0))


@pytest.mark.parametrize('i', range(24))
def test_root_consumers_sweep(i, gpu_device):
    """forward_zmax and zcolumns (no volume) equal the detector's expressions on the module's
    own volume bit for bit, over the random shapes and layouts (center_net.py:230,
    tempo_detector.py:204-206)."""
    import tempo_amd
    spec, layout = _root_spec(i)
    case = cases.to_device(cases.build_root(None, full=spec), gpu_device)
    args = list(case['args'])
    args[0] = _as_layout(args[0], layout)
    layer = tempo_amd.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    with torch.no_grad():
        cubes, _ = layer(*args)
        zmax = layer.forward_zmax(*args)
        B, C, X, Y, Z = cubes.shape
        K = min(7, X * Y)
        g = torch.Generator().manual_seed(400 + i)
        index = torch.stack([torch.randperm(X * Y, generator=g)[:K] for _ in range(B)])
        cols = layer.zcolumns(*args, index)
    assert _eq(zmax, cubes.max(dim=4)[0])
    idx = index.to(gpu_device).view(B, K, 1, 1).repeat(1, 1, C, Z)
    want = torch.gather(torch.flatten(cubes, 2, 3).permute(0, 2, 1, 3), dim=1, index=idx)
    assert _eq(cols, want)


@pytest.mark.parametrize('i', range(16))
def test_window_planes_sweep(i, gpu_device):
    """get_planes (volume optional) equals torch.max over each axis of the module's volume
    (tempo_pose.py:172-176), windows of every size: the generic epilogue, not only the fused
    64^3 kernel."""
    import tempo_amd
    spec, layout = _window_spec(i)
    case = cases.to_device(cases.build_window(None, full=spec), gpu_device)
    args = list(case['args'])
    args[0] = _as_layout(args[0], layout)
    layer = tempo_amd.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    with torch.no_grad():
        cubes, off = layer(*args)
        fms, grid_center = args[0], args[1]
        centers, scales, cameras = args[6:9]
        pxy, pxz, pyz, off2, none = layer.get_planes(fms, grid_center, centers, scales, cameras)
        qxy, qxz, qyz, _, vol = layer.get_planes(fms, grid_center, centers, scales, cameras,
                                                 want_cubes=True)
    assert none is None and torch.equal(off, off2) and _eq(vol, cubes)
    for p, q, dim in ((pxy, qxy, 4), (pxz, qxz, 3), (pyz, qyz, 2)):
        want = cubes.max(dim=dim)[0]
        assert _eq(p, want) and _eq(q, want)
