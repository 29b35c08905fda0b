"""GPU parity of the backward pass: gradient of a seeded linear functional of the
volume with respect to the feature maps, HIP (through the C ABI, via autograd)
vs torch autograd through the CPU oracle.

Bar: the forward's op order is reproduced bit for bit, but the backward adds
each texel's contributions with fp32 atomics, whose order is not fixed (and
torch's own CPU grid_sample backward accumulates sequentially), so the
comparison is to rounding: |diff| <= 2e-5 * max(1, max|ref|), stated here."""
import pytest
import torch

import cases
from oracle import project_layer_ref as O

pytestmark = pytest.mark.gpu
RTOL = 2e-5


def _grads_close(got, ref, name):
    scale = max(1.0, float(max(r.abs().max() for r in ref)))
    worst = max(float((g.cpu() - r).abs().max()) for g, r in zip(got, ref))
    nz = sum(int((r != 0).sum()) for r in ref)
    print(f'\n[{name}] grad max-abs diff {worst:.3e} (scale {scale:.3e}, {nz} non-zero entries)')
    assert nz > 0
    assert worst <= RTOL * scale
    # support must agree: texels no sample touches get exactly zero
    for g, r in zip(got, ref):
        far = (r == 0) & (g.cpu() != 0)
        assert int(far.sum()) == 0


def _leafs(fms, device=None):
    return [f.detach().clone().to(device or f.device).requires_grad_(True) for f in fms]


@pytest.mark.parametrize('name', list(cases.ROOT_CASES))
def test_root_backward_vs_autograd(name, gpu_device):
    import tempo_amd
    case = cases.build_root(name)
    args = list(case['args'])
    ref_fms = _leafs(args[0])
    ref_layer = O.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    ref_cubes, _ = ref_layer(ref_fms, *args[1:])
    g = torch.Generator().manual_seed(7)
    weight = torch.randn(ref_cubes.shape, generator=g)
    (ref_cubes * weight).sum().backward()
    ref_grads = [f.grad for f in ref_fms]

    layer = tempo_amd.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    layer.keep_mask = True
    fms = _leafs(args[0], gpu_device)
    cubes, grids = layer(fms, *args[1:])
    assert cubes.requires_grad and not grids.requires_grad
    assert float((cubes.detach().cpu() - ref_cubes.detach()).abs().max()) <= 1e-5
    (cubes * weight.to(gpu_device)).sum().backward()
    _grads_close([f.grad for f in fms], ref_grads, name)
    # the training forward equals the inference forward bit for bit
    with torch.no_grad():
        plain, _ = layer([f.detach() for f in fms], *args[1:])
    assert torch.equal(torch.nan_to_num(plain, nan=-1.0), torch.nan_to_num(cubes.detach(), nan=-1.0))


@pytest.mark.parametrize('name', list(cases.WINDOW_CASES))
def test_window_backward_vs_autograd(name, gpu_device):
    import tempo_amd
    case = cases.build_window(name)
    args = list(case['args'])
    ref_fms = _leafs(args[0])
    ref_layer = O.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    ref_cubes, _ = ref_layer(ref_fms, *args[1:])
    g = torch.Generator().manual_seed(8)
    weight = torch.randn(ref_cubes.shape, generator=g)
    (ref_cubes * weight).sum().backward()
    ref_grads = [f.grad for f in ref_fms]

    layer = tempo_amd.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    fms = _leafs(args[0], gpu_device)
    cubes, _ = layer(fms, args[1].to(gpu_device), *args[2:])
    (cubes * weight.to(gpu_device)).sum().backward()
    _grads_close([f.grad for f in fms], ref_grads, name)


@pytest.mark.parametrize('tile', [0, 121, 144])
@pytest.mark.parametrize('sub_mm,cube', [(400, 16), (560, 14), (800, 16)])
def test_window_backward_fine_lattices(sub_mm, cube, tile, gpu_device):
    """The window backward sums each 4x4x4 brick's footprints in an LDS tile before
    touching memory when the brick's footprint box fits the tile, retries the brick
    as two halves when it does not, and falls back to per-sample atomics last.
    Lattices of 25, 40 and 50 mm (the small cases above are 62-100 mm) reach all
    three; values outside [0,1] put both sides of the clamp mask into it, and a
    14^3 window leaves partial bricks on every axis."""
    import tempo_amd
    spec = dict(cases.WINDOW_CASES['win_panoptic_small'], sub=[sub_mm] * 3,
                sub_cube=[cube] * 3, seed=300 + cube + sub_mm, hm='noise')
    case = cases.build_window(None, full=spec)
    args = list(case['args'])
    base = [f * 3.0 - 1.0 for f in args[0]]
    ref_fms = _leafs(base)
    ref_layer = O.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    ref_cubes, _ = ref_layer(ref_fms, *args[1:])
    g = torch.Generator().manual_seed(9)
    weight = torch.randn(ref_cubes.shape, generator=g)
    (ref_cubes * weight).sum().backward()
    layer = tempo_amd.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    layer.backward_tile = tile  # 0: chosen from the lattice's pitch; both sizes must agree
    fms = _leafs(base, gpu_device)
    cubes, _ = layer(fms, args[1].to(gpu_device), *args[2:])
    assert float((cubes.detach().cpu() - ref_cubes.detach()).abs().max()) <= 1e-5
    (cubes * weight.to(gpu_device)).sum().backward()
    _grads_close([f.grad for f in fms], [f.grad for f in ref_fms], f'window {sub_mm} mm / {cube}^3')


@pytest.mark.parametrize('joints', [22, 27])
def test_window_backward_other_channel_widths(joints, gpu_device):
    """24 and 28 padded channels: two footprint texels per accumulate step instead of four
    (15, 17 and 32 channels are covered by the cases above)."""
    import tempo_amd
    spec = dict(cases.WINDOW_CASES['win_panoptic_small'], B=2, num_joints=joints,
                sub=[500] * 3, seed=500 + joints, hm='noise')
    case = cases.build_window(None, full=spec)
    args = list(case['args'])
    ref_fms = _leafs(args[0])
    ref_layer = O.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    ref_cubes, _ = ref_layer(ref_fms, *args[1:])
    g = torch.Generator().manual_seed(11)
    weight = torch.randn(ref_cubes.shape, generator=g)
    (ref_cubes * weight).sum().backward()
    layer = tempo_amd.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    fms = _leafs(args[0], gpu_device)
    cubes, _ = layer(fms, args[1].to(gpu_device), *args[2:])
    (cubes * weight.to(gpu_device)).sum().backward()
    _grads_close([f.grad for f in fms], [f.grad for f in ref_fms], f'window, {joints} channels')


def test_window_backward_on_maps_wider_than_the_tile_coordinates(gpu_device):
    """The LDS-tile kernel keeps footprint boxes in 16-bit texel coordinates; maps wider or
    taller than 16384 texels take the per-sample kernel instead. Same bar."""
    import tempo_amd
    spec = dict(cases.WINDOW_CASES['win_panoptic_small'], B=2, hm_size=[16400, 8],
                img_size=[16400, 8], seed=411, hm='noise')
    case = cases.build_window(None, full=spec)
    args = list(case['args'])
    ref_fms = _leafs(args[0])
    ref_layer = O.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    ref_cubes, _ = ref_layer(ref_fms, *args[1:])
    g = torch.Generator().manual_seed(10)
    weight = torch.randn(ref_cubes.shape, generator=g)
    (ref_cubes * weight).sum().backward()
    layer = tempo_amd.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    fms = _leafs(args[0], gpu_device)
    cubes, _ = layer(fms, args[1].to(gpu_device), *args[2:])
    assert float((cubes.detach().cpu() - ref_cubes.detach()).abs().max()) <= 1e-5
    (cubes * weight.to(gpu_device)).sum().backward()
    _grads_close([f.grad for f in fms], [f.grad for f in ref_fms], 'window, 16400-texel rows')


def test_backward_with_negative_and_saturated_values(gpu_device):
    """Feature maps outside [0,1] exercise both sides of the clamp mask."""
    import tempo_amd
    case = cases.build_root('root_panoptic_noise')
    args = list(case['args'])
    base = [f * 3.0 - 1.0 for f in args[0]]  # values in [-1, 2)
    ref_fms = _leafs(base)
    ref_layer = O.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    ref_cubes, _ = ref_layer(ref_fms, *args[1:])
    frac_clamped = float(((ref_cubes == 0) | (ref_cubes == 1)).float().mean())
    assert 0.05 < frac_clamped < 0.95
    ref_cubes.sum().backward()
    layer = tempo_amd.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    fms = _leafs(base, gpu_device)
    cubes, _ = layer(fms, *args[1:])
    cubes.sum().backward()
    _grads_close([f.grad for f in fms], [f.grad for f in ref_fms], 'clamp both sides')


def test_unrepack_inverts_repack(gpu_device):
    from tempo_amd import ops
    torch.manual_seed(9)
    # HW % 4 != 0 takes the 64-pixel kernel, the others the 256-pixel one (whole tiles, a
    # partial last tile, fewer pixels than one tile; 16 to 32 padded channels)
    for (B, V, C, H, W) in ((2, 3, 15, 7, 37), (1, 2, 17, 16, 64), (2, 2, 32, 5, 13),
                            (2, 2, 15, 6, 50), (1, 3, 20, 128, 240), (1, 2, 29, 12, 20),
                            (3, 1, 24, 3, 4)):
        fms = [torch.rand(B, C, H, W, device=gpu_device) for _ in range(V)]
        back = ops.unrepack_nchw(ops.repack_nhwc(fms), C)
        for a, b in zip(fms, back):
            assert torch.equal(a, b)


def test_backward_full_root_grid_config1(gpu_device):
    """BASELINE configs[0] size: one Panoptic frame, 80x80x20 grid, every texel's gradient."""
    import tempo_amd
    spec = dict(cfg='panoptic', B=1, cube=[80, 80, 20], hm='noise', seed=401)
    case = cases.build_root(None, full=spec)
    args = list(case['args'])
    ref_fms = _leafs(args[0])
    ref_layer = O.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    ref_cubes, _ = ref_layer(ref_fms, *args[1:])
    g = torch.Generator().manual_seed(9)
    weight = torch.randn(ref_cubes.shape, generator=g)
    (ref_cubes * weight).sum().backward()
    layer = tempo_amd.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    fms = _leafs(args[0], gpu_device)
    cubes, _ = layer(fms, *args[1:])
    (cubes * weight.to(gpu_device)).sum().backward()
    _grads_close([f.grad for f in fms], [f.grad for f in ref_fms], 'config 1 backward')


def test_backward_is_linear_and_matches_directional_derivative(gpu_device):
    """Size-independent properties at the batch-8 root size: the backward is linear in the
    incoming gradient, and <grad, d> equals the change of the (clamp-free) volume along d."""
    import tempo_amd
    spec = dict(cfg='panoptic', B=8, cube=[80, 80, 20], hm='noise', seed=402)
    case = cases.to_device(cases.build_root(None, full=spec), gpu_device)
    args = list(case['args'])
    layer = tempo_amd.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    layer.materialize_grids = False
    base = [f * 0.5 + 0.25 for f in args[0]]  # volume stays inside (0, 1): no clamping
    fms = _leafs(base)
    cubes, _ = layer(fms, *args[1:])
    g = torch.Generator(device=gpu_device).manual_seed(10)
    w1 = torch.randn(cubes.shape, device=gpu_device, generator=g)
    w2 = torch.randn(cubes.shape, device=gpu_device, generator=g)
    g1 = torch.autograd.grad((cubes * w1).sum(), fms, retain_graph=True)
    g2 = torch.autograd.grad((cubes * w2).sum(), fms, retain_graph=True)
    g12 = torch.autograd.grad((cubes * (w1 + 2 * w2)).sum(), fms)
    for a, b, c in zip(g1, g2, g12):
        assert float((a + 2 * b - c).abs().max()) <= 1e-4 * max(1.0, float(c.abs().max()))
    # the volume is linear in the maps where nothing clamps: f(x + d) - f(x) = J d exactly
    d = [torch.randn(f.shape, device=gpu_device, generator=g) * 0.01 for f in fms]
    with torch.no_grad():
        plus, _ = layer([f + e for f, e in zip(base, d)], *args[1:])
        here, _ = layer(base, *args[1:])
    finite = torch.isfinite(plus) & torch.isfinite(here)
    lhs = float(((plus - here) * w1)[finite].sum())
    rhs = float(sum((a * e).sum() for a, e in zip(g1, d)))
    assert abs(lhs - rhs) <= 2e-3 * max(1.0, abs(rhs))


def test_more_than_32_channels_goes_through_slices(gpu_device):
    """40 channels = two kernel calls (32 + 8) on every path: forward, z-max, planes,
    backward -- against the oracle and autograd through it."""
    import tempo_amd
    # root
    spec = dict(cases.ROOT_CASES['root_panoptic_small'], num_joints=40, seed=77)
    case = cases.build_root(None, full=spec)
    args = list(case['args'])
    ref_fms = _leafs(args[0])
    ref_layer = O.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    ref_cubes, _ = ref_layer(ref_fms, *args[1:])
    g = torch.Generator().manual_seed(11)
    weight = torch.randn(ref_cubes.shape, generator=g)
    (ref_cubes * weight).sum().backward()
    layer = tempo_amd.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    layer.keep_zmax = True
    fms = _leafs(args[0], gpu_device)
    with torch.no_grad():
        plain, _ = layer([f.detach() for f in fms], *args[1:])
    assert float((plain.cpu() - ref_cubes.detach()).abs().max()) <= 1e-5
    assert torch.equal(layer.last_zmax, plain.max(dim=4)[0])
    cubes, _ = layer(fms, *args[1:])
    assert torch.equal(cubes.detach(), plain)
    (cubes * weight.to(gpu_device)).sum().backward()
    _grads_close([f.grad for f in fms], [f.grad for f in ref_fms], 'root 40 channels')
    # window
    spec = dict(cases.WINDOW_CASES['win_panoptic_small'], num_joints=40, seed=78)
    case = cases.build_window(None, full=spec)
    args = list(case['args'])
    ref_fms = _leafs(args[0])
    ref_layer = O.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    ref_cubes, _ = ref_layer(ref_fms, *args[1:])
    weight = torch.randn(ref_cubes.shape, generator=g)
    (ref_cubes * weight).sum().backward()
    layer = tempo_amd.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    fms = _leafs(args[0], gpu_device)
    gc = args[1].to(gpu_device)
    with torch.no_grad():
        plain, _ = layer([f.detach() for f in fms], gc, *args[2:])
        pxy, pxz, pyz, _, vol = layer.get_planes([f.detach() for f in fms], gc, *args[6:],
                                                 want_cubes=True)
    assert float((plain.cpu() - ref_cubes.detach()).abs().max()) <= 1e-5
    assert torch.equal(vol, plain)
    for p, dim in ((pxy, 4), (pxz, 3), (pyz, 2)):
        assert torch.equal(p, plain.max(dim=dim)[0])
    cubes, _ = layer(fms, gc, *args[2:])
    (cubes * weight.to(gpu_device)).sum().backward()
    _grads_close([f.grad for f in fms], [f.grad for f in ref_fms], 'window 40 channels')


@pytest.mark.parametrize('views,joints', [(12, 15), (13, 15), (8, 30), (16, 17)])
def test_root_backward_with_more_views_than_one_launch_holds(views, joints, gpu_device):
    """The root backward keeps per-view footprints in LDS (11 views at CP=16, 7 at CP=32):
    more views go through in several launches.  Against autograd through the oracle."""
    import tempo_amd
    spec = dict(cfg='panoptic', B=1, cube=[10, 9, 7], hm='blobs', seed=61, num_views=views,
                num_joints=joints)
    case = cases.build_root(None, full=spec)
    args = list(case['args'])
    ref_fms = _leafs(args[0])
    ref_cubes, _ = O.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')(
        ref_fms, *args[1:])
    g = torch.Generator().manual_seed(9)
    weight = torch.randn(ref_cubes.shape, generator=g)
    (ref_cubes * weight).sum().backward()
    layer = tempo_amd.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    fms = _leafs(args[0], gpu_device)
    cubes, _ = layer(fms, *args[1:])
    # V >= 16: torch's view-sum order is reproduced to rounding only (unproject_math.h)
    assert float((cubes.detach().cpu() - ref_cubes.detach()).abs().max()) <= (
        1e-5 if views < 16 else 1e-4)
    (cubes * weight.to(gpu_device)).sum().backward()
    _grads_close([f.grad for f in fms], [f.grad for f in ref_fms], f'V={views} J={joints}')


def test_window_backward_many_views(gpu_device):
    """The default window backward keeps no per-view records: any view count."""
    import tempo_amd
    spec = dict(cfg='panoptic', B=2, sub_cube=[12, 12, 12], hm='blobs', seed=62,
                space=[2000, 2000, 2000], sub=[1000, 1000, 1000], num_views=13,
                invalid_every=0)
    case = cases.build_window(None, full=spec)
    args = list(case['args'])
    ref_fms = _leafs(args[0])
    ref_cubes, _ = O.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')(
        ref_fms, *args[1:])
    g = torch.Generator().manual_seed(10)
    weight = torch.randn(ref_cubes.shape, generator=g)
    (ref_cubes * weight).sum().backward()
    layer = tempo_amd.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    fms = _leafs(args[0], gpu_device)
    cubes, _ = layer(fms, args[1].to(gpu_device), *args[2:])
    (cubes * weight.to(gpu_device)).sum().backward()
    _grads_close([f.grad for f in fms], [f.grad for f in ref_fms], 'window V=13')


def test_backward_reports_nondeterminism_like_grid_sample(gpu_device):
    """torch.use_deterministic_algorithms(True): the op this backward replaces
    (grid_sampler_2d_backward_cuda) raises, or warns with warn_only; so does the HIP one."""
    import warnings
    import tempo_amd
    spec = dict(cfg='panoptic', B=1, cube=[8, 8, 4], hm='noise', seed=901)
    case = cases.build_root(None, full=spec)
    args = list(case['args'])
    layer = tempo_amd.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')

    def loss():
        fms = _leafs(args[0], gpu_device)
        return layer(fms, *args[1:])[0].sum(), fms

    try:
        torch.use_deterministic_algorithms(True)
        out, _ = loss()  # the forward is deterministic
        with pytest.raises(RuntimeError, match='deterministic'):
            out.backward()
        torch.use_deterministic_algorithms(True, warn_only=True)
        out, fms = loss()
        with warnings.catch_warnings(record=True) as seen:
            warnings.simplefilter('always')
            out.backward()
        assert any('deterministic' in str(w.message) for w in seen)
        assert fms[0].grad is not None and float(fms[0].grad.abs().max()) > 0
    finally:
        torch.use_deterministic_algorithms(False)
    out, fms = loss()
    out.backward()
    assert fms[0].grad is not None
