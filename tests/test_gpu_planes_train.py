"""Training THROUGH the fused window planes (tempo_unproject_window_planes +
tempo_window_planes_claims + tempo_unproject_window_planes_backward behind
``ProjectLayerWithMask.get_planes``): the gradient with respect to the heatmaps equals autograd
through the oracle's volume followed by the pose regressor's three ``torch.max``
(tempo_pose.py:172-176) -- first-index tie rule, clamp backward, masked mean -- and neither the
window volume nor its gradient volume is ever allocated.  Bar for gradients as in
tests/test_gpu_backward.py: |diff| <= 2e-5 * max(1, max|ref|) (fp32 atomics, order not fixed);
the claims are integers and must name the oracle's argmax exactly."""
import pytest
import torch

import cases
from oracle import project_layer_ref as O

pytestmark = pytest.mark.gpu
RTOL = 2e-5
DIMS = (4, 3, 2)  # torch.max over z, y, x -> the xy, xz, yz planes


def _oracle(case, maps, weights):
    leafs = [m.clone().requires_grad_(True) for m in maps]
    layer = O.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    cubes, _ = layer(leafs, *case['args'][1:])
    maxes = [cubes.max(dim=d) for d in DIMS]
    sum((torch.nan_to_num(m[0]) * w).sum() for m, w in zip(maxes, weights)).backward()
    return cubes.detach(), [m[0].detach() for m in maxes], [m[1] for m in maxes], \
        [f.grad for f in leafs]


def _run(case, maps, weights, dev, tile=0):
    import tempo_amd
    layer = tempo_amd.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    layer.backward_tile = tile  # 0: the module's own choice
    args = case['args']
    layer.set_space_size(*args[2:6], dev)
    leafs = [m.clone().to(dev).requires_grad_(True) for m in maps]
    props = args[1].to(dev)
    torch.cuda.reset_peak_memory_stats(dev)
    base = torch.cuda.memory_allocated(dev)
    pxy, pxz, pyz, _, none = layer.get_planes(leafs, props, *args[6:])
    assert none is None and pxy.requires_grad and pxz.requires_grad and pyz.requires_grad
    sum((torch.nan_to_num(p) * w.to(dev)).sum() for p, w in zip((pxy, pxz, pyz), weights)).backward()
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated(dev) - base
    with torch.no_grad():
        inf = layer.get_planes([m.to(dev) for m in maps], props, *args[6:])[:3]
    return [p.detach() for p in (pxy, pxz, pyz)], inf, [f.grad for f in leafs], peak, layer


def _check(name, case, maps, dev, peak_limit=None, tile=0):
    g = torch.Generator().manual_seed(17)
    B, C = maps[0].shape[0], maps[0].shape[1]
    NW = case['args'][1].shape[0]
    v = case['args'][5]
    weights = [torch.randn(NW, C, *sh, generator=g)
               for sh in ((v[0], v[1]), (v[0], v[2]), (v[1], v[2]))]
    cubes, ref_planes, ref_arg, ref_grads = _oracle(case, maps, weights)
    planes, inf, grads, peak, layer = _run(case, maps, weights, dev, tile)
    for p, q, r in zip(planes, inf, ref_planes):
        # the training forward is the inference kernel; both equal torch.max of the oracle
        assert torch.equal(torch.nan_to_num(p, nan=-7.0), torch.nan_to_num(q, nan=-7.0))
        assert torch.equal(p.isnan().cpu(), r.isnan())
        assert float((torch.nan_to_num(p).cpu() - torch.nan_to_num(r)).abs().max()) <= 1e-5
    scale = max(1.0, float(max(r.abs().max() for r in ref_grads)))
    worst = max(float((a.cpu() - b).abs().max()) for a, b in zip(grads, ref_grads))
    nz = sum(int((r != 0).sum()) for r in ref_grads)
    print(f'\n[{name}] grad max-abs diff {worst:.3e} (scale {scale:.3e}, {nz} non-zero), '
          f'peak extra memory {peak / 2**20:.1f} MiB')
    assert nz > 0 and worst <= RTOL * scale
    for a, b in zip(grads, ref_grads):
        assert int(((b == 0) & (a.cpu() != 0)).sum()) == 0
    if peak_limit is not None:
        assert peak <= peak_limit, 'something the size of the volume was allocated'
    return cubes, ref_arg, layer


@pytest.mark.parametrize('tile', [0, 81, 121, 144])
@pytest.mark.parametrize('name', list(cases.WINDOW_CASES))
def test_planes_gradient_named_cases(name, tile, gpu_device):
    """(every LDS tile size of the backward: boxes that fit, bricks done as halves, and the
    per-sample path for what still does not fit)"""
    case = cases.build_window(name)
    _check(f'{name} tile {tile}', case, case['args'][0], gpu_device, tile=tile)


@pytest.mark.parametrize('hm', ['blobs', 'noise'])
def test_planes_gradient_64_cubed_windows(hm, gpu_device):
    """BASELINE config 3's window: 64^3 (a wave = one z-column: the reduced xy claims), blob
    maps (nearly every line ties at 0 -> index 0, mostly masked or zero voxels) and noise."""
    spec = dict(cfg='panoptic', B=2, sub_cube=[64, 64, 64], hm=hm, seed=931, invalid_every=0)
    case = cases.build_window(None, full=spec)
    _check(f'64^3 {hm}', case, case['args'][0], gpu_device)


def test_planes_gradient_allocates_nothing_the_size_of_the_volume(gpu_device):
    """64^3 windows on small maps (60 x 32, so that everything map-sized is ~1 MB): the whole
    training step's extra memory stays far below ONE volume (31 MB here; the path through the
    volume holds it twice, value and gradient)."""
    spec = dict(cfg='panoptic', B=2, sub_cube=[64, 64, 64], hm='noise', seed=934,
                invalid_every=0, hm_size=[60, 32], img_size=[240, 128])
    case = cases.build_window(None, full=spec)
    C = case['args'][0][0].shape[1]
    volume = 2 * C * 64 ** 3 * 4
    _check('64^3 on small maps', case, case['args'][0], gpu_device, peak_limit=volume // 3)


def test_claims_name_the_oracles_argmax(gpu_device):
    """the claims decode to torch.max's indices (first index on ties) and to torch.clamp's
    pass rule, entry by entry"""
    from tempo_amd import ops
    import tempo_amd
    spec = dict(cfg='panoptic', B=2, sub_cube=[16, 16, 16], hm='blobs', seed=932,
                invalid_every=0, space=[2000, 2000, 2000], sub=[1000, 1000, 1000])
    case = cases.build_window(None, full=spec)
    maps = [m * 3.0 - 0.5 for m in case['args'][0]]  # values below 0 and above 1: clamp blocks
    layer = O.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    raw_layer = O.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    cubes, _ = layer(maps, *case['args'][1:])
    dev = gpu_device
    hip = tempo_amd.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    args = case['args']
    hip.set_space_size(*args[2:6], dev)
    dmaps = [m.to(dev) for m in maps]
    _, _, _, offset, live, cam, (tl, st, en) = hip._window_inputs(dmaps, args[1].to(dev), *args[6:])
    hm = ops.repack_nhwc(dmaps)
    C = maps[0].shape[1]
    out = ops.unproject_window_planes(hm, cam, hip._fine_axes, tl, st, en, hip._sub_cube, C,
                                      hip.heatmap_size, hip.image_size, live=live)
    claims = ops.window_planes_claims(hm, cam, hip._fine_axes, tl, st, en, hip._sub_cube, C,
                                      out[1:], hip.heatmap_size, hip.image_size, live=live)
    # pre-clamp values from the differentiable oracle: where does clamp pass the gradient?
    leaf = cubes.clone().requires_grad_(True)
    for claim, d in zip(claims, DIMS):
        val, idx = cubes.max(dim=d)
        got = claim.cpu()
        assert torch.equal(got >> 1, idx.to(torch.int32)), f'argmax along dim {d}'
        # a clamped maximum of exactly 0 or 1 may come from a blocked voxel; strictly inside
        # (0, 1) the clamp passed
        inside = (val > 0) & (val < 1)
        assert int(((got & 1) == 1)[inside].sum()) == 0
    assert any(int((c.cpu() & 1).sum()) > 0 for c in claims)  # some maxima ARE blocked


def test_planes_gradient_blocked_nan_skipped_and_sliced(gpu_device):
    """maps with values outside [0, 1] (the clamp blocks those maxima), a NaN patch, a skipped
    window, partial windows, and 35 channels (two channel slices)"""
    spec = dict(cfg='panoptic', B=4, sub_cube=[12, 12, 12], hm='noise', seed=933,
                invalid_every=3, num_joints=35, num_views=4)
    case = cases.build_window(None, full=spec)
    maps = [m * 2.5 - 0.75 for m in case['args'][0]]
    maps[1][0, 3, 10:14, 20:24] = float('nan')
    _check('blocked / NaN / skipped / 35 channels', case, maps, gpu_device)


def test_forward_sequence_planes_trains_like_the_callers_loops(gpu_device):
    """tempo_pose.py:145-176 + :266 under training: candidates x timesteps, planes only, ONE
    differentiable call -- planes and heatmap gradients equal the oracle's two loops of
    forward + torch.max under autograd (gradients of a frame's maps sum over its proposals;
    a skipped candidate contributes nothing)."""
    import tempo_amd
    from tempo_amd import synthetic as S
    cfg = S.PANOPTIC
    T, B, P = 2, 2, 3
    space, sub, sub_cube = [2000, 2000, 2000], [1000, 1000, 1000], [16, 16, 16]
    cams = [S.ring_cameras(5, seed=60 + b, target=cfg['space_center']) for b in range(B)]
    w, h = cfg['heatmap_size']
    maps_t = [S.make_heatmaps(B, 5, cfg['num_joints'], h, w, kind='noise', seed=70 + t)
              for t in range(T)]
    metas = [S.frame_meta(cams[b], 1, cfg['raw_size'], cfg['image_size']) for b in range(B)]
    centers = [m[0][0] for m in metas]
    scales = [m[1][0] for m in metas]
    cameras = [m[2][0] for m in metas]
    cand = S.make_proposals(B * P, cfg['space_center'], space, seed=61, invalid_every=4,
                            margin=0.0).view(B, P, 7)
    g = torch.Generator().manual_seed(23)
    C = cfg['num_joints']
    wts = [torch.randn(B, P, T, C, 16, 16, generator=g) for _ in range(3)]

    ref_leafs = [[m.clone().requires_grad_(True) for m in maps_t[t]] for t in range(T)]
    ref = O.ProjectLayerWithMask(cfg['image_size'], cfg['heatmap_size'], 'mean')
    loss = 0.0
    ref_planes = [[[None] * T for _ in range(P)] for _ in range(3)]
    for n in range(P):
        for t in range(T):
            cubes, _ = ref(ref_leafs[t], cand[:, n], cfg['space_center'], space, sub, sub_cube,
                           centers, scales, cameras)
            for k, d in enumerate(DIMS):
                pl = cubes.max(dim=d)[0]
                ref_planes[k][n][t] = pl.detach()
                loss = loss + (pl * wts[k][:, n, t]).sum()
    loss.backward()

    layer = tempo_amd.ProjectLayerWithMask(cfg['image_size'], cfg['heatmap_size'], 'mean')
    leafs = [[m.clone().to(gpu_device).requires_grad_(True) for m in maps_t[t]] for t in range(T)]
    pxy, pxz, pyz, off, none = layer.forward_sequence_planes(
        leafs, cand.to(gpu_device), cfg['space_center'], space, sub, sub_cube, [centers] * T,
        [scales] * T, [cameras] * T)
    assert none is None and pxy.requires_grad
    for k, p in enumerate((pxy, pxz, pyz)):
        for n in range(P):
            for t in range(T):
                assert float((p[:, n, t].detach().cpu() - ref_planes[k][n][t]).abs().max()) <= 1e-5
    sum((p * wt.to(gpu_device)).sum() for p, wt in zip((pxy, pxz, pyz), wts)).backward()
    for t in range(T):
        scale = max(1.0, float(max(f.grad.abs().max() for f in ref_leafs[t])))
        for a, b in zip(leafs[t], ref_leafs[t]):
            assert float((a.grad.cpu() - b.grad).abs().max()) <= RTOL * scale
            assert int(((b.grad == 0) & (a.grad.cpu() != 0)).sum()) == 0
