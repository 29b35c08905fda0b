"""Half-precision feature maps.  The reference decorates both ``get_voxel`` methods with
``@force_fp32(apply_to=('feature_maps', 'centers'))`` (project_layer.py:65, :242): under
mixed-precision training the 2-D head hands over fp16 / bf16 maps and the layer samples their
fp32 conversion.  The drop-in does the same: the result must equal the module fed the
converted maps bit for bit in the forward, and to one rounding step of the maps' dtype in the
backward (its atomics have no fixed order)."""
import pytest
import torch

import cases

pytestmark = pytest.mark.gpu


def _same(a, b):
    return torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0))


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('memory', ['nchw', 'channels_last'])
def test_root_half_maps_are_sampled_as_fp32(dtype, memory, gpu_device):
    import tempo_amd
    spec = dict(cfg='panoptic', B=3, cube=[20, 18, 12], hm='noise', seed=311)
    case = cases.to_device(cases.build_root(None, full=spec), gpu_device)
    args = list(case['args'])
    half = [m.to(dtype) for m in args[0]]
    if memory == 'channels_last':
        half = [m.contiguous(memory_format=torch.channels_last) for m in half]
    layer = tempo_amd.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    with torch.no_grad():
        got, _ = layer(half, *args[1:])
        want, _ = layer([m.float() for m in half], *args[1:])
    assert got.dtype == torch.float32 and _same(got, want)
    assert float(want.max()) > 0


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_window_half_maps_forward_and_backward(dtype, gpu_device):
    import tempo_amd
    spec = dict(cfg='panoptic', B=2, sub_cube=[12, 12, 12], hm='noise', seed=312, invalid_every=0)
    case = cases.to_device(cases.build_window(None, full=spec), gpu_device)
    args = list(case['args'])
    layer = tempo_amd.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    half = [m.to(dtype).requires_grad_(True) for m in args[0]]
    full = [m.detach().float().requires_grad_(True) for m in half]
    got, off_h = layer(half, *args[1:])
    want, off_f = layer(full, *args[1:])
    assert got.dtype == torch.float32 and _same(got, want) and torch.equal(off_h, off_f)
    g = torch.Generator().manual_seed(5)
    w = torch.rand(got.shape, generator=g).to(gpu_device)
    (got * w).sum().backward()
    (want * w).sum().backward()
    for h, f in zip(half, full):
        # autograd casts the fp32 gradient of `.float()` back to the map's dtype
        assert h.grad is not None and h.grad.dtype == dtype
        # (two backward runs: the fp32 atomics' order is not fixed, so equal up to the last
        # fp32 bits before the cast -- one rounding step of the 16-bit type at most)
        rtol = 1e-3 if dtype == torch.float16 else 8e-3
        assert torch.allclose(h.grad.float(), f.grad, rtol=rtol, atol=1e-6)
    assert any(float(f.grad.abs().max()) > 0 for f in full)


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('shape,c_begin,channels', [
    ((3, 15, 12, 20), 0, None),    # wide tile (HW % 4 == 0, aligned planes)
    ((2, 40, 16, 24), 32, 8),      # channel slice of a wider tensor, wide tile
    ((2, 17, 9, 7), 0, None),      # HW = 63: the generic tile
    ((2, 21, 11, 13), 4, 17),      # generic tile, slice, CP = 20
    ((1, 32, 8, 520), 0, None),    # more than one 256-pixel tile per row of tiles, CP = 32
])
def test_half_repack_equals_the_repack_of_the_converted_maps(dtype, shape, c_begin, channels,
                                                             gpu_device):
    """tempo_repack_nhwc_half widens inside the repack: texels bit-equal to tempo_repack_nhwc
    on ``.float()`` of the same maps, padding channels zero; special values included."""
    from tempo_amd import ops
    g = torch.Generator().manual_seed(sum(shape) + c_begin)
    maps = []
    for v in range(3):
        m = (torch.rand(*shape, generator=g) * 4 - 2).to(dtype)
        m.view(-1)[:6] = torch.tensor([0.0, -0.0, float('inf'), float('-inf'), float('nan'),
                                       6e-8], dtype=dtype)  # (6e-8: a half denormal)
        maps.append(m.to(gpu_device))
    got = ops.repack_nhwc(maps, c_begin, channels)
    want = ops.repack_nhwc([m.float() for m in maps], c_begin, channels)
    assert got.dtype == torch.float32 and got.shape == want.shape
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))
    C = shape[1] - c_begin if channels is None else channels
    assert not got[..., C:].any()
    # an odd view pointer offset (2-byte aligned only) goes through the generic tile
    flat = torch.zeros(maps[0].numel() + 1, dtype=dtype, device=gpu_device)
    shifted = []
    for m in maps:
        buf = flat.clone()
        buf[1:] = m.reshape(-1)
        shifted.append(buf[1:].view(shape))
    assert torch.equal(ops.repack_nhwc(shifted, c_begin, channels).view(torch.int32),
                       want.view(torch.int32))
