"""The library launches on torch's CURRENT stream (ops._stream_ptr): a call made inside
``torch.cuda.stream(side)`` must be ordered after the work already queued on ``side`` and must
not touch the default stream.  The inputs are produced by a long chain of kernels on the side
stream right before the call; a launch on the wrong stream would run ahead of that chain and
sample half-written maps."""
import pytest
import torch

import cases

pytestmark = pytest.mark.gpu


def _chain(x, n=60):
    for _ in range(n):  # a few ms of dependent elementwise kernels
        x = x * 1.0009765625 - x * 0.0009765625
    return x


@pytest.mark.parametrize('kind', ['root', 'window'])
def test_calls_follow_the_current_stream(kind, gpu_device):
    import tempo_amd
    if kind == 'root':
        spec = dict(cfg='panoptic', B=8, cube=[48, 48, 16], hm='noise', seed=611)
        case = cases.to_device(cases.build_root(None, full=spec), gpu_device)
        layer = tempo_amd.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    else:
        spec = dict(cfg='panoptic', B=6, sub_cube=[32, 32, 32], hm='noise', seed=612,
                    invalid_every=0)
        case = cases.to_device(cases.build_window(None, full=spec), gpu_device)
        layer = tempo_amd.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    args = list(case['args'])
    base = [m.clone() for m in args[0]]
    with torch.no_grad():
        want, _ = layer([_chain(m) for m in base], *args[1:])
    torch.cuda.synchronize()
    sides = [torch.cuda.Stream(gpu_device), torch.cuda.Stream(gpu_device)]
    outs = []
    with torch.no_grad():
        for rep in range(3):
            for s in sides:
                s.wait_stream(torch.cuda.current_stream(gpu_device))
                with torch.cuda.stream(s):
                    maps = [_chain(m) for m in base]
                    outs.append((s, layer(maps, *args[1:])[0]))
    for s, out in outs:
        s.synchronize()
        assert torch.equal(torch.nan_to_num(out, nan=-7.0), torch.nan_to_num(want, nan=-7.0))
