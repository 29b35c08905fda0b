"""The collective path on real hardware: a one-rank `nccl` (= RCCL) process group on the one
GPU a test box has, z-slab unprojection with the HIP layer + `all_gather_into_tensor` + the
device permute of tempo_amd.parallel, against the unsharded call.  (More ranks run on gloo
with the oracle as compute in tests/test_sharding.py; the reference's analogue is
tools/dist_train.sh:12-20.)"""
import socket

import pytest
import torch

import cases

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_zslab_allgather_over_rccl_one_rank(gpu_device):
    import torch.distributed as dist
    import tempo_amd
    from tempo_amd import parallel
    if dist.is_initialized():
        pytest.skip('a process group already exists in this process')
    torch.cuda.set_device(gpu_device)
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{_free_port()}', rank=0,
                            world_size=1, device_id=gpu_device)
    try:
        spec = dict(cfg='panoptic', B=2, cube=[24, 20, 16], hm='noise', seed=104)
        case = cases.to_device(cases.build_root(None, full=spec), gpu_device)
        layer = tempo_amd.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
        with torch.no_grad():
            full, _ = layer(*case['args'])
            got = parallel.unproject_root_zslab(layer, *case['args'])
            # the gather itself with more than one slab per rank: the slabs of a 4-way split,
            # gathered one "rank" at a time, reassemble the volume (permute path, R > 1)
            slabs = [layer.get_voxel(*case['args'], zslab=(4 * r, 4 * r + 4))[0] for r in range(4)]
        torch.cuda.synchronize()
        same = lambda a, b: torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0))
        assert got.shape == full.shape and same(got, full)
        assert same(parallel.assemble_z_slabs(torch.stack(slabs).contiguous()), full)
        # an all-reduce for good measure: the ranks RCCL connected
        one = torch.ones(1, device=gpu_device)
        dist.all_reduce(one)
        assert float(one) == 1.0
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('shape', [(8, 1, 15, 16, 12, 16), (3, 2, 5, 4, 3, 5), (2, 1, 3, 7, 5, 4)])
def test_assemble_zslabs_equals_the_permute(shape, gpu_device):
    """tempo_assemble_zslabs (16-byte and 4-byte paths) vs the torch permute it replaces."""
    from tempo_amd import parallel
    g = torch.Generator().manual_seed(sum(shape))
    slabs = torch.rand(*shape, generator=g).to(gpu_device)
    R, B, C, X, Y, Zr = shape
    ref = slabs.permute(1, 2, 3, 4, 0, 5).reshape(B, C, X, Y, R * Zr)
    got = parallel.assemble_z_slabs(slabs)
    assert got.shape == ref.shape and torch.equal(got, ref)
    assert torch.equal(parallel.assemble_z_slabs(slabs.cpu()), ref.cpu())  # the gloo tests' path
