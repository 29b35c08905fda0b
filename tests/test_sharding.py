"""N > 1 path on CPU: two gloo ranks.  The compute is the oracle (a stand-in
for the HIP modules with the same interface); what is tested is that the
frames x proposals partition and the z-slab all-gather reassemble exactly the
unsharded result."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases
from oracle import project_layer_ref as O
from tempo_amd import parallel


class _SlabOracle:
    """oracle ProjectLayer with the ``zslab`` extension of tempo_amd.ProjectLayer"""

    def __init__(self, image_size, heatmap_size):
        self.inner = O.ProjectLayer(image_size, heatmap_size)

    def get_voxel(self, *args, zslab=None):
        cubes, grids = self.inner.get_voxel(*args)
        if zslab is not None:
            cubes = cubes[..., zslab[0]:zslab[1]].contiguous()
        return cubes, grids


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        # (1) frames x proposals partition, no collective on the data path
        case = cases.build_root('root_panoptic_small')
        fms, gs, gc, cube, centers, scales, cameras = case['args']
        full, _ = O.ProjectLayer(case['image_size'], case['heatmap_size'])(*case['args'])
        sfm, (sc, ss, scam), (lo, hi) = parallel.shard_batch(
            fms, (centers, scales, cameras), rank, world)
        mine, _ = O.ProjectLayer(case['image_size'], case['heatmap_size'])(
            sfm, gs, gc, cube, sc, ss, scam)
        ok1 = torch.equal(mine, full[lo:hi])
        # (2) z-slab sharding + one all-gather
        layer = _SlabOracle(case['image_size'], case['heatmap_size'])
        whole = parallel.unproject_root_zslab(layer, *case['args'])
        ok2 = torch.equal(whole, full)
        # (3) window variant: windows sharded with their frames
        case = cases.build_window('win_panoptic_small')
        fms, props, *geo, centers, scales, cameras = case['args']
        fullw, offw = O.ProjectLayerWithMask(case['image_size'], case['heatmap_size'])(*case['args'])
        sfm, (sp, sc, ss, scam), (lo, hi) = parallel.shard_batch(
            fms, (props, centers, scales, cameras), rank, world)
        minew, off = O.ProjectLayerWithMask(case['image_size'], case['heatmap_size'])(
            sfm, sp, *geo, sc, ss, scam)
        ok3 = torch.equal(minew, fullw[lo:hi]) and torch.equal(off, offw[lo:hi])
        torch.save((ok1, ok2, ok3), os.path.join(out_dir, f'r{rank}.pt'))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharding(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert torch.load(os.path.join(str(tmp_path), f'r{r}.pt')) == (True, True, True)


def test_shard_ranges_cover_everything():
    for n in (1, 7, 32, 320, 1024):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert parallel.zslab_range(128, 7, 8) == (112, 128)
    with pytest.raises(ValueError):
        parallel.zslab_range(20, 0, 8)
