"""tempo_root_step_static -- the entry point that allocates nothing (its staging buffer
is a file-scope __device__ array) -- against tempo_root_step with a caller-allocated
workspace, through the C ABI: volume, grids and valid-view mask bit-identical, on two
calls in a row, the second with a smaller input (the array keeps the first call's
maps).  Collected last (test_zz_*): a failure here must not hide other tests."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(dev, B, V, J, H, W, cube, seed):
    from tempo_amd import synthetic as S
    from tempo_amd.camera import pack_cameras
    cfg = S.PANOPTIC
    cams = S.ring_cameras(V, seed=seed, target=cfg['space_center'])
    centers, scales, cameras = S.frame_meta(cams, B, cfg['raw_size'], cfg['image_size'])
    cam = pack_cameras(centers, scales, cameras, cfg['image_size']).to(dev)
    fms = [f.to(dev) for f in S.make_heatmaps(B, V, J, H, W, kind='noise', seed=seed + 1)]
    axes = [(torch.linspace(-s / 2, s / 2, n) + c).contiguous().to(dev)
            for s, n, c in zip(cfg['space_size'], cube, cfg['space_center'])]
    return cfg, fms, cam, axes


def _call(entry, dev, cfg, fms, cam, axes, J, cube, chunk, workspace=None):
    """One call through ctypes; entry = 'tempo_root_step' (with workspace) or
    'tempo_root_step_static' (without)."""
    from tempo_amd import _lib, numerics, ops
    lib = _lib.load()
    B, _, H, W = fms[0].shape
    V = len(fms)
    cp = ops.padded_channels(J)
    X, Y, Z = cube
    N = X * Y * Z
    out = torch.full((B, J, X, Y, Z), -3.0, device=dev)
    grids = torch.full((B, N, 3), -3.0, device=dev)
    mask = torch.full((B, N), 77, dtype=torch.uint8, device=dev)
    ptrs = (ctypes.c_void_p * V)(*[f.data_ptr() for f in fms])
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    head = (ptrs, J, 0, P(cam), P(axes[0]), P(axes[1]), P(axes[2]), 0, None, P(out), P(grids),
            P(mask), 1, B, V, J, cp, H, W, X, Y, Z, float(cfg['heatmap_size'][0]),
            float(cfg['heatmap_size'][1]), float(cfg['image_size'][0]),
            float(cfg['image_size'][1]), numerics.abi_flag('auto'))
    with torch.cuda.device(dev):
        if entry == 'tempo_root_step':
            rc = lib.tempo_root_step(*head, P(workspace), chunk, stream)
        else:
            rc = lib.tempo_root_step_static(*head, chunk, stream)
    assert rc == 0, (entry, rc, lib.tempo_error_string(rc))
    torch.cuda.synchronize()
    return out, grids, mask


def test_static_scratch_entry_equals_workspace_entry(gpu_device):
    from tempo_amd import _lib, ops
    dev = gpu_device
    canon = lambda t: torch.nan_to_num(t.float(), nan=-1.0)
    # first call: 7 frames of 5 x 128 x 240 maps (9.8 MB per frame: 6 frames fit the 64 MiB
    # array, so chunk 8 is cut to 6 and the last chunk is ragged); second call: smaller
    # maps, fewer frames, fewer views, 17 channels (padded to 20)
    for (B, V, J, H, W, cube, chunk, seed) in ((7, 5, 15, 128, 240, [24, 20, 10], 8, 301),
                                               (3, 4, 17, 64, 96, [12, 10, 9], 2, 302)):
        cfg, fms, cam, axes = _inputs(dev, B, V, J, H, W, cube, seed)
        cfg = dict(cfg, heatmap_size=[W, H])
        cp = ops.padded_channels(J)
        ws = ops.root_step_workspace(B, V, H, W, cp, chunk, dev)
        ref = _call('tempo_root_step', dev, cfg, fms, cam, axes, J, cube, chunk, ws)
        got = _call('tempo_root_step_static', dev, cfg, fms, cam, axes, J, cube, chunk)
        for a, b, what in zip(got, ref, ('cubes', 'grids', 'mask')):
            assert torch.equal(canon(a), canon(b)), (what, B, V, J)
        assert float(got[0].min()) >= 0.0 and float(got[0].max()) > 0.0


def test_static_scratch_rejects_what_does_not_fit(gpu_device):
    """One frame's maps beyond the array: a clear error, nothing launched."""
    from tempo_amd import _lib
    lib = _lib.load()
    one = ctypes.c_void_p(16)
    ptrs = (ctypes.c_void_p * 8)(*([16] * 8))
    # 8 views x 1024 x 1024 x 16 channels x 4 B = 512 MiB per frame > 64 MiB
    rc = lib.tempo_root_step_static(ptrs, 15, 0, one, one, one, one, 0, None, one, None, None, 0,
                                    1, 8, 15, 16, 1024, 1024, 8, 8, 8, 1024.0, 1024.0, 1024.0,
                                    1024.0, 0, 1, None)
    assert rc == -4
    assert b'static scratch' in lib.tempo_error_string(rc)
    assert lib.tempo_root_step_static(ptrs, 15, 0, one, one, one, one, 0, None, one, None, None,
                                      0, 1, 8, 15, 16, 64, 64, 8, 8, 8, 64.0, 64.0, 64.0, 64.0,
                                      0, 0, None) == -2  # chunk <= 0


def test_static_scratch_entry_under_graph_capture(gpu_device):
    """The callers the entry point exists for: captured into a HIP graph (where hipMalloc is
    not allowed) and replayed on new maps written into the same tensors."""
    from tempo_amd import _lib, numerics, ops
    lib = _lib.load()
    dev = gpu_device
    B, V, J, H, W, cube, chunk = 5, 5, 15, 64, 96, [12, 10, 9], 2
    cfg, fms, cam, axes = _inputs(dev, B, V, J, H, W, cube, 311)
    cfg = dict(cfg, heatmap_size=[W, H])
    # eager first: the array's device address is looked up at the first call on a device
    ref1 = _call('tempo_root_step_static', dev, cfg, fms, cam, axes, J, cube, chunk)[0]
    cp = ops.padded_channels(J)
    X, Y, Z = cube
    out = torch.zeros(B, J, X, Y, Z, device=dev)
    ptrs = (ctypes.c_void_p * V)(*[f.data_ptr() for f in fms])
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            rc = lib.tempo_root_step_static(
                ptrs, J, 0, P(cam), P(axes[0]), P(axes[1]), P(axes[2]), 0, None, P(out), None,
                None, 0, B, V, J, cp, H, W, X, Y, Z, float(W), float(H),
                float(cfg['image_size'][0]), float(cfg['image_size'][1]),
                numerics.abi_flag('auto'), chunk,
                ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    assert rc == 0, lib.tempo_error_string(rc)
    torch.cuda.current_stream(dev).wait_stream(side)
    canon = lambda t: torch.nan_to_num(t, nan=-1.0)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(canon(out), canon(ref1))
    for f in fms:  # new maps in the captured tensors
        f.mul_(0.5).add_(0.25)
    ref2 = _call('tempo_root_step_static', dev, cfg, fms, cam, axes, J, cube, chunk)[0]
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(canon(out), canon(ref2)) and not torch.equal(canon(ref1), canon(ref2))
