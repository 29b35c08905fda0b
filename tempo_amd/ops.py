"""Functional layer over the C ABI: torch tensors in, torch tensors out.

Everything here is plumbing (device memory, streams, argument checks); the
arithmetic happens in ``csrc/tempo_unproject.hip``.  All entry points require
CUDA/HIP tensors and raise otherwise -- there is no CPU path.
"""
import ctypes

import torch

from . import _lib, numerics


_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _raw_stream(device):
    """hipStream_t of torch's current stream on ``device`` as an int (0 = the null stream).
    The raw accessor skips the Stream object: 0.3 instead of 4 us, several times per call."""
    index = device.index
    if index is None:
        index = torch.cuda.current_device()
    if _RAW_STREAM is not None:
        return _RAW_STREAM(index)
    return torch.cuda.current_stream(device).cuda_stream


def _stream_ptr(device):
    return ctypes.c_void_p(_raw_stream(device))


_STAGING = {}
STAGING_MAX_BYTES = 512 << 20  # larger requests are plain allocations, not kept


def release_staging():
    """Drop the persistent staging buffers (see :func:`_staging`); they come back on demand."""
    _STAGING.clear()


def _staging(dev, numel):
    """A persistent fp32 staging buffer of at least ``numel`` elements for the current stream
    of ``dev`` (the channel-last maps of one chunk: written and read inside one call, dead
    afterwards).  Kept across calls, grow-only, instead of asking the caching allocator for
    ~100 MB per call.  One buffer per (device, stream): calls on different streams do not
    share it.  Not kept: requests above ``STAGING_MAX_BYTES`` (a whole batch's maps for the
    one-launch planes step), and anything asked for while the stream is being captured into
    a graph (the allocation would belong to the capture's private pool);
    :func:`release_staging` frees the rest."""
    numel = int(numel)
    if numel * 4 > STAGING_MAX_BYTES or torch.cuda.is_current_stream_capturing():
        return torch.empty(numel, dtype=torch.float32, device=dev)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(),
           _raw_stream(dev))
    buf = _STAGING.get(key)
    if buf is None or buf.numel() < numel:
        buf = None
        _STAGING.pop(key, None)
        if len(_STAGING) > 16:
            _STAGING.clear()
        buf = torch.empty(numel, dtype=torch.float32, device=dev)
        _STAGING[key] = buf
    return buf


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _no_grad_path(what, *tensors):
    """Forward-only entry points raise when autograd would expect a gradient through them
    (the torch expressions they replace carry one); silently cutting the graph would train
    on wrong gradients."""
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise NotImplementedError(
            f'{what} is forward-only: call it under torch.no_grad() / on detached tensors, '
            'or use the torch expression it replaces for a training step')


def _need(t, dtype, name, device=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError(f'{name}: expected a tensor on a HIP device (tempo_amd has no CPU path)')
    if t.dtype != dtype:
        raise TypeError(f'{name}: expected {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise ValueError(f'{name}: must be contiguous')
    if device is not None and t.device != device:
        raise ValueError(f'{name}: on {t.device}, expected {device}')
    return t


def padded_channels(channels):
    cp = _lib.load().tempo_padded_channels(int(channels))
    _lib.check(min(cp, 0), 'tempo_padded_channels')
    return cp


HALF_DTYPES = {torch.float16: 1, torch.bfloat16: 2}  # TEMPO_DTYPE_F16 / TEMPO_DTYPE_BF16


def repack_nhwc(feature_maps, c_begin=0, channels=None, out=None):
    """List of V ``[B,C,H,W]`` tensors -> fp32 ``[B,V,H,W,CP]`` channel-last.  fp16 / bf16
    maps (one dtype for all views) are widened inside the repack -- the reference's
    ``@force_fp32`` without the fp32 copy; any other dtype goes through ``.float()``."""
    lib = _lib.load()
    V = len(feature_maps)
    fm0 = feature_maps[0]
    dev = fm0.device
    B, C_total, H, W = fm0.shape
    C = C_total - c_begin if channels is None else channels
    CP = padded_channels(C)
    half = HALF_DTYPES.get(fm0.dtype, 0)
    if half and any(fm.dtype != fm0.dtype for fm in feature_maps):
        half = 0
    keep = []
    for i, fm in enumerate(feature_maps):
        if tuple(fm.shape) != (B, C_total, H, W):
            raise ValueError('feature maps of all views must share one shape')
        if half:
            keep.append(_need(fm.contiguous(), fm0.dtype, f'feature_maps[{i}]', dev))
            continue
        if fm.dtype != torch.float32:
            fm = fm.float()  # the reference's force_fp32
        keep.append(_need(fm.contiguous(), torch.float32, f'feature_maps[{i}]', dev))
    if out is None:
        out = torch.empty(B, V, H, W, CP, dtype=torch.float32, device=dev)
    else:
        _need(out, torch.float32, 'out', dev)
        if out.numel() != B * V * H * W * CP:
            raise ValueError('out has the wrong size')
    ptrs = (ctypes.c_void_p * V)(*[fm.data_ptr() for fm in keep])
    with torch.cuda.device(dev):
        if half:
            rc = lib.tempo_repack_nhwc_half(ptrs, half, V, B, C_total, c_begin, C, H, W, CP,
                                            _ptr(out), _stream_ptr(dev))
        else:
            rc = lib.tempo_repack_nhwc(ptrs, V, B, C_total, c_begin, C, H, W, CP,
                                       _ptr(out), _stream_ptr(dev))
    _lib.check(rc, 'tempo_repack_nhwc_half' if half else 'tempo_repack_nhwc')
    return out


GROUP_MAX_ITEMS = 8  # items per work item of a grouped launch (amortisation vs parallelism)


def share_groups(keys, max_items=None):
    """Group table for the ``*_grouped`` entry points: runs of consecutive items
    with equal ``keys[i]`` (any hashable: the bytes of the item's camera records,
    its grid centre, its window geometry ...), cut into pieces of at most
    ``max_items``.  Returns a CPU int32 ``[NG,2]`` tensor of (first, count), or
    None when nothing is shared."""
    max_items = GROUP_MAX_ITEMS if max_items is None else int(max_items)
    rows, i, n = [], 0, len(keys)
    while i < n:
        j = i + 1
        while j < n and j - i < max_items and keys[j] == keys[i]:
            j += 1
        rows.append((i, j - i))
        i = j
    if len(rows) == n:
        return None
    return torch.tensor(rows, dtype=torch.int32)


def camera_keys(cam_cpu, extra=None):
    """One hashable key per item from its packed camera records (bit-exact
    comparison) and an optional per-item ``extra`` (list of hashables)."""
    arr = cam_cpu.contiguous().numpy()
    return [(arr[b].tobytes(), None if extra is None else extra[b])
            for b in range(arr.shape[0])]


def _groups_arg(groups, n_items, dev):
    if groups is None:
        return None, 0
    _need(groups, torch.int32, 'groups', dev)
    if groups.dim() != 2 or groups.shape[1] != 2 or groups.shape[0] > n_items:
        raise ValueError('groups must be int32 [NG,2] with NG <= number of items')
    return groups, int(groups.shape[0])


def clamp_mask_words(num_voxels, cp):
    n = _lib.load().tempo_clamp_mask_words(int(num_voxels), int(cp))
    _lib.check(min(n, 0), 'tempo_clamp_mask_words')
    return n


def new_clamp_mask(items, num_voxels, cp, device):
    """Buffer for the clamp mask a training forward records (int64 words)."""
    return torch.empty(items, clamp_mask_words(num_voxels, cp), dtype=torch.int64, device=device)


def _check_clamp_mask(t, items, num_voxels, cp, dev):
    _need(t, torch.int64, 'clamp_mask', dev)
    if t.numel() != items * clamp_mask_words(num_voxels, cp):
        raise ValueError('clamp_mask has the wrong size (see new_clamp_mask)')


def unrepack_nchw(hm, channels_total, c_begin=0, channels=None, out=None):
    """Inverse of :func:`repack_nhwc`: ``[B,V,H,W,CP]`` -> list of V fp32
    ``[B,channels_total,H,W]`` tensors (channels ``[c_begin, c_begin+channels)``
    written; ``out`` = existing tensors to write into, else new zero tensors)."""
    lib = _lib.load()
    dev = hm.device
    _need(hm, torch.float32, 'hm')
    B, V, H, W, CP = hm.shape
    C = channels_total - c_begin if channels is None else channels
    if out is None:
        out = [torch.zeros(B, channels_total, H, W, dtype=torch.float32, device=dev)
               for _ in range(V)]
    for i, t in enumerate(out):
        _need(t, torch.float32, f'out[{i}]', dev)
        if tuple(t.shape) != (B, channels_total, H, W):
            raise ValueError('out tensors must be [B, channels_total, H, W]')
    ptrs = (ctypes.c_void_p * V)(*[t.data_ptr() for t in out])
    with torch.cuda.device(dev):
        rc = lib.tempo_unrepack_nchw(_ptr(hm), ptrs, V, B, channels_total, c_begin, C, H, W,
                                     CP, _stream_ptr(dev))
    _lib.check(rc, 'tempo_unrepack_nchw')
    return out


def unproject_root_backward(grad_cubes, cam, axes, clamp_mask, hm_shape, heatmap_size,
                            image_size, live=None, mm_mode='auto'):
    """Gradient of :func:`unproject_root`'s volume with respect to the repacked
    heatmaps: ``grad_cubes [B,C,X,Y,Z]`` -> ``[B,V,H,W,CP]`` (``hm_shape``).
    ``clamp_mask`` is the one the training forward recorded."""
    lib = _lib.load()
    dev = grad_cubes.device
    _need(grad_cubes, torch.float32, 'grad_cubes')
    B, V, H, W, CP = (int(v) for v in hm_shape)
    _need(cam, torch.float32, 'cam', dev)
    ax = [_need(a, torch.float32, 'axes', dev) for a in axes]
    batched = ax[0].dim() == 2
    X, Y, Z = (int(a.shape[-1]) for a in ax)
    C = int(grad_cubes.shape[1])
    if tuple(grad_cubes.shape) != (B, C, X, Y, Z):
        raise ValueError(f'grad_cubes must be [{B},C,{X},{Y},{Z}]')
    _check_clamp_mask(clamp_mask, B, X * Y * Z, CP, dev)
    if live is not None:
        _need(live, torch.uint8, 'live', dev)
    ghm = torch.empty(B, V, H, W, CP, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.tempo_unproject_root_backward(
            _ptr(grad_cubes), _ptr(cam), _ptr(ax[0]), _ptr(ax[1]), _ptr(ax[2]), int(batched),
            _ptr(live), _ptr(clamp_mask), _ptr(ghm), B, V, C, CP, H, W, X, Y, Z,
            float(heatmap_size[0]), float(heatmap_size[1]), float(image_size[0]),
            float(image_size[1]), numerics.abi_flag(mm_mode), _stream_ptr(dev))
    _lib.check(rc, 'tempo_unproject_root_backward')
    return ghm


BACKWARD_TILE_PITCH = 2.2  # heatmap pixels per voxel from which the 144-texel tile pays


def backward_tile_texels(cam_host, fine_axes_host, heatmap_size, image_size, planes=False):
    """Tile size for :func:`unproject_window_backward` (``planes``: for
    :func:`unproject_window_planes_backward`, see the end) from the lattice's pitch in heatmap
    pixels: voxel pitch x focal length x affine scale x heatmap / image width, over the
    cameras' distance to the middle of the lattice (host arithmetic on the packed camera
    table, no device sync).  121 texels up to ``BACKWARD_TILE_PITCH`` pixels per voxel (config
    3: 1.5), 144 beyond (H36M windows on 384 x 384 maps: 2.8; 5.00 -> 4.67 ms).  The planes'
    backward boxes only a brick's hot voxels, a third of them, and wants the small tile at
    either pitch (ms per launch at 81 / 121 / 144 texels: config 3 10.9 / 13.5 / 15.9, the
    H36M slice 8.6 / 10.8 / 11.2): 81."""
    cam = cam_host.reshape(-1, 32)[:64].double()
    ax = [a.double() for a in fine_axes_host]
    mid = torch.stack([a[a.numel() // 2] for a in ax])
    pitch_mm = max(float((a[-1] - a[0]).abs() / max(a.numel() - 1, 1)) for a in ax)
    depth = (mid[None, :] @ cam[:, 0:9].reshape(-1, 3, 3))[:, 0, 2] + cam[:, 11]
    px = pitch_mm * cam[:, 12].abs() * cam[:, 24].abs() * (heatmap_size[0] / image_size[0]) \
        / depth.abs().clamp_min(1.0)
    if planes:
        return 81
    return 144 if float(px.median()) >= BACKWARD_TILE_PITCH else 121


def unproject_window_backward(grad_cubes, cam, fine_axes, tl, start, end, clamp_mask,
                              hm_shape, heatmap_size, image_size, win_frame=None, live=None,
                              mm_mode='auto', tile_texels=0):
    """Gradient of :func:`unproject_window`'s volume with respect to the repacked
    heatmaps: ``grad_cubes [NW,C,VX,VY,VZ]`` -> ``[F,V,H,W,CP]``.  ``tile_texels``: 0
    (library default), 121 or 144, see :func:`backward_tile_texels`."""
    lib = _lib.load()
    dev = grad_cubes.device
    _need(grad_cubes, torch.float32, 'grad_cubes')
    F_, V, H, W, CP = (int(v) for v in hm_shape)
    _need(cam, torch.float32, 'cam', dev)
    fa = [_need(a, torch.float32, 'fine_axes', dev) for a in fine_axes]
    NW, C, VX, VY, VZ = (int(v) for v in grad_cubes.shape)
    for name, t in (('tl', tl), ('start', start), ('end', end)):
        _need(t, torch.int32, name, dev)
        if tuple(t.shape) != (NW, 3):
            raise ValueError(f'{name} must be [{NW},3]')
    if win_frame is not None:
        _need(win_frame, torch.int32, 'win_frame', dev)
    elif NW != F_:
        raise ValueError('without win_frame there must be one window per frame')
    if live is not None:
        _need(live, torch.uint8, 'live', dev)
    _check_clamp_mask(clamp_mask, NW, VX * VY * VZ, CP, dev)
    ghm = torch.empty(F_, V, H, W, CP, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.tempo_unproject_window_backward(
            _ptr(grad_cubes), _ptr(cam), _ptr(win_frame), _ptr(fa[0]), _ptr(fa[1]),
            _ptr(fa[2]), _ptr(tl), _ptr(start), _ptr(end), _ptr(live), _ptr(clamp_mask),
            _ptr(ghm), NW, F_, V, C, CP, H, W, int(fa[0].numel()), int(fa[1].numel()),
            int(fa[2].numel()), VX, VY, VZ, float(heatmap_size[0]), float(heatmap_size[1]),
            float(image_size[0]), float(image_size[1]), numerics.abi_flag(mm_mode),
            int(tile_texels), _stream_ptr(dev))
    _lib.check(rc, 'tempo_unproject_window_backward')
    return ghm


def window_planes_claims(hm, cam, fine_axes, tl, start, end, sub_cube, channels, planes,
                         heatmap_size, image_size, win_frame=None, live=None, mm_mode='auto'):
    """For training through :func:`unproject_window_planes`: per entry of the three FINAL
    ``planes`` (that call's outputs on the same inputs) the claim of the voxel ``torch.max``
    sends the entry's gradient to -- int32 tensors in the planes' shapes, 2 * (index along
    the reduced axis of the first voxel attaining the maximum) + 1 if the clamp blocks the
    gradient there (``tempo_window_planes_claims``)."""
    lib = _lib.load()
    dev, (F_, V, H, W, CP), fa, NW, (VX, VY, VZ), C = _window_args(
        hm, cam, fine_axes, tl, start, end, sub_cube, channels, win_frame, live)
    shapes = [(NW, C, VX, VY), (NW, C, VX, VZ), (NW, C, VY, VZ)]
    planes = list(planes)
    if len(planes) != 3:
        raise ValueError('planes: the xy, xz and yz planes')
    for t, sh in zip(planes, shapes):
        _need(t, torch.float32, 'planes', dev)
        if tuple(t.shape) != sh:
            raise ValueError(f'planes: expected {sh}, got {tuple(t.shape)}')
    claims = [torch.empty(sh, dtype=torch.int32, device=dev) for sh in shapes]
    with torch.cuda.device(dev):
        rc = lib.tempo_window_planes_claims(
            _ptr(hm), _ptr(cam), _ptr(win_frame), _ptr(fa[0]), _ptr(fa[1]), _ptr(fa[2]),
            _ptr(tl), _ptr(start), _ptr(end), _ptr(live), _ptr(planes[0]), _ptr(planes[1]),
            _ptr(planes[2]), _ptr(claims[0]), _ptr(claims[1]), _ptr(claims[2]), NW, F_, V, C,
            CP, H, W, int(fa[0].numel()), int(fa[1].numel()), int(fa[2].numel()), VX, VY, VZ,
            float(heatmap_size[0]), float(heatmap_size[1]), float(image_size[0]),
            float(image_size[1]), numerics.abi_flag(mm_mode), _stream_ptr(dev))
    _lib.check(rc, 'tempo_window_planes_claims')
    return claims


def unproject_window_planes_backward(grad_planes, claims, cam, fine_axes, tl, start, end,
                                     hm_shape, heatmap_size, image_size, win_frame=None,
                                     live=None, mm_mode='auto', tile_texels=0):
    """Gradient of :func:`unproject_window_planes`'s three planes with respect to the
    repacked heatmaps, routed through :func:`window_planes_claims`:
    ``grad_planes`` (xy, xz, yz) -> ``[F,V,H,W,CP]``.  Neither the window volume nor its
    gradient volume exists at any point."""
    lib = _lib.load()
    dev = grad_planes[0].device
    F_, V, H, W, CP = (int(v) for v in hm_shape)
    _need(cam, torch.float32, 'cam', dev)
    fa = [_need(a, torch.float32, 'fine_axes', dev) for a in fine_axes]
    NW, C, VX, VY = (int(v) for v in grad_planes[0].shape)
    VZ = int(grad_planes[1].shape[3])
    shapes = [(NW, C, VX, VY), (NW, C, VX, VZ), (NW, C, VY, VZ)]
    if len(grad_planes) != 3 or len(claims) != 3:
        raise ValueError('grad_planes and claims: one tensor per plane (xy, xz, yz)')
    for g, c, sh in zip(grad_planes, claims, shapes):
        _need(g, torch.float32, 'grad_planes', dev)
        _need(c, torch.int32, 'claims', dev)
        if tuple(g.shape) != sh or tuple(c.shape) != sh:
            raise ValueError(f'grad_planes / claims: expected {sh}')
    for name, t in (('tl', tl), ('start', start), ('end', end)):
        _need(t, torch.int32, name, dev)
        if tuple(t.shape) != (NW, 3):
            raise ValueError(f'{name} must be [{NW},3]')
    if tuple(cam.shape) != (F_, V, 32):
        raise ValueError(f'cam must be [{F_},{V},32], got {tuple(cam.shape)}')
    if win_frame is not None:
        _need(win_frame, torch.int32, 'win_frame', dev)
        if win_frame.numel() != NW:
            raise ValueError('win_frame must have one entry per window')
    elif NW != F_:
        raise ValueError('without win_frame there must be one window per frame')
    if live is not None:
        _need(live, torch.uint8, 'live', dev)
        if live.numel() != NW:
            raise ValueError('live must have one entry per window')
    if C > CP:
        raise ValueError('hm_shape: fewer padded channels than the planes have channels')
    ghm = torch.empty(F_, V, H, W, CP, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.tempo_unproject_window_planes_backward(
            _ptr(grad_planes[0]), _ptr(grad_planes[1]), _ptr(grad_planes[2]), _ptr(claims[0]),
            _ptr(claims[1]), _ptr(claims[2]), _ptr(cam), _ptr(win_frame), _ptr(fa[0]),
            _ptr(fa[1]), _ptr(fa[2]), _ptr(tl), _ptr(start), _ptr(end), _ptr(live), _ptr(ghm),
            NW, F_, V, C, CP, H, W, int(fa[0].numel()), int(fa[1].numel()), int(fa[2].numel()),
            VX, VY, VZ, float(heatmap_size[0]), float(heatmap_size[1]), float(image_size[0]),
            float(image_size[1]), numerics.abi_flag(mm_mode), int(tile_texels),
            _stream_ptr(dev))
    _lib.check(rc, 'tempo_unproject_window_planes_backward')
    return ghm


# channel-last maps of one chunk stay in the 256 MB Infinity Cache between the repack that
# writes them and the gather that reads them; the volume goes out with streaming stores and
# does not compete for it (measured at config 2: 8 / 12 / 16 / 32 frames per chunk
# 0.315 / 0.297 / 0.279 / 0.375 ms per 32-frame step)
ROOT_CHUNK_FRAMES = 16
CHUNK_BYTES = 190e6  # (16 Panoptic frames = 157 MB; 4 H36M frames = 189 MB: 3 / 4 / 8 frames 4.26 / 4.03 / 4.72 ms)


def chunk_frames(V, H, W, cp, limit=ROOT_CHUNK_FRAMES):
    """Frames per repack+unproject chunk so that the chunk's channel-last maps stay
    cache-resident between the two kernels (at most ``limit`` = one frame per XCD)."""
    return int(max(1, min(limit, CHUNK_BYTES // (V * H * W * cp * 4))))


_WIN_FRAME = {}


def _regular_win_frame(frames, per_frame, dev):
    """int32 ``[frames * per_frame]``: window j looks at frame ``j // per_frame`` (cached: the
    chunk loop of :func:`window_step` asks for the same two tables every step)."""
    key = (frames, per_frame, str(dev))
    if key not in _WIN_FRAME:
        if len(_WIN_FRAME) > 64:
            _WIN_FRAME.clear()
        _WIN_FRAME[key] = (torch.arange(frames * per_frame, device=dev, dtype=torch.int32)
                           // per_frame).contiguous()
    return _WIN_FRAME[key]


def window_step(feature_maps, cam, fine_axes, tl, start, end, sub_cube, heatmap_size,
                image_size, windows_per_frame=1, c_begin=0, channels=None, live=None,
                out=None, planes_out=None, planes=False, want_cubes=True, mm_mode='auto',
                chunk=None, hm_buf=None):
    """Repack + window unprojection for a regular batch -- window j looks at frame
    ``j // windows_per_frame`` -- in chunks of frames that reuse one channel-last
    buffer (see :func:`root_step`; at the H36M map size the repack otherwise writes
    6 GB to HBM per 128 frames of which the windows read 5 %).  ``planes=True`` adds
    the fused axis-max planes (:func:`unproject_window_planes`).  Returns
    ``(cubes | None, planes | None)``."""
    fm0 = feature_maps[0]
    dev = fm0.device
    F_, C_total, H, W = fm0.shape
    V = len(feature_maps)
    C = C_total - c_begin if channels is None else channels
    P = int(windows_per_frame)
    NW = int(tl.shape[0])
    if NW != F_ * P:
        raise ValueError('window_step needs windows_per_frame windows for every frame, in order')
    cp = padded_channels(C)
    if chunk is None:  # cache-sized (config 3: 2 / 4 / 8 / 16 / 32 frames per chunk
        # 3.73 / 3.60 / 3.57 / 3.55 / 3.76 ms per step)
        chunk = chunk_frames(V, H, W, cp)
        if planes:  # the column-walk kernel has long workgroups: one launch is faster
            chunk = F_
    chunk = int(chunk)
    vx, vy, vz = (int(v) for v in sub_cube)
    if want_cubes and out is None:
        out = torch.empty(NW, C, vx, vy, vz, dtype=torch.float32, device=dev)
    if planes and planes_out is None:
        planes_out = [torch.empty(NW, C, *sh, dtype=torch.float32, device=dev)
                      for sh in ((vx, vy), (vx, vz), (vy, vz))]
    if hm_buf is None:
        n0 = min(chunk, F_)
        hm_buf = _staging(dev, n0 * V * H * W * cp)[:n0 * V * H * W * cp].view(n0, V, H, W, cp)
    for f0 in range(0, F_, chunk):
        n = min(chunk, F_ - f0)
        fs, ws = slice(f0, f0 + n), slice(f0 * P, (f0 + n) * P)
        hm = repack_nhwc([fm[fs] for fm in feature_maps], c_begin, C, out=hm_buf[:n])
        wf = None if P == 1 else _regular_win_frame(n, P, dev)
        lv = None if live is None else live[ws]
        if planes:
            unproject_window_planes(hm, cam[fs], fine_axes, tl[ws], start[ws], end[ws], sub_cube,
                                    C, heatmap_size, image_size, win_frame=wf, live=lv,
                                    want_cubes=want_cubes, mm_mode=mm_mode,
                                    out=out[ws] if want_cubes else None,
                                    planes_out=[p[ws] for p in planes_out])
        else:
            unproject_window(hm, cam[fs], fine_axes, tl[ws], start[ws], end[ws], sub_cube, C,
                             heatmap_size, image_size, win_frame=wf, live=lv, out=out[ws],
                             mm_mode=mm_mode)
    return (out if want_cubes else None), (planes_out if planes else None)


def mask_dtype(num_views):
    """(bytes, torch dtype) of one valid-view bitmap word for ``num_views`` views."""
    nbytes = 1 if num_views <= 8 else (2 if num_views <= 16 else 4)
    return nbytes, {1: torch.uint8, 2: torch.int16, 4: torch.int32}[nbytes]


def root_step(feature_maps, cam, axes, heatmap_size, image_size, c_begin=0, channels=None,
              live=None, want_grids=False, want_mask=False, out=None, mm_mode='auto',
              chunk=None, hm_buf=None, in_library=True, zmax_out=None, want_cubes=True):
    """Repack + root unprojection of a whole batch, in chunks of ``chunk`` frames
    (default :func:`chunk_frames`: 16 at config 2) that reuse one channel-last buffer: a
    chunk's maps stay in the 256 MB Infinity Cache between the repack that writes them and
    the gather that reads them, instead of making a round trip through HBM (config 2:
    0.39 -> 0.28 ms per 32-frame step).  Arguments as :func:`repack_nhwc` +
    :func:`unproject_root`; ``axes`` and ``live`` may be per batch item.  Returns
    ``(cubes, grids | None, mask | None)``.  ``want_cubes=False`` (with ``zmax_out``): the
    z-max planes alone, the volume is never written (``cubes`` is None)."""
    fm0 = feature_maps[0]
    dev = fm0.device
    B, C_total, H, W = fm0.shape
    V = len(feature_maps)
    C = C_total - c_begin if channels is None else channels
    chunk = chunk_frames(V, H, W, padded_channels(C)) if chunk is None else int(chunk)
    X, Y, Z = (int(a.shape[-1]) for a in axes)
    N = X * Y * Z
    if not want_cubes:
        if zmax_out is None:
            raise ValueError('want_cubes=False needs zmax_out (some output must be asked for)')
        out = None
    elif out is None:
        out = torch.empty(B, C, X, Y, Z, dtype=torch.float32, device=dev)
    grids = torch.empty(B, N, 3, dtype=torch.float32, device=dev) if want_grids else None
    mask = torch.empty(B, N, dtype=mask_dtype(V)[1], device=dev) if want_mask else None
    if zmax_out is not None:
        _need(zmax_out, torch.float32, 'zmax_out', dev)
        if tuple(zmax_out.shape) != (B, C, X, Y):
            raise ValueError(f'zmax_out must be [{B},{C},{X},{Y}]')
    # (16-bit maps: the library's step reads fp32 tensors; the chunk loop below widens them
    # chunk by chunk inside the repack instead of converting the whole batch first)
    all_half = fm0.dtype in HALF_DTYPES and all(fm.dtype == fm0.dtype for fm in feature_maps)
    if in_library and not all_half:  # the chunk loop inside the library: one call
        return _root_step_library(feature_maps, cam, axes, heatmap_size, image_size, c_begin, C,
                                live, out, grids, mask, mm_mode, chunk, hm_buf, zmax_out)
    if B <= chunk:
        hm = repack_nhwc(feature_maps, c_begin, C, out=None if hm_buf is None else hm_buf[:B])
        unproject_root(hm, cam, axes, C, heatmap_size, image_size, live=live,
                       want_grids=want_grids, want_mask=want_mask, out=out, mm_mode=mm_mode,
                       grids_out=grids, mask_out=mask, zmax_out=zmax_out,
                       want_cubes=want_cubes)
        return out, grids, mask
    cp = padded_channels(C)
    if hm_buf is None:
        hm_buf = _staging(dev, chunk * V * H * W * cp)[:chunk * V * H * W * cp].view(
            chunk, V, H, W, cp)
    batched = axes[0].dim() == 2
    for c0 in range(0, B, chunk):
        n = min(chunk, B - c0)
        sl = slice(c0, c0 + n)
        hm = repack_nhwc([fm[sl] for fm in feature_maps], c_begin, C, out=hm_buf[:n])
        unproject_root(hm, cam[sl], [a[sl] for a in axes] if batched else axes, C,
                       heatmap_size, image_size, live=None if live is None else live[sl],
                       want_grids=want_grids, want_mask=want_mask,
                       out=None if out is None else out[sl],
                       mm_mode=mm_mode, grids_out=None if grids is None else grids[sl],
                       mask_out=None if mask is None else mask[sl],
                       zmax_out=None if zmax_out is None else zmax_out[sl],
                       want_cubes=want_cubes)
    return out, grids, mask


def root_step_workspace(B, V, H, W, cp, chunk, device):
    """Workspace of :func:`root_step`: the channel-last maps of one chunk."""
    n = _lib.load().tempo_root_step_workspace(B, V, H, W, cp, chunk)
    _lib.check(min(int(n), 0), 'tempo_root_step_workspace')
    return torch.empty(int(n), dtype=torch.float32, device=device)


def _root_step_library(feature_maps, cam, axes, heatmap_size, image_size, c_begin, C, live, out,
                     grids, mask, mm_mode, chunk, workspace, zmax=None):
    """``tempo_root_step`` (``tempo_root_step_zmax`` with ``zmax``): the chunk loop of
    :func:`root_step` inside the library."""
    lib = _lib.load()
    fm0 = feature_maps[0]
    dev = fm0.device
    B, C_total, H, W = fm0.shape
    V = len(feature_maps)
    cp = padded_channels(C)
    keep = []
    for i, fm in enumerate(feature_maps):
        if tuple(fm.shape) != (B, C_total, H, W):
            raise ValueError('feature maps of all views must share one shape')
        if fm.dtype != torch.float32:
            fm = fm.float()
        keep.append(_need(fm.contiguous(), torch.float32, f'feature_maps[{i}]', dev))
    _need(cam, torch.float32, 'cam', dev)
    if tuple(cam.shape) != (B, V, 32):
        raise ValueError(f'cam must be [{B},{V},32], got {tuple(cam.shape)}')
    ax = [_need(a, torch.float32, 'axes', dev) for a in axes]
    batched = ax[0].dim() == 2
    for a in ax:
        if (a.dim() == 2) != batched or (batched and a.shape[0] != B):
            raise ValueError('axes must be all [n] or all [B,n]')
    X, Y, Z = (int(a.shape[-1]) for a in ax)
    if live is not None:
        _need(live, torch.uint8, 'live', dev)
        if live.numel() != B:
            raise ValueError('live must have B entries')
    if out is not None:
        _need(out, torch.float32, 'out', dev)
        if out.numel() != B * C * X * Y * Z:
            raise ValueError('out has the wrong size')
    elif zmax is None:
        raise ValueError('no output asked for')
    need = int(lib.tempo_root_step_workspace(B, V, H, W, cp, chunk))
    if workspace is None or workspace.numel() < need:
        workspace = _staging(dev, need)
    _need(workspace, torch.float32, 'workspace', dev)
    mask_bytes = 0 if mask is None else mask_dtype(V)[0]
    ptrs = (ctypes.c_void_p * V)(*[fm.data_ptr() for fm in keep])
    head = (ptrs, C_total, c_begin, _ptr(cam), _ptr(ax[0]), _ptr(ax[1]), _ptr(ax[2]),
            int(batched), _ptr(live), _ptr(out), _ptr(grids), _ptr(mask), mask_bytes, B, V, C,
            cp, H, W, X, Y, Z, float(heatmap_size[0]), float(heatmap_size[1]),
            float(image_size[0]), float(image_size[1]), numerics.abi_flag(mm_mode))
    with torch.cuda.device(dev):
        if zmax is None:
            rc = lib.tempo_root_step(*head, _ptr(workspace), int(chunk), _stream_ptr(dev))
        else:
            if not zmax.is_contiguous():
                raise ValueError('zmax_out must be contiguous')
            rc = lib.tempo_root_step_zmax(*head, _ptr(zmax), _ptr(workspace), int(chunk),
                                          _stream_ptr(dev))
    _lib.check(rc, 'tempo_root_step' if zmax is None else 'tempo_root_step_zmax')
    return out, grids, mask


def unproject_root(hm, cam, axes, channels, heatmap_size, image_size, live=None,
                   want_grids=False, want_mask=False, out=None, mm_mode='auto',
                   groups=None, clamp_mask=None, grids_out=None, mask_out=None, zmax_out=None,
                   want_cubes=True):
    """Root-grid unprojection on a repacked heatmap tensor.

    hm ``[B,V,H,W,CP]``; cam ``[B,V,32]``; axes = (ax_x, ax_y, ax_z), each
    ``[X]`` (shared) or ``[B,X]`` (per batch item).  ``groups``: device int32
    ``[NG,2]`` from :func:`share_groups` (items of a group share camera records
    and axes; the projection then runs once per group).  Returns
    ``(cubes [B,C,X,Y,Z], grids [B,N,3] | None, mask [B,N] | None)``.
    """
    lib = _lib.load()
    dev = hm.device
    _need(hm, torch.float32, 'hm')
    B, V, H, W, CP = hm.shape
    _need(cam, torch.float32, 'cam', dev)
    if tuple(cam.shape) != (B, V, 32):
        raise ValueError(f'cam must be [{B},{V},32], got {tuple(cam.shape)}')
    ax = [_need(a, torch.float32, 'axes', dev) for a in axes]
    batched = ax[0].dim() == 2
    for a in ax:
        if (a.dim() == 2) != batched or (batched and a.shape[0] != B):
            raise ValueError('axes must be all [n] or all [B,n]')
    X, Y, Z = (int(a.shape[-1]) for a in ax)
    N = X * Y * Z
    C = int(channels)
    if live is not None:
        _need(live, torch.uint8, 'live', dev)
        if live.numel() != B:
            raise ValueError('live must have B entries')
    if not want_cubes:  # z-max planes alone: the volume is never written
        if zmax_out is None:
            raise ValueError('want_cubes=False needs zmax_out')
        out = None
    elif out is None:
        out = torch.empty(B, C, X, Y, Z, dtype=torch.float32, device=dev)
    else:
        _need(out, torch.float32, 'out', dev)
        if out.numel() != B * C * N:
            raise ValueError('out has the wrong size')
    grids = None
    if want_grids:
        grids = grids_out if grids_out is not None else torch.empty(
            B, N, 3, dtype=torch.float32, device=dev)
        _need(grids, torch.float32, 'grids_out', dev)
        if grids.numel() != B * N * 3:
            raise ValueError('grids_out has the wrong size')
    mask, mask_bytes = None, 0
    if want_mask:
        mask_bytes, mdt = mask_dtype(V)
        mask = mask_out if mask_out is not None else torch.empty(B, N, dtype=mdt, device=dev)
        _need(mask, mdt, 'mask_out', dev)
        if mask.numel() != B * N:
            raise ValueError('mask_out has the wrong size')
    groups, ng = _groups_arg(groups, B, dev)
    if zmax_out is not None:  # also the volume's maximum over z (center_net.py:230)
        _need(zmax_out, torch.float32, 'zmax_out', dev)
        if zmax_out.numel() != B * C * X * Y:
            raise ValueError('zmax_out must be [B,C,X,Y]')
        if groups is not None or clamp_mask is not None:
            raise ValueError('zmax_out cannot be combined with groups or clamp_mask')
        with torch.cuda.device(dev):
            rc = lib.tempo_unproject_root_zmax(
                _ptr(hm), _ptr(cam), _ptr(ax[0]), _ptr(ax[1]), _ptr(ax[2]), int(batched),
                _ptr(live), _ptr(out), _ptr(grids), _ptr(mask), mask_bytes, B, V, C, CP, H,
                W, X, Y, Z, float(heatmap_size[0]), float(heatmap_size[1]),
                float(image_size[0]), float(image_size[1]), numerics.abi_flag(mm_mode),
                _ptr(zmax_out), _stream_ptr(dev))
        _lib.check(rc, 'tempo_unproject_root_zmax')
        return out, grids, mask
    if clamp_mask is not None:  # training forward: also record the clamp mask
        _check_clamp_mask(clamp_mask, B, N, CP, dev)
        with torch.cuda.device(dev):
            rc = lib.tempo_unproject_root_train(
                _ptr(hm), _ptr(cam), _ptr(ax[0]), _ptr(ax[1]), _ptr(ax[2]), int(batched),
                _ptr(live), _ptr(out), _ptr(grids), _ptr(mask), mask_bytes, B, V, C, CP, H,
                W, X, Y, Z, float(heatmap_size[0]), float(heatmap_size[1]),
                float(image_size[0]), float(image_size[1]), numerics.abi_flag(mm_mode),
                _ptr(clamp_mask), _stream_ptr(dev))
        _lib.check(rc, 'tempo_unproject_root_train')
        return out, grids, mask
    with torch.cuda.device(dev):
        rc = lib.tempo_unproject_root_grouped(
            _ptr(hm), _ptr(cam), _ptr(ax[0]), _ptr(ax[1]), _ptr(ax[2]), int(batched),
            _ptr(live), _ptr(out), _ptr(grids), _ptr(mask), mask_bytes, B, V, C, CP, H, W,
            X, Y, Z, float(heatmap_size[0]), float(heatmap_size[1]),
            float(image_size[0]), float(image_size[1]), numerics.abi_flag(mm_mode),
            _ptr(groups), ng, _stream_ptr(dev))
    _lib.check(rc, 'tempo_unproject_root_grouped')
    return out, grids, mask


def _window_args(hm, cam, fine_axes, tl, start, end, sub_cube, channels, win_frame, live):
    """Checks shared by the window entry points; returns
    ``(device, hm.shape, fine axes, NW, (VX, VY, VZ), C)``."""
    dev = hm.device
    _need(hm, torch.float32, 'hm')
    F_, V, H, W, CP = hm.shape
    _need(cam, torch.float32, 'cam', dev)
    if tuple(cam.shape) != (F_, V, 32):
        raise ValueError(f'cam must be [{F_},{V},32], got {tuple(cam.shape)}')
    fa = [_need(a, torch.float32, 'fine_axes', dev) for a in fine_axes]
    NW = int(tl.shape[0])
    for name, t in (('tl', tl), ('start', start), ('end', end)):
        _need(t, torch.int32, name, dev)
        if tuple(t.shape) != (NW, 3):
            raise ValueError(f'{name} must be [{NW},3]')
    if win_frame is not None:
        _need(win_frame, torch.int32, 'win_frame', dev)
        if win_frame.numel() != NW:
            raise ValueError('win_frame must have one entry per window')
    elif NW != F_:
        raise ValueError('without win_frame there must be one window per frame')
    if live is not None:
        _need(live, torch.uint8, 'live', dev)
        if live.numel() != NW:
            raise ValueError('live must have one entry per window')
    return dev, (F_, V, H, W, CP), fa, NW, tuple(int(v) for v in sub_cube), int(channels)


def unproject_window(hm, cam, fine_axes, tl, start, end, sub_cube, channels,
                     heatmap_size, image_size, win_frame=None, live=None, out=None,
                     mm_mode='auto', groups=None, clamp_mask=None):
    """Per-person window unprojection.  hm ``[F,V,H,W,CP]``, cam ``[F,V,32]``,
    tl/start/end int32 ``[NW,3]``, win_frame int32 ``[NW]`` or None (NW == F).
    ``groups``: device int32 ``[NG,2]``, runs of consecutive windows with equal
    tl/start/end whose frames have equal camera records (the T timesteps of one
    proposal).  Returns cubes ``[NW,C,VX,VY,VZ]``."""
    lib = _lib.load()
    dev, (F_, V, H, W, CP), fa, NW, (VX, VY, VZ), C = _window_args(
        hm, cam, fine_axes, tl, start, end, sub_cube, channels, win_frame, live)
    if out is None:
        out = torch.empty(NW, C, VX, VY, VZ, dtype=torch.float32, device=dev)
    else:
        _need(out, torch.float32, 'out', dev)
        if out.numel() != NW * C * VX * VY * VZ:
            raise ValueError('out has the wrong size')
    groups, ng = _groups_arg(groups, NW, dev)
    if clamp_mask is not None:  # training forward: also record the clamp mask
        _check_clamp_mask(clamp_mask, NW, VX * VY * VZ, CP, dev)
        with torch.cuda.device(dev):
            rc = lib.tempo_unproject_window_train(
                _ptr(hm), _ptr(cam), _ptr(win_frame), _ptr(fa[0]), _ptr(fa[1]), _ptr(fa[2]),
                _ptr(tl), _ptr(start), _ptr(end), _ptr(live), _ptr(out), NW, F_, V, C, CP, H,
                W, int(fa[0].numel()), int(fa[1].numel()), int(fa[2].numel()), VX, VY, VZ,
                float(heatmap_size[0]), float(heatmap_size[1]), float(image_size[0]),
                float(image_size[1]), numerics.abi_flag(mm_mode), _ptr(clamp_mask),
                _stream_ptr(dev))
        _lib.check(rc, 'tempo_unproject_window_train')
        return out
    with torch.cuda.device(dev):
        rc = lib.tempo_unproject_window_grouped(
            _ptr(hm), _ptr(cam), _ptr(win_frame), _ptr(fa[0]), _ptr(fa[1]), _ptr(fa[2]),
            _ptr(tl), _ptr(start), _ptr(end), _ptr(live), _ptr(out), NW, F_, V, C, CP, H,
            W, int(fa[0].numel()), int(fa[1].numel()), int(fa[2].numel()), VX, VY, VZ,
            float(heatmap_size[0]), float(heatmap_size[1]), float(image_size[0]),
            float(image_size[1]), numerics.abi_flag(mm_mode), _ptr(groups), ng,
            _stream_ptr(dev))
    _lib.check(rc, 'tempo_unproject_window_grouped')
    return out


def unproject_window_planes(hm, cam, fine_axes, tl, start, end, sub_cube, channels,
                            heatmap_size, image_size, win_frame=None, live=None,
                            want_cubes=False, mm_mode='auto', out=None, planes_out=None):
    """Window unprojection with the pose regressor's axis-max fused in
    (``tempo_pose.py:172-176``).  Arguments as :func:`unproject_window`.  Returns
    ``(cubes | None, plane_xy [NW,C,VX,VY], plane_xz [NW,C,VX,VZ],
    plane_yz [NW,C,VY,VZ])`` = ``torch.max`` of the volume over z, y, x; with
    ``want_cubes=False`` the volume is never written."""
    lib = _lib.load()
    dev, (F_, V, H, W, CP), fa, NW, (VX, VY, VZ), C = _window_args(
        hm, cam, fine_axes, tl, start, end, sub_cube, channels, win_frame, live)
    cubes = None
    if want_cubes:
        cubes = out if out is not None else torch.empty(NW, C, VX, VY, VZ,
                                                        dtype=torch.float32, device=dev)
        _need(cubes, torch.float32, 'out', dev)
        if cubes.numel() != NW * C * VX * VY * VZ:
            raise ValueError('out has the wrong size')
    shapes = [(NW, C, VX, VY), (NW, C, VX, VZ), (NW, C, VY, VZ)]
    if planes_out is None:
        planes = [torch.empty(sh, dtype=torch.float32, device=dev) for sh in shapes]
    else:
        planes = list(planes_out)
        for t, sh in zip(planes, shapes):
            _need(t, torch.float32, 'planes_out', dev)
            if tuple(t.shape) != sh:
                raise ValueError(f'planes_out: expected {sh}, got {tuple(t.shape)}')
    with torch.cuda.device(dev):
        rc = lib.tempo_unproject_window_planes(
            _ptr(hm), _ptr(cam), _ptr(win_frame), _ptr(fa[0]), _ptr(fa[1]), _ptr(fa[2]),
            _ptr(tl), _ptr(start), _ptr(end), _ptr(live), _ptr(cubes), _ptr(planes[0]),
            _ptr(planes[1]), _ptr(planes[2]), NW, F_, V, C, CP, H, W, int(fa[0].numel()),
            int(fa[1].numel()), int(fa[2].numel()), VX, VY, VZ, float(heatmap_size[0]),
            float(heatmap_size[1]), float(image_size[0]), float(image_size[1]),
            numerics.abi_flag(mm_mode), _stream_ptr(dev))
    _lib.check(rc, 'tempo_unproject_window_planes')
    return cubes, planes[0], planes[1], planes[2]


def project_points(pts, cam, heatmap_size, image_size, mm_mode='auto'):
    """pts ``[N,3]``, cam ``[M,32]`` -> (pix ``[M,N,2]``, grid ``[M,N,2]``,
    inside ``[M,N]`` uint8): the coordinate chain alone, for parity checks."""
    lib = _lib.load()
    dev = pts.device
    _need(pts, torch.float32, 'pts')
    _need(cam, torch.float32, 'cam', dev)
    N, M = int(pts.shape[0]), int(cam.shape[0])
    pix = torch.empty(M, N, 2, dtype=torch.float32, device=dev)
    grid = torch.empty(M, N, 2, dtype=torch.float32, device=dev)
    inside = torch.empty(M, N, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.tempo_project_points(
            _ptr(pts), _ptr(cam), M, N, float(heatmap_size[0]), float(heatmap_size[1]),
            float(image_size[0]), float(image_size[1]), numerics.abi_flag(mm_mode),
            _ptr(pix), _ptr(grid), _ptr(inside), _stream_ptr(dev))
    _lib.check(rc, 'tempo_project_points')
    return pix, grid, inside


# --------------------------------------------------------------------------
# heatmaps read in place (channel-last tensors)
# --------------------------------------------------------------------------

class MapLayout:
    """Where the forward kernels find the heatmaps: V base pointers (frame 0's map of each
    view), the byte distance between frames of one view, and between texels.  Keeps the
    tensors alive."""

    def __init__(self, tensors, ptrs, frame_bytes, texel_bytes, shape):
        self.tensors = tensors
        self.V = len(ptrs)
        self.ptrs = (ctypes.c_void_p * self.V)(*ptrs)
        self.frame_bytes = int(frame_bytes)
        self.texel_bytes = int(texel_bytes)
        self.F, self.H, self.W = shape
        self.device = tensors[0].device


def packed_layout(hm):
    """Layout of a repacked ``[F,V,H,W,CP]`` buffer (:func:`repack_nhwc`)."""
    _need(hm, torch.float32, 'hm')
    F_, V, H, W, CP = hm.shape
    base = hm.data_ptr()
    view = H * W * CP * 4
    return MapLayout([hm], [base + v * view for v in range(V)], V * view, CP * 4, (F_, H, W))


def is_channels_last(fm):
    """fp32 ``[B,C,H,W]`` tensor whose memory order is ``[B][H][W][C]``, densely packed."""
    if fm.dim() != 4 or fm.dtype != torch.float32:
        return False
    B, C, H, W = fm.shape
    return C > 1 and tuple(fm.stride()) == (H * W * C, 1, W * C, C)


def channels_last_layout(feature_maps, c_begin=0):
    """Layout of V ``torch.channels_last`` tensors ``[B,C_total,H,W]``, read in place
    (no repack); channels from ``c_begin`` on."""
    fm0 = feature_maps[0]
    B, C_total, H, W = fm0.shape
    dev = fm0.device
    for i, fm in enumerate(feature_maps):
        if not isinstance(fm, torch.Tensor) or not fm.is_cuda or fm.device != dev:
            raise TypeError(f'feature_maps[{i}]: expected a tensor on {dev}')
        if tuple(fm.shape) != (B, C_total, H, W) or not is_channels_last(fm):
            raise ValueError(f'feature_maps[{i}]: expected a channels_last fp32 '
                             f'[{B},{C_total},{H},{W}] tensor')
    return MapLayout(list(feature_maps), [fm.data_ptr() + 4 * c_begin for fm in feature_maps],
                     H * W * C_total * 4, C_total * 4, (B, H, W))


def unproject_root_views(layout, cam, axes, channels, heatmap_size, image_size, live=None,
                         want_grids=False, want_mask=False, out=None, mm_mode='auto',
                         zmax_out=None, want_cubes=True):
    """:func:`unproject_root` on heatmaps in place (any :class:`MapLayout`); ``zmax_out``
    ``[B,C,X,Y]`` also receives the volume's maximum over z (center_net.py:230)."""
    lib = _lib.load()
    dev = layout.device
    B, V, H, W = layout.F, layout.V, layout.H, layout.W
    _need(cam, torch.float32, 'cam', dev)
    if tuple(cam.shape) != (B, V, 32):
        raise ValueError(f'cam must be [{B},{V},32], got {tuple(cam.shape)}')
    ax = [_need(a, torch.float32, 'axes', dev) for a in axes]
    batched = ax[0].dim() == 2
    X, Y, Z = (int(a.shape[-1]) for a in ax)
    N = X * Y * Z
    C = int(channels)
    CP = padded_channels(C)
    if live is not None:
        _need(live, torch.uint8, 'live', dev)
    if not want_cubes:
        if zmax_out is None:
            raise ValueError('want_cubes=False needs zmax_out')
        out = None
    elif out is None:
        out = torch.empty(B, C, X, Y, Z, dtype=torch.float32, device=dev)
    else:
        _need(out, torch.float32, 'out', dev)
        if out.numel() != B * C * N:
            raise ValueError('out has the wrong size')
    grids = torch.empty(B, N, 3, dtype=torch.float32, device=dev) if want_grids else None
    mask, mask_bytes = None, 0
    if want_mask:
        mask_bytes, mdt = mask_dtype(V)
        mask = torch.empty(B, N, dtype=mdt, device=dev)
    head = (layout.ptrs, layout.frame_bytes, layout.texel_bytes, _ptr(cam), _ptr(ax[0]),
            _ptr(ax[1]), _ptr(ax[2]), int(batched), _ptr(live), _ptr(out), _ptr(grids),
            _ptr(mask), mask_bytes, B, V, C, CP, H, W, X, Y, Z, float(heatmap_size[0]),
            float(heatmap_size[1]), float(image_size[0]), float(image_size[1]),
            numerics.abi_flag(mm_mode))
    with torch.cuda.device(dev):
        if zmax_out is None:
            rc = lib.tempo_unproject_root_views(*head, _stream_ptr(dev))
        else:
            _need(zmax_out, torch.float32, 'zmax_out', dev)
            if tuple(zmax_out.shape) != (B, C, X, Y):
                raise ValueError(f'zmax_out must be [{B},{C},{X},{Y}]')
            rc = lib.tempo_unproject_root_views_zmax(*head, _ptr(zmax_out), _stream_ptr(dev))
    _lib.check(rc, 'tempo_unproject_root_views' if zmax_out is None
               else 'tempo_unproject_root_views_zmax')
    return out, grids, mask


def unproject_window_views(layout, cam, fine_axes, tl, start, end, sub_cube, channels,
                           heatmap_size, image_size, win_frame=None, live=None, out=None,
                           mm_mode='auto'):
    """:func:`unproject_window` on heatmaps in place (any :class:`MapLayout`)."""
    lib = _lib.load()
    dev = layout.device
    F_, V, H, W = layout.F, layout.V, layout.H, layout.W
    _need(cam, torch.float32, 'cam', dev)
    if tuple(cam.shape) != (F_, V, 32):
        raise ValueError(f'cam must be [{F_},{V},32], got {tuple(cam.shape)}')
    fa = [_need(a, torch.float32, 'fine_axes', dev) for a in fine_axes]
    NW = int(tl.shape[0])
    for name, t in (('tl', tl), ('start', start), ('end', end)):
        _need(t, torch.int32, name, dev)
        if tuple(t.shape) != (NW, 3):
            raise ValueError(f'{name} must be [{NW},3]')
    if win_frame is not None:
        _need(win_frame, torch.int32, 'win_frame', dev)
    elif NW != F_:
        raise ValueError('without win_frame there must be one window per frame')
    if live is not None:
        _need(live, torch.uint8, 'live', dev)
    VX, VY, VZ = (int(v) for v in sub_cube)
    C = int(channels)
    CP = padded_channels(C)
    if out is None:
        out = torch.empty(NW, C, VX, VY, VZ, dtype=torch.float32, device=dev)
    else:
        _need(out, torch.float32, 'out', dev)
        if out.numel() != NW * C * VX * VY * VZ:
            raise ValueError('out has the wrong size')
    with torch.cuda.device(dev):
        rc = lib.tempo_unproject_window_views(
            layout.ptrs, layout.frame_bytes, layout.texel_bytes, _ptr(cam), _ptr(win_frame),
            _ptr(fa[0]), _ptr(fa[1]), _ptr(fa[2]), _ptr(tl), _ptr(start), _ptr(end), _ptr(live),
            _ptr(out), NW, F_, V, C, CP, H, W, int(fa[0].numel()), int(fa[1].numel()),
            int(fa[2].numel()), VX, VY, VZ, float(heatmap_size[0]), float(heatmap_size[1]),
            float(image_size[0]), float(image_size[1]), numerics.abi_flag(mm_mode),
            _stream_ptr(dev))
    _lib.check(rc, 'tempo_unproject_window_views')
    return out


def unproject_window_planes_views(layout, cam, fine_axes, tl, start, end, sub_cube, channels,
                                  heatmap_size, image_size, win_frame=None, live=None,
                                  want_cubes=False, mm_mode='auto', out=None, planes_out=None):
    """:func:`unproject_window_planes` on heatmaps in place (any :class:`MapLayout`): the
    three axis-max planes without a repack and, with ``want_cubes=False``, without the
    volume.  Returns ``(cubes | None, plane_xy, plane_xz, plane_yz)``."""
    lib = _lib.load()
    dev = layout.device
    F_, V, H, W = layout.F, layout.V, layout.H, layout.W
    _need(cam, torch.float32, 'cam', dev)
    if tuple(cam.shape) != (F_, V, 32):
        raise ValueError(f'cam must be [{F_},{V},32], got {tuple(cam.shape)}')
    fa = [_need(a, torch.float32, 'fine_axes', dev) for a in fine_axes]
    NW = int(tl.shape[0])
    for name, t in (('tl', tl), ('start', start), ('end', end)):
        _need(t, torch.int32, name, dev)
        if tuple(t.shape) != (NW, 3):
            raise ValueError(f'{name} must be [{NW},3]')
    if win_frame is not None:
        _need(win_frame, torch.int32, 'win_frame', dev)
        if win_frame.numel() != NW:
            raise ValueError('win_frame must have one entry per window')
    elif NW != F_:
        raise ValueError('without win_frame there must be one window per frame')
    if live is not None:
        _need(live, torch.uint8, 'live', dev)
        if live.numel() != NW:
            raise ValueError('live must have one entry per window')
    VX, VY, VZ = (int(v) for v in sub_cube)
    C = int(channels)
    CP = padded_channels(C)
    cubes = None
    if want_cubes:
        cubes = out if out is not None else torch.empty(NW, C, VX, VY, VZ,
                                                        dtype=torch.float32, device=dev)
        _need(cubes, torch.float32, 'out', dev)
        if cubes.numel() != NW * C * VX * VY * VZ:
            raise ValueError('out has the wrong size')
    shapes = [(NW, C, VX, VY), (NW, C, VX, VZ), (NW, C, VY, VZ)]
    if planes_out is None:
        planes = [torch.empty(sh, dtype=torch.float32, device=dev) for sh in shapes]
    else:
        planes = list(planes_out)
        for t, sh in zip(planes, shapes):
            _need(t, torch.float32, 'planes_out', dev)
            if tuple(t.shape) != sh:
                raise ValueError(f'planes_out: expected {sh}, got {tuple(t.shape)}')
    with torch.cuda.device(dev):
        rc = lib.tempo_unproject_window_planes_views(
            layout.ptrs, layout.frame_bytes, layout.texel_bytes, _ptr(cam), _ptr(win_frame),
            _ptr(fa[0]), _ptr(fa[1]), _ptr(fa[2]), _ptr(tl), _ptr(start), _ptr(end), _ptr(live),
            _ptr(cubes), _ptr(planes[0]), _ptr(planes[1]), _ptr(planes[2]), NW, F_, V, C, CP,
            H, W, int(fa[0].numel()), int(fa[1].numel()), int(fa[2].numel()), VX, VY, VZ,
            float(heatmap_size[0]), float(heatmap_size[1]), float(image_size[0]),
            float(image_size[1]), numerics.abi_flag(mm_mode), _stream_ptr(dev))
    _lib.check(rc, 'tempo_unproject_window_planes_views')
    return cubes, planes[0], planes[1], planes[2]


def gather_zcolumns(cubes, index, out=None):
    """``cubes [B,C,X,Y,Z]``, ``index`` int64 ``[B,K]`` (cell = x * Y + y) ->
    ``[B,K,C,Z]``: the z-columns the root detector's 1-D stage reads
    (``tempo_detector.py:204-206``)."""
    lib = _lib.load()
    dev = cubes.device
    _no_grad_path('gather_zcolumns', cubes)
    _need(cubes, torch.float32, 'cubes')
    _need(index, torch.int64, 'index', dev)
    B, C, X, Y, Z = (int(v) for v in cubes.shape)
    if index.dim() != 2 or index.shape[0] != B:
        raise ValueError(f'index must be [{B},K]')
    K = int(index.shape[1])
    if out is None:
        out = torch.empty(B, K, C, Z, dtype=torch.float32, device=dev)
    else:
        _need(out, torch.float32, 'out', dev)
        if out.numel() != B * K * C * Z:
            raise ValueError('out has the wrong size')
    with torch.cuda.device(dev):
        rc = lib.tempo_gather_zcolumns(_ptr(cubes), _ptr(index), _ptr(out), B, K, C, X, Y, Z,
                                       _stream_ptr(dev))
    _lib.check(rc, 'tempo_gather_zcolumns')
    return out


def assemble_zslabs(slabs, out=None):
    """``slabs [R,B,C,X,Y,Zr]`` (what one all-gather of the ranks' z-slabs leaves) ->
    the volume ``[B,C,X,Y,R*Zr]`` (``tempo_assemble_zslabs``)."""
    lib = _lib.load()
    dev = slabs.device
    _need(slabs, torch.float32, 'slabs')
    R, B, C, X, Y, Zr = (int(v) for v in slabs.shape)
    if out is None:
        out = torch.empty(B, C, X, Y, R * Zr, dtype=torch.float32, device=dev)
    else:
        _need(out, torch.float32, 'out', dev)
        if out.numel() != slabs.numel():
            raise ValueError('out has the wrong size')
    with torch.cuda.device(dev):
        rc = lib.tempo_assemble_zslabs(_ptr(slabs), _ptr(out), R, B * C * X * Y, Zr,
                                       _stream_ptr(dev))
    _lib.check(rc, 'tempo_assemble_zslabs')
    return out


def _columns_common(cam, axes, index, B, V, channels, live, out, dev):
    _need(cam, torch.float32, 'cam', dev)
    if tuple(cam.shape) != (B, V, 32):
        raise ValueError(f'cam must be [{B},{V},32], got {tuple(cam.shape)}')
    ax = [_need(a, torch.float32, 'axes', dev) for a in axes]
    batched = ax[0].dim() == 2
    for a in ax:
        if (a.dim() == 2) != batched or (batched and a.shape[0] != B):
            raise ValueError('axes must be all [n] or all [B,n]')
    X, Y, Z = (int(a.shape[-1]) for a in ax)
    _need(index, torch.int64, 'index', dev)
    if index.dim() != 2 or index.shape[0] != B:
        raise ValueError(f'index must be int64 [{B},K]')
    K = int(index.shape[1])
    C = int(channels)
    if live is not None:
        _need(live, torch.uint8, 'live', dev)
        if live.numel() != B:
            raise ValueError('live must have B entries')
    if out is None:
        out = torch.empty(B, K, C, Z, dtype=torch.float32, device=dev)
    else:
        _need(out, torch.float32, 'out', dev)
        if out.numel() != B * K * C * Z:
            raise ValueError('out has the wrong size')
    return ax, batched, (X, Y, Z), K, C, out


def unproject_root_columns(maps, cam, axes, index, heatmap_size, image_size, channels=None,
                           c_begin=0, live=None, out=None, mm_mode='auto'):
    """The z-columns of the root volume at ``index`` int64 ``[B,K]`` (cell = x * Y + y)
    computed straight from the heatmaps -> ``[B,K,C,Z]``, bit-identical to
    :func:`gather_zcolumns` on the volume :func:`unproject_root` writes
    (``tempo_detector.py:204-206``), without that volume.  ``maps``: the repacked
    ``[B,V,H,W,CP]`` tensor (``channels`` = real channels), a :class:`MapLayout` of
    channel-last tensors in place, or the reference's list of V NCHW ``[B,C_total,H,W]``
    tensors (channels ``[c_begin, c_begin + channels)``)."""
    lib = _lib.load()
    if isinstance(maps, MapLayout):
        dev, B, V, H, W = maps.device, maps.F, maps.V, maps.H, maps.W
        if channels is None:
            raise ValueError('channels is required with a MapLayout')
        C = int(channels)
        CP = padded_channels(C)
        ax, batched, (X, Y, Z), K, C, out = _columns_common(cam, axes, index, B, V, C, live,
                                                            out, dev)
        head = (maps.ptrs, maps.frame_bytes, maps.texel_bytes)
        fn, name = lib.tempo_unproject_root_columns_views, 'tempo_unproject_root_columns_views'
    elif isinstance(maps, torch.Tensor):
        dev = maps.device
        _need(maps, torch.float32, 'hm')
        B, V, H, W, CP = (int(v) for v in maps.shape)
        if channels is None:
            raise ValueError('channels is required with a repacked tensor')
        ax, batched, (X, Y, Z), K, C, out = _columns_common(cam, axes, index, B, V, channels,
                                                            live, out, dev)
        if padded_channels(C) != CP:
            raise ValueError('hm.shape[-1] must be padded_channels(channels)')
        head = (_ptr(maps),)
        fn, name = lib.tempo_unproject_root_columns, 'tempo_unproject_root_columns'
    else:
        fm0 = maps[0]
        dev = fm0.device
        B, C_total, H, W = (int(v) for v in fm0.shape)
        V = len(maps)
        C = C_total - c_begin if channels is None else int(channels)
        CP = padded_channels(C)
        keep = []
        for i, fm in enumerate(maps):
            if tuple(fm.shape) != (B, C_total, H, W):
                raise ValueError('feature maps of all views must share one shape')
            if fm.dtype != torch.float32:
                fm = fm.float()
            keep.append(_need(fm.contiguous(), torch.float32, f'feature_maps[{i}]', dev))
        ax, batched, (X, Y, Z), K, C, out = _columns_common(cam, axes, index, B, V, C, live,
                                                            out, dev)
        ptrs = (ctypes.c_void_p * V)(*[fm.data_ptr() for fm in keep])
        head = (ptrs, C_total, int(c_begin))
        fn, name = lib.tempo_root_step_columns, 'tempo_root_step_columns'
    with torch.cuda.device(dev):
        rc = fn(*head, _ptr(cam), _ptr(ax[0]), _ptr(ax[1]), _ptr(ax[2]), int(batched),
                _ptr(live), _ptr(index), _ptr(out), B, K, V, C, CP, H, W, X, Y, Z,
                float(heatmap_size[0]), float(heatmap_size[1]), float(image_size[0]),
                float(image_size[1]), numerics.abi_flag(mm_mode), _stream_ptr(dev))
    _lib.check(rc, name)
    return out


def conv1x1_nhwc(x, weight, bias, num_views, timesteps=1, out=None):
    """The 2-D head's 1x1 convolution (``rnn_multiview_pose.py:61-66, :321``) writing the
    unprojection's channel-last layout directly.

    ``x`` fp32 ``[T*V*B, K, H, W]`` (images ordered (t, v, b), as the reference stacks them),
    ``weight`` ``[C, K]`` or ``[C, K, 1, 1]``, ``bias`` ``[C]`` or None.  Returns
    ``[T*B, V, H, W, CP]``: ``out[t*B:(t+1)*B]`` is the repacked ``hm`` of timestep t (what
    :func:`repack_nhwc` would produce from ``feature_maps[t]``)."""
    lib = _lib.load()
    dev = x.device
    _no_grad_path('conv1x1_nhwc', x, weight, bias)
    _need(x, torch.float32, 'x')
    M, K, H, W = (int(v) for v in x.shape)
    V, T = int(num_views), int(timesteps)
    if M % (V * T):
        raise ValueError('x.shape[0] must be timesteps * num_views * batch')
    B = M // (V * T)
    w2 = weight.reshape(weight.shape[0], -1)
    _need(w2, torch.float32, 'weight', dev)
    C = int(w2.shape[0])
    if int(w2.shape[1]) != K:
        raise ValueError(f'weight must be [C,{K}]')
    if bias is not None:
        _need(bias, torch.float32, 'bias', dev)
        if bias.numel() != C:
            raise ValueError('bias must have C entries')
    CP = padded_channels(C)
    if out is None:
        out = torch.empty(T * B, V, H, W, CP, dtype=torch.float32, device=dev)
    else:
        _need(out, torch.float32, 'out', dev)
        if out.numel() != T * B * V * H * W * CP:
            raise ValueError('out has the wrong size')
    with torch.cuda.device(dev):
        rc = lib.tempo_conv1x1_nhwc(_ptr(x), _ptr(w2), _ptr(bias), _ptr(out), T, V, B, K, C, CP,
                                    H, W, _stream_ptr(dev))
    _lib.check(rc, 'tempo_conv1x1_nhwc')
    return out


def window_geometry(grid_center, scale, bias, space_size, sub_space_size, fine_bins, sub_cube,
                    want_live=True):
    """``project_layer.py:299-324`` for all windows of a call in one launch
    (``tempo_window_geometry``): returns ``(centers_tl, offset, start, end, live | None)``,
    bit-identical to the reference's tensor ops.  ``grid_center`` fp32 ``[B, 3 | >=7]`` on the
    device; ``scale`` / ``bias`` / ``space_size`` / ``sub_space_size`` device fp32 ``[3]``."""
    lib = _lib.load()
    dev = grid_center.device
    _need(grid_center, torch.float32, 'grid_center')
    B, cols = grid_center.shape
    for name, t in (('scale', scale), ('bias', bias), ('space_size', space_size),
                    ('sub_space_size', sub_space_size)):
        _need(t, torch.float32, name, dev)
        if t.numel() != 3:
            raise ValueError(f'{name} must have 3 entries')
    tl = torch.empty(B, 3, dtype=torch.int32, device=dev)
    start = torch.empty_like(tl)
    end = torch.empty_like(tl)
    offset = torch.empty(B, 3, dtype=torch.float32, device=dev)
    live = torch.empty(B, dtype=torch.uint8, device=dev) if (want_live and cols != 3) else None
    fx, fy, fz = (int(v) for v in fine_bins)
    vx, vy, vz = (int(v) for v in sub_cube)
    with torch.cuda.device(dev):
        rc = lib.tempo_window_geometry(_ptr(grid_center), B, cols, _ptr(scale), _ptr(bias),
                                       _ptr(space_size), _ptr(sub_space_size), fx, fy, fz, vx,
                                       vy, vz, _ptr(tl), _ptr(start), _ptr(end), _ptr(offset),
                                       _ptr(live), _stream_ptr(dev))
    _lib.check(rc, 'tempo_window_geometry')
    return tl, offset, start, end, live
