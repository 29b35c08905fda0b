"""Drop-in ``ProjectLayer`` / ``ProjectLayerWithMask`` backed by the HIP kernels.

Same constructor arguments, ``forward`` signatures, return tuples and public
attributes as ``tempo/models/project_layer.py`` of the reference (``:18-158``
and ``:160-388``), so ``tempo_detector.py:47,187-190`` and
``tempo_pose.py:41-46,152-162`` can use these classes unchanged.

Differences, all deliberate:

* HIP devices only.  There is no CPU / PyTorch fallback: CPU tensors raise.
* Differentiable with respect to the feature maps only (cameras, centres and
  grids are constants, as in the reference's training): the backward pass is a
  HIP kernel as well (``csrc``: ``unproject_backward_kernel``).
* ``agg_method`` must be ``'mean'`` (what every config uses; the reference's
  ``'learned'`` branch is hard-wired to 5 views x 32 channels in ``ProjectLayer``
  and references undefined names in ``ProjectLayerWithMask``, ``:360-366``).
* camera dicts with ``'intrinsics'`` raise ``NotImplementedError`` (the
  reference would hit a ``NameError``: ``FisheyeCameraTorch`` is undefined).
* an error in sampling raises instead of dropping into ``pdb`` (``:285-286``).
"""
import warnings

import torch
import torch.nn as nn

from . import ops
from .camera import pack_cameras, to_device


def _pair(v):
    return [v, v] if isinstance(v, int) else v


def _triple(v, kinds):
    return [v, v, v] if isinstance(v, kinds) else v


class PackedHeatmaps:
    """The V per-view feature maps of one timestep, already in the kernels' channel-last
    layout: ``hm [B,V,H,W,CP]`` with ``channels`` real channels (what
    :class:`ProcessLayerNHWC` / ``tempo_conv1x1_nhwc`` writes).  Pass it to the project
    layers where the reference passes ``feature_maps[t]``: the repack is skipped.  For
    other consumers it still behaves like the reference's list: ``len()`` = V and ``[v]``
    is a ``[B,C,H,W]`` view (strided, no copy)."""

    def __init__(self, hm, channels):
        if hm.dim() != 5 or hm.dtype != torch.float32 or not hm.is_contiguous():
            raise ValueError('PackedHeatmaps needs a contiguous fp32 [B,V,H,W,CP] tensor')
        if not 0 < channels <= hm.shape[-1] or hm.shape[-1] != ops.padded_channels(channels):
            raise ValueError('the last dimension must be padded_channels(channels)')
        self.hm = hm
        self.channels = int(channels)

    def __len__(self):
        return int(self.hm.shape[1])

    def __getitem__(self, v):
        return self.hm[:, v, :, :, :self.channels].permute(0, 3, 1, 2)

    def __iter__(self):
        return (self[v] for v in range(len(self)))


class ProcessLayerNHWC(nn.Module):
    """The reference's ``process_layer = nn.Conv2d(256, num_features, 1)``
    (``rnn_multiview_pose.py:61-66``) with the unprojection's layout as its output: same
    parameters (``weight [C,K,1,1]``, ``bias [C]``: state dicts are interchangeable),
    ``forward`` returns what ``:324-326`` builds -- one entry per timestep -- as
    :class:`PackedHeatmaps` instead of lists of NCHW tensors.  Inference only: a training
    step goes through ``F.conv2d`` (autograd) and returns the reference's lists."""

    def __init__(self, in_channels=256, out_channels=32):
        super().__init__()
        conv = nn.Conv2d(in_channels, out_channels, kernel_size=1)  # the reference's init
        self.weight, self.bias = conv.weight, conv.bias

    def forward(self, backbone_map, num_views, timesteps=1):
        if torch.is_grad_enabled() and (backbone_map.requires_grad or self.weight.requires_grad):
            out = torch.nn.functional.conv2d(backbone_map, self.weight, self.bias)
            return [list(torch.chunk(m, num_views)) for m in torch.chunk(out, timesteps)]
        hm = ops.conv1x1_nhwc(backbone_map.float().contiguous(), self.weight.detach(),
                              self.bias.detach(), num_views, timesteps)
        B = hm.shape[0] // timesteps
        C = int(self.weight.shape[0])
        return [PackedHeatmaps(hm[t * B:(t + 1) * B], C) for t in range(timesteps)]


def _packed_sequence(feature_maps_t):
    """``[T*B,V,H,W,CP]`` view when the timesteps are :class:`PackedHeatmaps` that follow
    each other in one buffer (what :class:`ProcessLayerNHWC` returns), else None."""
    if not all(isinstance(f, PackedHeatmaps) for f in feature_maps_t):
        return None
    first = feature_maps_t[0].hm
    step = first.numel() * first.element_size()
    for t, f in enumerate(feature_maps_t):
        if (f.hm.shape != first.shape or f.channels != feature_maps_t[0].channels or
                f.hm.untyped_storage().data_ptr() != first.untyped_storage().data_ptr() or
                f.hm.data_ptr() != first.data_ptr() + t * step):
            return None
    T = len(feature_maps_t)
    return torch.as_strided(first, (T * first.shape[0],) + tuple(first.shape[1:]), first.stride())


def _map_source(feature_maps):
    """How the forward kernels get at the maps: ('packed', PackedHeatmaps),
    ('views', list of channels_last tensors read in place) or ('nchw', list to repack)."""
    if isinstance(feature_maps, PackedHeatmaps):
        return 'packed'
    if all(ops.is_channels_last(fm) and fm.is_cuda for fm in feature_maps):
        return 'views'
    return 'nchw'


def _check_inputs(feature_maps):
    if isinstance(feature_maps, PackedHeatmaps):
        fm0 = feature_maps.hm
    else:
        fm0 = feature_maps[0]
    if not fm0.is_cuda:
        raise RuntimeError('tempo_amd.ProjectLayer runs on HIP devices only; got a '
                           f'{fm0.device} tensor (there is no CPU fallback)')
    return fm0.device


def _requires_grad(feature_maps):
    """autograd is recording and some input map asks for a gradient"""
    if not torch.is_grad_enabled():
        return False
    if isinstance(feature_maps, PackedHeatmaps):
        return bool(feature_maps.hm.requires_grad)
    return any(fm.requires_grad for fm in feature_maps)


def _wants_grad(feature_maps):
    """True when the call must go through the differentiable path.  Packed maps have none
    (the backward writes per-view NCHW gradients): asking for one raises instead of
    silently returning a volume that is cut off from the graph."""
    if isinstance(feature_maps, PackedHeatmaps):
        if _requires_grad(feature_maps):
            raise NotImplementedError(
                'PackedHeatmaps carry no gradient: pass the per-view [B,C,H,W] tensors '
                '(e.g. list(packed)) for a training step, or call under torch.no_grad()')
        return False
    return _requires_grad(feature_maps)


def _batch_channels(feature_maps):
    if isinstance(feature_maps, PackedHeatmaps):
        return int(feature_maps.hm.shape[0]), feature_maps.channels
    return int(feature_maps[0].shape[0]), int(feature_maps[0].shape[1])


def _alert_not_deterministic():
    """The backward kernels add each texel's contributions with fp32 atomics, in no fixed
    order: last-bit differences between runs.  So does the op they replace --
    ``grid_sampler_2d_backward_cuda`` of ``F.grid_sample`` (project_layer.py:128-133, :278-284)
    -- and torch reports it under ``torch.use_deterministic_algorithms``: an error, or a warning
    with ``warn_only=True``.  Same behaviour here."""
    if not torch.are_deterministic_algorithms_enabled():
        return
    msg = ('tempo_amd unprojection backward does not have a deterministic implementation '
           '(fp32 atomic adds into the heatmap gradient), like grid_sampler_2d_backward_cuda '
           'which it replaces; torch.use_deterministic_algorithms(True, warn_only=True) turns '
           'this error into a warning')
    if torch.is_deterministic_algorithms_warn_only_enabled():
        warnings.warn(msg)
    else:
        raise RuntimeError(msg)


class _Unproject(torch.autograd.Function):
    """Differentiable unprojection (root or window): the forward records the clamp
    mask, the backward scatters the volume's gradient into the heatmaps
    (``tempo_unproject_*_backward``) -- what autograd does through
    ``project_layer.py:128-151`` / ``:278-378`` of the reference.  ``meta`` holds the
    constant inputs (cameras, geometry); grids and the valid-view bitmap come back
    through it, they carry no gradient."""

    @staticmethod
    def forward(ctx, meta, *feature_maps):
        dev = feature_maps[0].device
        B, C, H, W = feature_maps[0].shape
        V = len(feature_maps)
        window = meta['kind'] == 'window'
        items = meta['tl'].shape[0] if window else B
        shape = meta['sub_cube'] if window else meta['cube_size']
        n_vox = shape[0] * shape[1] * shape[2]
        cubes = torch.empty(items, C, *shape, dtype=torch.float32, device=dev)
        ctx.meta, ctx.masks, ctx.fm_shape, ctx.V = meta, [], (B, C, H, W), V
        for c0, cs in _Slices(C).items:
            hm = ops.repack_nhwc(feature_maps, c0, cs)
            cmask = ops.new_clamp_mask(items, n_vox, hm.shape[-1], dev)
            whole = cs == C
            if window:
                out = ops.unproject_window(
                    hm, meta['cam'], meta['fine_axes'], meta['tl'], meta['start'],
                    meta['end'], meta['sub_cube'], cs, meta['heatmap_size'],
                    meta['image_size'], live=meta['live'], out=cubes if whole else None,
                    mm_mode=meta['mm_mode'], clamp_mask=cmask)
            else:
                first = c0 == 0
                out, g, m = ops.unproject_root(
                    hm, meta['cam'], meta['axes'], cs, meta['heatmap_size'],
                    meta['image_size'], live=meta['live'],
                    want_grids=first and meta['want_grids'],
                    want_mask=first and meta['want_mask'], out=cubes if whole else None,
                    mm_mode=meta['mm_mode'], clamp_mask=cmask)
                if first:
                    meta['grids'], meta['mask'] = g, m
            if not whole:
                cubes[:, c0:c0 + cs] = out
            ctx.masks.append((c0, cs, cmask, tuple(hm.shape)))
        return cubes

    @staticmethod
    def backward(ctx, grad_cubes):
        _alert_not_deterministic()
        meta = ctx.meta
        B, C, H, W = ctx.fm_shape
        dev = grad_cubes.device
        grads = [torch.zeros(B, C, H, W, dtype=torch.float32, device=dev)
                 for _ in range(ctx.V)]
        window = meta['kind'] == 'window'
        for c0, cs, cmask, hm_shape in ctx.masks:
            g = grad_cubes[:, c0:c0 + cs].float().contiguous()
            if window:
                ghm = ops.unproject_window_backward(
                    g, meta['cam'], meta['fine_axes'], meta['tl'], meta['start'], meta['end'],
                    cmask, hm_shape, meta['heatmap_size'], meta['image_size'],
                    live=meta['live'], mm_mode=meta['mm_mode'],
                    tile_texels=meta.get('tile_texels', 0))
            else:
                ghm = ops.unproject_root_backward(
                    g, meta['cam'], meta['axes'], cmask, hm_shape, meta['heatmap_size'],
                    meta['image_size'], live=meta['live'], mm_mode=meta['mm_mode'])
            ops.unrepack_nchw(ghm, C, c0, cs, out=grads)
        need = ctx.needs_input_grad[1:]
        return (None,) + tuple(g if n else None for g, n in zip(grads, need))


class _UnprojectPlanes(torch.autograd.Function):
    """Differentiable fused window planes: forward = the inference kernel (no volume) plus the
    claim pass that records which voxel ``torch.max`` would route each plane entry's gradient
    to; backward = the window backward kernel fed from the three plane gradients through
    those claims.  What autograd does through ``project_layer.py:278-378`` followed by the
    three ``torch.max`` of ``tempo_pose.py:172-176`` -- without the ``[B,C,X,Y,Z]`` volume
    or its gradient volume (config 3's shape, 320 windows: 48 ms through the volume)."""

    @staticmethod
    def forward(ctx, meta, *feature_maps):
        """``feature_maps``: T x V tensors ``[B,C,H,W]``, timestep-major (T = ``meta['T']``,
        default 1); frames are numbered ``t * B + b``; ``meta['win_frame']`` maps windows to
        frames (None: window i looks at frame i)."""
        dev = feature_maps[0].device
        B, C, H, W = feature_maps[0].shape
        T = int(meta.get('T', 1))
        V = len(feature_maps) // T
        NW = meta['tl'].shape[0]
        vx, vy, vz = meta['sub_cube']
        planes = [torch.empty(NW, C, *sh, dtype=torch.float32, device=dev)
                  for sh in ((vx, vy), (vx, vz), (vy, vz))]
        ctx.meta, ctx.saved, ctx.fm_shape, ctx.TV = meta, [], (B, C, H, W), (T, V)
        for c0, cs in _Slices(C).items:
            if T == 1:
                hm = ops.repack_nhwc(feature_maps, c0, cs)
            else:
                hm = torch.empty(T * B, V, H, W, ops.padded_channels(cs), dtype=torch.float32,
                                 device=dev)
                for t in range(T):
                    ops.repack_nhwc(feature_maps[t * V:(t + 1) * V], c0, cs,
                                    out=hm[t * B:(t + 1) * B])
            args = (hm, meta['cam'], meta['fine_axes'], meta['tl'], meta['start'], meta['end'],
                    meta['sub_cube'], cs)
            kw = dict(win_frame=meta.get('win_frame'), live=meta['live'],
                      mm_mode=meta['mm_mode'])
            whole = cs == C
            out = ops.unproject_window_planes(
                *args, meta['heatmap_size'], meta['image_size'], want_cubes=False,
                planes_out=planes if whole else None, **kw)
            claims = ops.window_planes_claims(*args, out[1:], meta['heatmap_size'],
                                              meta['image_size'], **kw)
            if not whole:
                for dst, src in zip(planes, out[1:]):
                    dst[:, c0:c0 + cs] = src
            ctx.saved.append((c0, cs, claims, tuple(hm.shape)))
        return tuple(planes)

    @staticmethod
    def backward(ctx, g_xy, g_xz, g_yz):
        _alert_not_deterministic()
        meta = ctx.meta
        B, C, H, W = ctx.fm_shape
        T, V = ctx.TV
        dev = g_xy.device
        grads = [torch.zeros(B, C, H, W, dtype=torch.float32, device=dev)
                 for _ in range(T * V)]
        for c0, cs, claims, hm_shape in ctx.saved:
            gp = [g[:, c0:c0 + cs].float().contiguous() for g in (g_xy, g_xz, g_yz)]
            ghm = ops.unproject_window_planes_backward(
                gp, claims, meta['cam'], meta['fine_axes'], meta['tl'], meta['start'],
                meta['end'], hm_shape, meta['heatmap_size'], meta['image_size'],
                win_frame=meta.get('win_frame'), live=meta['live'], mm_mode=meta['mm_mode'],
                tile_texels=meta.get('tile_texels', 0))
            for t in range(T):
                ops.unrepack_nchw(ghm[t * B:(t + 1) * B], C, c0, cs,
                                  out=grads[t * V:(t + 1) * V])
        need = ctx.needs_input_grad[1:]
        return (None,) + tuple(g if n else None for g, n in zip(grads, need))


def _base_axis(size, bins):
    """``torch.linspace(-size/2, size/2, bins)`` exactly as compute_grid calls it
    (on the CPU: torch's CPU linspace is the oracle's arithmetic)."""
    return torch.linspace(-size / 2, size / 2, bins)


class _Slices:
    """channel slices of at most 32 (one kernel call each)"""

    def __init__(self, channels):
        self.items = [(c0, min(32, channels - c0)) for c0 in range(0, channels, 32)]


class ProjectLayer(nn.Module):
    """Root-grid unprojection, ``project_layer.py:18-158``."""

    def __init__(self, image_size, heatmap_size, agg_method='mean'):
        super().__init__()
        if agg_method != 'mean':
            raise NotImplementedError(f"agg_method={agg_method!r}: only 'mean' is supported")
        self.image_size = _pair(image_size)
        self.heatmap_size = _pair(heatmap_size)
        self.agg_method = agg_method
        self.mm_mode = 'auto'  # rounding flavour of the K=3 matmuls (numerics.py)
        self.last_mask = None  # valid-view bitmap of the last call when keep_mask
        self.keep_mask = False
        self.materialize_grids = True  # the reference's 2nd return; callers drop it
        # extension: keep torch.max(cubes, dim=4) of the last (inference) call in
        # ``last_zmax`` -- what CenterNet computes first (center_net.py:230)
        self.keep_zmax = False
        self.last_zmax = None
        self._axes_cache = {}
        self._train_volume = None  # see _training_volume
        # batch items with bit-identical camera records and grid centre (frames of
        # one recording) share one projection per voxel; results do not change
        # (opt-in: on MI355X the root kernel is bound by the texel gather, not by the
        # projection arithmetic, and the grouped launch measured slower -- DESIGN.md)
        self.share_projection = False

    def compute_grid(self, box_size, box_center, num_bins, device=None):
        """``project_layer.py:43-63``: ``[N,3]`` voxel centres, z fastest."""
        box_size = _triple(box_size, (int, float))
        num_bins = _triple(num_bins, int)
        ax = [(_base_axis(box_size[i], num_bins[i]) + box_center[i]).to(device)
              for i in range(3)]
        gx, gy, gz = torch.meshgrid(*ax, indexing='ij')
        return torch.stack([gx.reshape(-1), gy.reshape(-1), gz.reshape(-1)], dim=1)

    def _axes(self, grid_size, grid_center, cube_size, B, device):
        grid_size = _triple(grid_size, (int, float))
        cube_size = _triple(cube_size, int)
        if len(grid_center) == 1:
            # one shared centre (the detector's call): the three axis vectors depend on
            # a handful of numbers, keep them on the device between calls
            gc = grid_center[0]
            key = (tuple(float(v) for v in grid_size), tuple(float(gc[i]) for i in range(3)),
                   tuple(cube_size), str(device))
            hit = self._axes_cache.get(key)
            if hit is None:
                hit = [(_base_axis(grid_size[i], cube_size[i]) + float(gc[i])).to(device)
                       for i in range(3)]
                if len(self._axes_cache) >= 32:
                    self._axes_cache.clear()
                self._axes_cache[key] = hit
            return list(hit), list(cube_size)
        base = [_base_axis(grid_size[i], cube_size[i]) for i in range(3)]
        if isinstance(grid_center, torch.Tensor):
            gc = grid_center[:, :3].to(device=device, dtype=torch.float32)
        else:
            gc = torch.tensor([[float(c[i]) for i in range(3)] for c in grid_center],
                              dtype=torch.float32, device=device)
        if gc.shape[0] != B:
            raise IndexError('grid_center needs 1 or batch_size entries')
        return [(base[i].to(device)[None, :] + gc[:, i:i + 1]).contiguous()
                for i in range(3)], cube_size

    def _live(self, grid_center, B, device):
        """skip rule of ``project_layer.py:81``."""
        if len(grid_center[0]) == 3:
            return None
        if isinstance(grid_center, torch.Tensor):
            return (grid_center[:B, 3] >= 0).to(device=device, dtype=torch.uint8).contiguous()
        flags = [1 if float(grid_center[b][3]) >= 0 else 0 for b in range(B)]
        return None if all(flags) else torch.tensor(flags, dtype=torch.uint8, device=device)

    def _share_groups(self, cam_host, grid_center, B):
        """Runs of consecutive batch items whose packed camera records and grid
        centre are bit-identical (host-side comparison, no device sync)."""
        if not self.share_projection or B < 2 or cam_host.is_cuda:
            return None
        if len(grid_center) == 1:
            extra = None
        elif isinstance(grid_center, torch.Tensor):
            if grid_center.is_cuda:
                return None  # comparing centres would need a device sync
            extra = [tuple(float(v) for v in grid_center[b, :3]) for b in range(B)]
        else:
            extra = [tuple(float(grid_center[b][i]) for i in range(3)) for b in range(B)]
        return ops.share_groups(ops.camera_keys(cam_host, extra))

    def get_voxel(self, feature_maps, grid_size, grid_center, cube_size, centers,
                  scales, cameras, zslab=None):
        """``zslab=(z0, z1)`` (extension, used by ``tempo_amd.parallel``): compute
        only planes ``[z0, z1)`` of the grid; returns ``[B,C,X,Y,z1-z0]``."""
        device = _check_inputs(feature_maps)
        B, C = _batch_channels(feature_maps)
        live = self._live(grid_center, B, device)
        axes, cube_size = self._axes(grid_size, grid_center, cube_size, B, device)
        if zslab is not None:
            z0, z1 = zslab
            if not 0 <= z0 < z1 <= cube_size[2]:
                raise ValueError(f'zslab {zslab} outside [0, {cube_size[2]}]')
            axes[2] = axes[2][..., z0:z1].contiguous()
            cube_size = [cube_size[0], cube_size[1], z1 - z0]
        cam_host = pack_cameras(centers, scales, cameras, self.image_size, self.mm_mode)
        cam = to_device(cam_host, device)
        groups = self._share_groups(cam_host, grid_center, B)
        if groups is not None:
            groups = groups.to(device)
        if _wants_grad(feature_maps):  # training: differentiable path
            meta = dict(kind='root', cam=cam, axes=axes, cube_size=list(cube_size), live=live,
                        heatmap_size=self.heatmap_size, image_size=self.image_size,
                        mm_mode=self.mm_mode, want_grids=self.materialize_grids,
                        want_mask=self.keep_mask)
            cubes = _Unproject.apply(meta, *[fm.float() for fm in feature_maps])
            self.last_mask = meta.get('mask')
            self.last_zmax = None  # (the training forward does not emit the z-max planes)
            return cubes, meta.get('grids')
        cubes = torch.empty(B, C, *cube_size, dtype=torch.float32, device=device)
        grids = mask = None
        source = _map_source(feature_maps)
        if source in ('packed', 'views'):
            # maps already channel-last: read in place, no repack
            self.last_zmax = None
            return self._get_voxel_in_place(feature_maps, source, cam, axes, live, cubes)
        zmax = (torch.empty(B, C, cube_size[0], cube_size[1], dtype=torch.float32,
                            device=device) if self.keep_zmax and groups is None else None)
        for c0, cs in _Slices(C).items:
            first = c0 == 0
            if groups is None:  # frames in cache-sized chunks (ops.root_step)
                zs = None if zmax is None else (zmax if cs == C else torch.empty(
                    B, cs, cube_size[0], cube_size[1], dtype=torch.float32, device=device))
                out, g, m = ops.root_step(
                    feature_maps, cam, axes, self.heatmap_size, self.image_size, c0, cs,
                    live=live, want_grids=first and self.materialize_grids,
                    want_mask=first and self.keep_mask, out=cubes if cs == C else None,
                    mm_mode=self.mm_mode, zmax_out=zs)
                if zs is not None and cs != C:
                    zmax[:, c0:c0 + cs] = zs
            else:
                hm = ops.repack_nhwc(feature_maps, c0, cs)
                out, g, m = ops.unproject_root(
                    hm, cam, axes, cs, self.heatmap_size, self.image_size, live=live,
                    want_grids=first and self.materialize_grids,
                    want_mask=first and self.keep_mask,
                    out=cubes if cs == C else None, mm_mode=self.mm_mode, groups=groups)
            if cs != C:
                cubes[:, c0:c0 + cs] = out
            if first:
                grids, mask = g, m
        self.last_mask = mask
        self.last_zmax = zmax
        return cubes, grids

    def _get_voxel_in_place(self, feature_maps, source, cam, axes, live, cubes, plain=False):
        """Maps read where they lie; ``plain``: the volume only (no grids, mask, z-max; the
        module's ``last_*`` attributes are left alone)."""
        B, C = cubes.shape[:2]
        batched = axes[0].dim() == 2
        grids = mask = None
        zmax = None
        if self.keep_zmax and not plain:
            zmax = torch.empty(B, C, cubes.shape[2], cubes.shape[3], dtype=torch.float32,
                               device=cubes.device)
        chunk = B  # maps read where they lie: one launch over the batch is fastest
        for c0, cs in _Slices(C).items:
            first = c0 == 0
            out = cubes if cs == C else torch.empty(B, cs, *cubes.shape[2:], dtype=torch.float32,
                                                    device=cubes.device)
            zs = None if zmax is None else (zmax if cs == C else torch.empty(
                B, cs, *cubes.shape[2:4], dtype=torch.float32, device=cubes.device))
            gl, ml = [], []
            for b0 in range(0, B, chunk):
                sl = slice(b0, min(B, b0 + chunk))
                ax = [a[sl] for a in axes] if batched else axes
                lv = None if live is None else live[sl]
                kw = dict(live=lv, want_grids=first and self.materialize_grids and not plain,
                          want_mask=first and self.keep_mask and not plain, out=out[sl],
                          mm_mode=self.mm_mode)
                if source == 'packed':  # (at most 32 channels: one slice)
                    _, g, m = ops.unproject_root(
                        feature_maps.hm[sl], cam[sl], ax, cs, self.heatmap_size, self.image_size,
                        zmax_out=None if zs is None else zs[sl], **kw)
                else:
                    lay = ops.channels_last_layout([fm[sl] for fm in feature_maps], c0)
                    _, g, m = ops.unproject_root_views(
                        lay, cam[sl], ax, cs, self.heatmap_size, self.image_size,
                        zmax_out=None if zs is None else zs[sl], **kw)
                gl.append(g)
                ml.append(m)
            if cs != C:
                cubes[:, c0:c0 + cs] = out
                if zs is not None:
                    zmax[:, c0:c0 + cs] = zs
            if first:
                grids = torch.cat(gl, 0) if gl[0] is not None else None
                mask = torch.cat(ml, 0) if ml[0] is not None else None
        if plain:
            return cubes, None
        self.last_mask = mask
        self.last_zmax = zmax
        return cubes, grids

    def forward(self, feature_maps, grid_size, grid_center, cube_size, centers,
                scales, cameras):
        return self.get_voxel(feature_maps, grid_size, grid_center, cube_size,
                              centers, scales, cameras)

    def _consumer_inputs(self, feature_maps, grid_size, grid_center, cube_size, centers, scales,
                         cameras, what):
        device = _check_inputs(feature_maps)
        B, C = _batch_channels(feature_maps)
        live = self._live(grid_center, B, device)
        axes, cube_size = self._axes(grid_size, grid_center, cube_size, B, device)
        cam = to_device(pack_cameras(centers, scales, cameras, self.image_size, self.mm_mode),
                        device)
        return device, B, C, live, axes, cube_size, cam

    def _training_volume(self, feature_maps, grid_size, grid_center, cube_size, centers, scales,
                         cameras):
        """The differentiable volume for :meth:`forward_zmax` / :meth:`zcolumns` under
        autograd.  The detector calls them one after the other on the same inputs (z-max ->
        CenterNet -> top-K cells -> z-columns, ``tempo_detector.py:193-206``): the second call
        reuses the first one's volume -- same tensor objects at the same versions, same
        geometry, same packed camera table -- instead of unprojecting (and later
        back-propagating) twice.  One volume is kept alive until the next call."""
        table = pack_cameras(centers, scales, cameras, self.image_size, self.mm_mode)
        key = (tuple((id(fm), fm._version) for fm in feature_maps), id(table), repr(grid_size),
               repr([list(map(float, c)) for c in grid_center]) if not isinstance(
                   grid_center, torch.Tensor) else (id(grid_center), grid_center._version),
               repr(cube_size), self.mm_mode if isinstance(self.mm_mode, str) else
               repr(self.mm_mode))
        hit = getattr(self, '_train_volume', None)
        if hit is not None and hit[0] == key and hit[1] is table and all(
                a is b for a, b in zip(hit[2], feature_maps)):
            return hit[3]
        cubes, _ = self.get_voxel(feature_maps, grid_size, grid_center, cube_size, centers,
                                  scales, cameras)
        # (the tensors are kept with the key: an id can only be reused once its owner is gone)
        self._train_volume = (key, table, list(feature_maps), cubes)
        return cubes

    def forward_zmax(self, feature_maps, grid_size, grid_center, cube_size, centers, scales,
                     cameras):
        """Extension: ``torch.max(forward(...)[0], dim=4)[0]`` -- the first
        thing the root detector computes from the volume (``center_net.py:230``) -- without
        the volume: ``[B,C,X,Y]``, bit-identical, 246 MB of stores fewer at the Panoptic
        batch.  Arguments as :meth:`forward`.  The volume's only other consumer, the K
        z-columns of ``tempo_detector.py:204-206``, comes from :meth:`zcolumns`.  When a feature
        map requires a gradient the call goes through :meth:`forward` and ``torch.max``
        (differentiable, unfused)."""
        if _wants_grad(feature_maps):  # training: the volume and autograd's own torch.max
            cubes = self._training_volume(feature_maps, grid_size, grid_center, cube_size,
                                          centers, scales, cameras)
            return torch.max(cubes, dim=4)[0]
        self._train_volume = None
        device, B, C, live, axes, cube_size, cam = self._consumer_inputs(
            feature_maps, grid_size, grid_center, cube_size, centers, scales, cameras,
            'forward_zmax')
        zmax = torch.empty(B, C, cube_size[0], cube_size[1], dtype=torch.float32, device=device)
        source = _map_source(feature_maps)
        for c0, cs in _Slices(C).items:
            zs = zmax if cs == C else torch.empty(B, cs, *zmax.shape[2:], dtype=torch.float32,
                                                  device=device)
            kw = dict(live=live, mm_mode=self.mm_mode, zmax_out=zs, want_cubes=False)
            if source == 'packed':  # (at most 32 channels: one slice)
                ops.unproject_root(feature_maps.hm, cam, axes, cs, self.heatmap_size,
                                   self.image_size, **kw)
            elif source == 'views':
                ops.unproject_root_views(ops.channels_last_layout(list(feature_maps), c0), cam,
                                         axes, cs, self.heatmap_size, self.image_size, **kw)
            else:
                ops.root_step(feature_maps, cam, axes, self.heatmap_size, self.image_size, c0,
                              cs, **kw)
            if cs != C:
                zmax[:, c0:c0 + cs] = zs
        return zmax

    def zcolumns(self, feature_maps, grid_size, grid_center, cube_size, centers, scales,
                 cameras, index):
        """Extension: the z-columns of the volume at the cells ``index``
        int64 ``[B,K]`` (cell = x * Y + y, the detector's ``topk_2d_flatten_index``) ->
        ``[B,K,C,Z]`` = ``torch.gather(torch.flatten(cubes, 2, 3).permute(0, 2, 1, 3), 1,
        index[..., None, None].repeat(1, 1, C, Z))`` of ``tempo_detector.py:204-206``,
        computed from the heatmaps (only those K * Z voxels are unprojected): bit-identical
        to gathering from :meth:`forward`'s volume, which need not exist."""
        if _wants_grad(feature_maps):  # training: the detector's own expression on the volume
            cubes = self._training_volume(feature_maps, grid_size, grid_center, cube_size,
                                          centers, scales, cameras)
            B_, C_, Z_ = cubes.shape[0], cubes.shape[1], cubes.shape[4]
            idx = index.to(device=cubes.device, dtype=torch.int64).view(B_, -1, 1, 1)
            return torch.gather(torch.flatten(cubes, 2, 3).permute(0, 2, 1, 3), dim=1,
                                index=idx.repeat(1, 1, C_, Z_))
        device, B, C, live, axes, cube_size, cam = self._consumer_inputs(
            feature_maps, grid_size, grid_center, cube_size, centers, scales, cameras,
            'zcolumns')
        index = index.to(device=device, dtype=torch.int64).contiguous()
        K = int(index.shape[1])
        out = torch.empty(B, K, C, cube_size[2], dtype=torch.float32, device=device)
        source = _map_source(feature_maps)
        for c0, cs in _Slices(C).items:
            dst = out if cs == C else None
            kw = dict(channels=cs, live=live, out=dst, mm_mode=self.mm_mode)
            if source == 'packed':
                got = ops.unproject_root_columns(feature_maps.hm, cam, axes, index,
                                                 self.heatmap_size, self.image_size, **kw)
            elif source == 'views':
                got = ops.unproject_root_columns(
                    ops.channels_last_layout(list(feature_maps), c0), cam, axes, index,
                    self.heatmap_size, self.image_size, **kw)
            else:
                got = ops.unproject_root_columns(list(feature_maps), cam, axes, index,
                                                 self.heatmap_size, self.image_size,
                                                 c_begin=c0, **kw)
            if cs != C:
                out[:, :, c0:c0 + cs] = got
        return out

    def forward_sequence(self, feature_maps_t, grid_size, grid_center_t, cube_size,
                         centers_t, scales_t, cameras_t):
        """Extension: the detector's loop over ``time_idx``
        (``tempo_detector.py:175-191``) in one call.  Every argument that the
        reference indexes with ``time_idx`` is a list over T here:
        ``feature_maps_t[t][v]`` = ``[B,C,H,W]``, ``grid_center_t[t]`` = the
        ``[space_center]`` list of timestep t, ``centers_t[t][b][v]`` etc.  Returns
        the cubes of ``torch.cat([forward(t) for t], dim=0)``: ``[T*B,C,X,Y,Z]``
        (bit-identical; forward-only, grids are not materialised)."""
        T = len(feature_maps_t)
        device = _check_inputs(feature_maps_t[0])
        if any(_wants_grad(f) for f in feature_maps_t):  # training: the caller's own loop
            return torch.cat([self.forward(feature_maps_t[t], grid_size, grid_center_t[t],
                                           cube_size, centers_t[t], scales_t[t],
                                           cameras_t[t])[0] for t in range(T)], dim=0)
        B, C = feature_maps_t[0][0].shape[:2]
        ax, lives = [], []
        for t in range(T):
            a, cs = self._axes(grid_size, grid_center_t[t], cube_size, B, device)
            ax.append([x if x.dim() == 2 else x[None, :].expand(B, -1) for x in a])
            lv = self._live(grid_center_t[t], B, device)
            lives.append(torch.ones(B, dtype=torch.uint8, device=device) if lv is None else lv)
        axes = [torch.cat([ax[t][i] for t in range(T)], 0).contiguous() for i in range(3)]
        live = torch.cat(lives).contiguous()
        cam = torch.cat([pack_cameras(centers_t[t], scales_t[t], cameras_t[t], self.image_size,
                                      self.mm_mode) for t in range(T)], 0).to(device)
        cubes = torch.empty(T * B, C, *cs, dtype=torch.float32, device=device)
        # one library call per timestep: its chunk loop keeps a chunk's channel-last maps
        # cached between the repack and the gather (one launch over all T*B frames reads
        # them back from HBM: 0.34 against 0.27 ms per 32 frames at config 2)
        for t in range(T):  # timesteps whose maps are already channel-last: in place
            source = _map_source(feature_maps_t[t])
            if source != 'nchw':
                sl = slice(t * B, (t + 1) * B)
                self._get_voxel_in_place(feature_maps_t[t], source, cam[sl],
                                         [a[sl] for a in axes], live[sl], cubes[sl], plain=True)
        for c0, n in _Slices(C).items:
            for t in range(T):
                if _map_source(feature_maps_t[t]) != 'nchw':
                    continue
                sl = slice(t * B, (t + 1) * B)
                out, _, _ = ops.root_step(
                    feature_maps_t[t], cam[sl], [a[sl] for a in axes], self.heatmap_size,
                    self.image_size, c_begin=c0, channels=n, live=live[sl],
                    out=cubes[sl] if n == C else None, mm_mode=self.mm_mode)
                if n != C:
                    cubes[sl, c0:c0 + n] = out
        return cubes


class ProjectLayerWithMask(nn.Module):
    """Per-person window unprojection, ``project_layer.py:160-388``."""

    def __init__(self, image_size, heatmap_size, agg_method='mean'):
        super().__init__()
        if agg_method != 'mean':
            raise NotImplementedError(f"agg_method={agg_method!r}: only 'mean' is supported")
        self.image_size = _pair(image_size)
        self.heatmap_size = _pair(heatmap_size)
        self.agg_method = agg_method
        self.mm_mode = 'auto'  # rounding flavour of the K=3 matmuls (numerics.py)
        self._fine_axes = None  # device axes; cached like the reference's fine_grid
        self._fine_grid = None
        self.center_grid = None

    # the reference caches ``fine_grid`` ([fx*fy*fz, 3], 49 MB at the Panoptic
    # sizes) at the first call and never refreshes it; only its three axis
    # vectors are needed here, the full grid is built on first access.
    @property
    def fine_grid(self):
        if self._fine_grid is None and self._fine_axes is not None:
            gx, gy, gz = torch.meshgrid(*self._fine_axes, indexing='ij')
            self._fine_grid = torch.stack(
                [gx.reshape(-1), gy.reshape(-1), gz.reshape(-1)], dim=1)
        return self._fine_grid

    def set_space_size(self, space_center, space_size, sub_space_size,
                       sub_cube_size, device):
        """``project_layer.py:184-201``, same tensor ops on ``device`` (skipped when the
        arguments are the ones of the previous call: the pose regressor calls with the same
        geometry for every candidate and timestep)."""
        key = (tuple(space_center), tuple(space_size), tuple(sub_space_size),
               tuple(sub_cube_size), str(device))
        if getattr(self, '_space_key', None) == key:
            return
        self.whole_space_center = torch.tensor(space_center, device=device)
        self.whole_space_size = torch.tensor(space_size, device=device)
        self.ind_space_size = torch.tensor(sub_space_size, device=device)
        self.voxels_per_axis = torch.tensor(sub_cube_size, device=device,
                                            dtype=torch.int32)
        self.fine_voxels_per_axis = (self.whole_space_size / self.ind_space_size *
                                     (self.voxels_per_axis - 1)).int() + 1
        self.scale = (self.fine_voxels_per_axis.float() - 1) / self.whole_space_size
        self.bias = (-self.ind_space_size / 2.0 / self.whole_space_size *
                     (self.fine_voxels_per_axis - 1) - self.scale *
                     (self.whole_space_center - self.whole_space_size / 2.0))
        self.num_bins = (self.voxels_per_axis[0] * self.voxels_per_axis[1] *
                         self.voxels_per_axis[2])
        self.num_fine_bins = (self.fine_voxels_per_axis[0] *
                              self.fine_voxels_per_axis[1] *
                              self.fine_voxels_per_axis[2])
        # host copies of the integer geometry (same ops on CPU tensors, no sync)
        ws = torch.tensor(space_size)
        ins = torch.tensor(sub_space_size)
        vpa = torch.tensor(sub_cube_size, dtype=torch.int32)
        self._sub_cube = [int(v) for v in vpa]
        self._fine_bins = [int(v) for v in ((ws / ins * (vpa - 1)).int() + 1)]
        self._space = (list(space_center), list(space_size), list(sub_space_size))
        # fp32 copies for tempo_window_geometry (integer configs give int64 tensors above; the
        # reference's ops promote them to fp32 operand by operand: same values)
        self._wss_f = self.whole_space_size.float().contiguous()
        self._iss_f = self.ind_space_size.float().contiguous()
        self._space_key = key

    def compute_grid(self, box_size, box_center, num_bins, device=None):
        return ProjectLayer.compute_grid(self, box_size, box_center, num_bins, device)

    def window_geometry(self, grid_center):
        """``project_layer.py:299-324``: (centers_tl, offset, start, end)."""
        device = grid_center.device
        B = grid_center.shape[0]
        centers_tl = torch.round(grid_center[:, :3].float() * self.scale +
                                 self.bias).int()
        offset = (centers_tl.float() / (self.fine_voxels_per_axis - 1) *
                  self.whole_space_size - self.whole_space_size / 2.0 +
                  self.ind_space_size / 2.0)
        mask = ((1 - grid_center[:, 5:7]) / 2 * (self.voxels_per_axis[0:2] - 1)).int()
        mask[mask < 0] = 0
        mask = torch.cat(
            [mask, torch.zeros((B, 1), device=device, dtype=torch.int32)], dim=1)
        start = torch.where(centers_tl + mask >= 0, centers_tl + mask,
                            torch.zeros_like(centers_tl))
        end = torch.where(
            centers_tl + self.voxels_per_axis - mask <= self.fine_voxels_per_axis,
            centers_tl + self.voxels_per_axis - mask, self.fine_voxels_per_axis)
        end = torch.maximum(end, start + 1)
        return centers_tl, offset, start, end

    def _ensure_grids(self, device):
        if self._fine_axes is not None:
            return
        center, size, sub = self._space
        self._fine_axes_host = [_base_axis(size[i], self._fine_bins[i]) + center[i]
                                for i in range(3)]
        self._fine_axes = [a.to(device) for a in self._fine_axes_host]
        vx = self._sub_cube
        ax = [_base_axis(sub[i], vx[i]) + center[i] for i in range(3)]
        gx, gy, gz = torch.meshgrid(*ax, indexing='ij')
        grid = torch.stack([gx, gy, gz], dim=-1)
        self.center_grid = torch.stack([
            grid[:, :, 0, :2].reshape(-1, 2), grid[:, 0, :, ::2].reshape(-1, 2),
            grid[0, :, :, 1:].reshape(-1, 2)
        ]).to(device)

    def _window_inputs(self, feature_maps, grid_center, centers, scales, cameras):
        """Per-call constants of the window kernels: integer window geometry
        (``project_layer.py:299-324``), skip flags (``:232``), packed cameras."""
        device = _check_inputs(feature_maps)
        B, C = _batch_channels(feature_maps)
        if grid_center.device != device:
            grid_center = grid_center.to(device)
        cols = grid_center.shape[1]
        if (grid_center.dtype == torch.float32 and grid_center.is_contiguous()
                and self.scale.dtype == torch.float32 and self.bias.dtype == torch.float32
                and cols >= 7):
            # one launch instead of the reference's fifteen tensor ops (bit-identical)
            centers_tl, offset, start, end, live = ops.window_geometry(
                grid_center, self.scale, self.bias, self._wss_f, self._iss_f,
                self._fine_bins, self._sub_cube)
        else:
            centers_tl, offset, start, end = self.window_geometry(grid_center)
            live = None
            if grid_center.shape[1] != 3:
                live = (grid_center[:, 3] >= 0).to(torch.uint8).contiguous()
        self._ensure_grids(device)
        self._cam_host = pack_cameras(centers, scales, cameras, self.image_size, self.mm_mode)
        cam = to_device(self._cam_host, device)
        geo = tuple(t.to(torch.int32).contiguous() for t in (centers_tl, start, end))
        return device, B, C, offset, live, cam, geo

    def get_voxel(self, feature_maps, grid_center, centers, scales, cameras):
        device, B, C, offset, live, cam, (tl, st, en) = self._window_inputs(
            feature_maps, grid_center, centers, scales, cameras)
        if _wants_grad(feature_maps):  # training: differentiable path
            # the backward's LDS tile: from the lattice's pitch in heatmap pixels, unless set
            tile = int(getattr(self, 'backward_tile', 0) or 0)
            if tile == 0 and not self._cam_host.is_cuda:
                tile = ops.backward_tile_texels(self._cam_host, self._fine_axes_host,
                                                self.heatmap_size, self.image_size)
            meta = dict(kind='window', cam=cam, fine_axes=self._fine_axes, tl=tl, start=st,
                        end=en, sub_cube=list(self._sub_cube), live=live,
                        heatmap_size=self.heatmap_size, image_size=self.image_size,
                        mm_mode=self.mm_mode, tile_texels=tile)
            cubes = _Unproject.apply(meta, *[fm.float() for fm in feature_maps])
            return cubes, offset
        cubes = torch.empty(B, C, *self._sub_cube, dtype=torch.float32, device=device)
        source = _map_source(feature_maps)
        if source == 'packed':  # already channel-last: no repack
            ops.unproject_window(feature_maps.hm, cam, self._fine_axes, tl, st, en,
                                 self._sub_cube, C, self.heatmap_size, self.image_size,
                                 live=live, out=cubes, mm_mode=self.mm_mode)
            return cubes, offset
        if source == 'views':  # channels_last tensors read in place
            for c0, cs in _Slices(C).items:
                lay = ops.channels_last_layout(list(feature_maps), c0)
                out = ops.unproject_window_views(
                    lay, cam, self._fine_axes, tl, st, en, self._sub_cube, cs,
                    self.heatmap_size, self.image_size, live=live,
                    out=cubes if cs == C else None, mm_mode=self.mm_mode)
                if cs != C:
                    cubes[:, c0:c0 + cs] = out
            return cubes, offset
        for c0, cs in _Slices(C).items:
            out, _ = ops.window_step(
                feature_maps, cam, self._fine_axes, tl, st, en, self._sub_cube,
                self.heatmap_size, self.image_size, windows_per_frame=1, c_begin=c0,
                channels=cs, live=live, out=cubes if cs == C else None, mm_mode=self.mm_mode)
            if cs != C:
                cubes[:, c0:c0 + cs] = out
        return cubes, offset

    def get_planes(self, feature_maps, grid_center, centers, scales, cameras,
                   want_cubes=False):
        """Extension: the three axis-max planes the pose regressor
        takes from the window volume (``tempo_pose.py:172-176``), computed inside the
        unprojection kernel.  Call ``set_space_size`` first (as ``forward`` does).
        Returns ``(plane_xy [B,C,X,Y], plane_xz [B,C,X,Z], plane_yz [B,C,Y,Z], offset,
        cubes | None)``; with ``want_cubes=False`` the ``[B,C,X,Y,Z]`` volume is never
        written to memory -- in training either: when a feature map requires a gradient the
        planes come from :class:`_UnprojectPlanes`, whose backward sends the planes'
        gradients to the heatmaps without a volume or a gradient volume."""
        if _wants_grad(feature_maps):
            if want_cubes:
                # the caller wants the volume as well: the differentiable volume path and
                # autograd's own torch.max (tempo_pose.py:172-176 as written)
                cubes, offset = self.get_voxel(feature_maps, grid_center, centers, scales,
                                               cameras)
                return (torch.max(cubes, dim=4)[0], torch.max(cubes, dim=3)[0],
                        torch.max(cubes, dim=2)[0], offset, cubes)
            # training through the planes alone: no volume, no gradient volume
            device, B, C, offset, live, cam, (tl, st, en) = self._window_inputs(
                feature_maps, grid_center, centers, scales, cameras)
            tile = int(getattr(self, 'backward_tile', 0) or 0)
            if tile == 0 and not self._cam_host.is_cuda:
                tile = ops.backward_tile_texels(self._cam_host, self._fine_axes_host,
                                                self.heatmap_size, self.image_size, planes=True)
            meta = dict(cam=cam, fine_axes=self._fine_axes, tl=tl, start=st, end=en,
                        sub_cube=list(self._sub_cube), live=live,
                        heatmap_size=self.heatmap_size, image_size=self.image_size,
                        mm_mode=self.mm_mode, tile_texels=tile)
            pxy, pxz, pyz = _UnprojectPlanes.apply(meta, *[fm.float() for fm in feature_maps])
            return pxy, pxz, pyz, offset, None
        device, B, C, offset, live, cam, (tl, st, en) = self._window_inputs(
            feature_maps, grid_center, centers, scales, cameras)
        vx, vy, vz = self._sub_cube
        cubes = (torch.empty(B, C, vx, vy, vz, dtype=torch.float32, device=device)
                 if want_cubes else None)
        planes = [torch.empty(B, C, *sh, dtype=torch.float32, device=device)
                  for sh in ((vx, vy), (vx, vz), (vy, vz))]
        source = _map_source(feature_maps)
        for c0, cs in _Slices(C).items:
            whole = cs == C
            kw = dict(live=live, want_cubes=want_cubes, mm_mode=self.mm_mode,
                      out=cubes if whole else None, planes_out=planes if whole else None)
            if source == 'views':  # channels_last tensors read in place: no repack
                out = ops.unproject_window_planes_views(
                    ops.channels_last_layout(list(feature_maps), c0), cam, self._fine_axes, tl,
                    st, en, self._sub_cube, cs, self.heatmap_size, self.image_size, **kw)
            else:  # (packed maps hold at most 32 channels: one slice)
                hm = feature_maps.hm if source == 'packed' else ops.repack_nhwc(feature_maps, c0, cs)
                out = ops.unproject_window_planes(
                    hm, cam, self._fine_axes, tl, st, en, self._sub_cube, cs,
                    self.heatmap_size, self.image_size, **kw)
            if not whole:
                if want_cubes:
                    cubes[:, c0:c0 + cs] = out[0]
                for dst, src in zip(planes, out[1:]):
                    dst[:, c0:c0 + cs] = src
        return planes[0], planes[1], planes[2], offset, cubes

    def forward_sequence_planes(self, feature_maps_t, candidates, space_center, space_size,
                                sub_space_size, sub_cube_size, centers_t, scales_t,
                                cameras_t, want_cubes=False):
        """Extension (SURVEY 8f rows 1 + 4): the pose regressor's two loops -- over
        candidates (``tempo_pose.py:266``) and over ``time_idx`` (``:145-156``) -- and
        its axis-max (``:172-176``) in one repack per timestep and ONE kernel launch.

        ``feature_maps_t[t][v]`` = ``[B,C,H,W]``; ``candidates`` = ``[B,P,7]``
        (``human_candidates[:, -1]``: the reference unprojects candidate n at every
        timestep around its LAST-timestep centre); ``centers_t[t][b][v]`` etc.
        Returns ``(plane_xy, plane_xz, plane_yz)`` each ``[B,P,T,C,64,64]`` =
        ``torch.max`` over z / y / x of what ``forward(feature_maps_t[t],
        candidates[:, n], ...)`` returns, ``offset [B,P,3]``, and the cubes
        ``[B,P,T,C,X,Y,Z]`` when ``want_cubes`` (else None: never written)."""
        T = len(feature_maps_t)
        device = _check_inputs(feature_maps_t[0])
        self.set_space_size(space_center, space_size, sub_space_size, sub_cube_size, device)
        if any(_wants_grad(f) for f in feature_maps_t) and not want_cubes:
            # training through the planes: all B * P * T windows in one differentiable call
            # (one repack per timestep, one claim pass, one backward whose atomics also do
            # the summing over proposals that autograd would do tensor by tensor)
            B, C = feature_maps_t[0][0].shape[:2]
            V = len(feature_maps_t[0])
            P = candidates.shape[1]
            cand = candidates.to(device).reshape(B * P, -1)
            centers_tl, offset, start, end = self.window_geometry(cand)
            self._ensure_grids(device)
            live = None
            if cand.shape[1] != 3:
                live = (cand[:, 3] >= 0).to(torch.uint8).repeat_interleave(T).contiguous()
            rep = lambda x: x.to(torch.int32).repeat_interleave(T, 0).contiguous()
            i = torch.arange(B * P * T, device=device)
            win_frame = ((i % T) * B + i // (P * T)).to(torch.int32).contiguous()
            cam_host = torch.cat([pack_cameras(centers_t[t], scales_t[t], cameras_t[t],
                                               self.image_size, self.mm_mode)
                                  for t in range(T)], 0)
            tile = int(getattr(self, 'backward_tile', 0) or 0)
            if tile == 0 and not cam_host.is_cuda:
                tile = ops.backward_tile_texels(cam_host, self._fine_axes_host,
                                                self.heatmap_size, self.image_size, planes=True)
            meta = dict(cam=cam_host.to(device), fine_axes=self._fine_axes, tl=rep(centers_tl),
                        start=rep(start), end=rep(end), sub_cube=list(self._sub_cube),
                        live=live, win_frame=win_frame, T=T, heatmap_size=self.heatmap_size,
                        image_size=self.image_size, mm_mode=self.mm_mode, tile_texels=tile)
            flat = [fm.float() for fm_t in feature_maps_t for fm in fm_t]
            planes = _UnprojectPlanes.apply(meta, *flat)
            shape = lambda x: x.view(B, P, T, *x.shape[1:])
            return (shape(planes[0]), shape(planes[1]), shape(planes[2]),
                    offset.view(B, P, 3), None)
        if any(_wants_grad(f) for f in feature_maps_t):
            # training with the volume asked for: the caller's two loops as written
            # (tempo_pose.py:145-156, :266), each call through the differentiable volume path
            P_ = candidates.shape[1]
            rows = [[self.get_planes(feature_maps_t[t], candidates[:, n].to(device),
                                     centers_t[t], scales_t[t], cameras_t[t],
                                     want_cubes=want_cubes) for t in range(T)]
                    for n in range(P_)]
            pick = lambda k: torch.stack([torch.stack([rows[n][t][k] for t in range(T)], 1)
                                          for n in range(P_)], 1)
            offset = torch.stack([rows[n][0][3] for n in range(P_)], 1)
            return pick(0), pick(1), pick(2), offset, (pick(4) if want_cubes else None)
        B, C = feature_maps_t[0][0].shape[:2]
        V = len(feature_maps_t[0])
        H, W = feature_maps_t[0][0].shape[2:]
        P = candidates.shape[1]
        cand = candidates.to(device).reshape(B * P, -1)
        centers_tl, offset, start, end = self.window_geometry(cand)
        self._ensure_grids(device)
        live = None
        if cand.shape[1] != 3:
            live = (cand[:, 3] >= 0).to(torch.uint8).repeat_interleave(T).contiguous()
        # windows ordered (b, p, t); frames ordered (t, b)
        rep = lambda x: x.to(torch.int32).repeat_interleave(T, 0).contiguous()
        tl, st, en = rep(centers_tl), rep(start), rep(end)
        i = torch.arange(B * P * T, device=device)
        win_frame = ((i % T) * B + i // (P * T)).to(torch.int32).contiguous()
        cam = torch.cat([pack_cameras(centers_t[t], scales_t[t], cameras_t[t], self.image_size,
                                      self.mm_mode) for t in range(T)], 0).to(device)
        vx, vy, vz = self._sub_cube
        NW = B * P * T
        cubes = (torch.empty(NW, C, vx, vy, vz, dtype=torch.float32, device=device)
                 if want_cubes else None)
        planes = [torch.empty(NW, C, *sh, dtype=torch.float32, device=device)
                  for sh in ((vx, vy), (vx, vz), (vy, vz))]
        packed_all = _packed_sequence(feature_maps_t)
        for c0, n in _Slices(C).items:
            cp = ops.padded_channels(n)
            if packed_all is not None:  # one buffer, frames ordered (t, b): no repack
                hm = packed_all
            else:
                hm = torch.empty(T * B, V, H, W, cp, dtype=torch.float32, device=device)
                for t in range(T):
                    fm_t = feature_maps_t[t]
                    if isinstance(fm_t, PackedHeatmaps):
                        hm[t * B:(t + 1) * B] = fm_t.hm
                    else:
                        ops.repack_nhwc(fm_t, c0, n, out=hm[t * B:(t + 1) * B])
            whole = n == C
            out = ops.unproject_window_planes(
                hm, cam, self._fine_axes, tl, st, en, self._sub_cube, n, self.heatmap_size,
                self.image_size, win_frame=win_frame, live=live, want_cubes=want_cubes,
                mm_mode=self.mm_mode, out=cubes if whole else None,
                planes_out=planes if whole else None)
            if not whole:
                if want_cubes:
                    cubes[:, c0:c0 + n] = out[0]
                for dst, src in zip(planes, out[1:]):
                    dst[:, c0:c0 + n] = src
        shape = lambda x: x.view(B, P, T, *x.shape[1:])
        return (shape(planes[0]), shape(planes[1]), shape(planes[2]), offset.view(B, P, 3),
                None if cubes is None else shape(cubes))

    def forward(self, feature_maps, grid_center, space_center, space_size,
                sub_space_size, sub_cube_size, centers, scales, cameras):
        device = feature_maps[0].device
        self.set_space_size(space_center, space_size, sub_space_size,
                            sub_cube_size, device)
        return self.get_voxel(feature_maps, grid_center, centers, scales, cameras)
