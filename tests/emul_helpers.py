"""Run the gcc build of the kernel arithmetic (tests/emul) on a test case,
feeding it through the PRODUCT's host-side code (camera packing, axes, window
geometry) so that code is covered on CPU as well."""
import ctypes

import numpy as np
import torch

from tempo_amd import numerics, project_layer as PL
from tempo_amd.camera import pack_cameras


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def nhwc(feature_maps, cp):
    """[V x [B,C,H,W]] -> [B,V,H,W,CP] (torch reference of the repack kernel)."""
    x = torch.stack(feature_maps, dim=1).permute(0, 1, 3, 4, 2)
    B, V, H, W, C = x.shape
    out = torch.zeros(B, V, H, W, cp)
    out[..., :C] = x
    return out.contiguous()


def padded(c):
    cp = (c + 3) // 4 * 4
    return max(cp, 16)


def run_root(emul, case, mm_mode='auto'):
    fms, grid_size, grid_center, cube_size, centers, scales, cameras = case['args']
    B, C, H, W = fms[0].shape
    V = len(fms)
    layer = PL.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    live = layer._live(grid_center, B, 'cpu')
    axes, cube = layer._axes(grid_size, grid_center, cube_size, B, 'cpu')
    cam = pack_cameras(centers, scales, cameras, layer.image_size, mm_mode).numpy()
    cp = padded(C)
    hm = nhwc(fms, cp).numpy()
    X, Y, Z = cube
    N = X * Y * Z
    cubes = np.zeros((B, C, N), np.float32)
    mask = np.zeros((B, N), np.uint32)
    f = ctypes.c_float
    batched = axes[0].dim() == 2
    for b in range(B):  # emul_root takes shared axes: call per batch item
        ax = [np.ascontiguousarray((a[b] if batched else a).numpy()) for a in axes]
        lv = None if live is None else np.ascontiguousarray(live[b:b + 1].numpy())
        emul.emul_root(_p(hm[b:b + 1]), _p(cam[b:b + 1]), _p(ax[0]), _p(ax[1]), _p(ax[2]),
                       _p(lv) if lv is not None else None, 1, V, C, cp, H, W, X, Y, Z,
                       f(layer.heatmap_size[0]), f(layer.heatmap_size[1]),
                       f(layer.image_size[0]), f(layer.image_size[1]),
                       int(numerics.abi_flag(mm_mode) != 0), _p(cubes[b:b + 1]),
                       _p(mask[b:b + 1]))
    return cubes.reshape(B, C, X, Y, Z), mask


def run_window(emul, case, mm_mode='auto'):
    (fms, grid_center, space_center, space_size, sub_space, sub_cube, centers, scales,
     cameras) = case['args']
    B, C, H, W = fms[0].shape
    V = len(fms)
    layer = PL.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    layer.set_space_size(space_center, space_size, sub_space, sub_cube, 'cpu')
    tl, offset, st, en = layer.window_geometry(grid_center)
    layer._ensure_grids('cpu')
    live = (grid_center[:, 3] >= 0).to(torch.uint8).numpy()
    cam = pack_cameras(centers, scales, cameras, layer.image_size, mm_mode).numpy()
    cp = padded(C)
    hm = nhwc(fms, cp).numpy()
    fa = [np.ascontiguousarray(a.numpy()) for a in layer._fine_axes]
    VX, VY, VZ = layer._sub_cube
    cubes = np.zeros((B, C, VX * VY * VZ), np.float32)
    f = ctypes.c_float
    tl, st, en = (np.ascontiguousarray(t.to(torch.int32).numpy()) for t in (tl, st, en))
    emul.emul_window(_p(hm), _p(cam), _p(fa[0]), _p(fa[1]), _p(fa[2]), _p(tl), _p(st),
                     _p(en), _p(live), B, V, C, cp, H, W, VX, VY, VZ,
                     f(layer.heatmap_size[0]), f(layer.heatmap_size[1]),
                     f(layer.image_size[0]), f(layer.image_size[1]),
                     int(numerics.abi_flag(mm_mode) != 0), _p(cubes))
    return cubes.reshape(B, C, VX, VY, VZ), offset, layer.center_grid
