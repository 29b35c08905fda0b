"""The kernels' arithmetic contract (tempo_amd/csrc/unproject_math.h), built with
gcc (tests/emul) and run on the CPU, against the oracle -- bit for bit.  The
GPU runs the same header through hipcc; tests/test_gpu_parity.py checks that
the device result equals this one's reference on a real MI355X."""
import ctypes

import numpy as np
import pytest
import torch

import cases
import emul_helpers as E
from oracle import project_layer_ref as O
from tempo_amd import numerics


def _bits(bounding):
    V = bounding.shape[-1]
    return (bounding.to(torch.int32) << torch.arange(V, dtype=torch.int32)).sum(-1)


@pytest.mark.parametrize('name', list(cases.ROOT_CASES))
def test_root_contract_bit_exact(emul, name):
    case = cases.build_root(name)
    layer = O.ProjectLayer(case['image_size'], case['heatmap_size'])
    ref, _ = layer(*case['args'])
    got, mask = E.run_root(emul, case)
    assert np.array_equal(got, ref.numpy()), float(np.abs(got - ref.numpy()).max())
    assert np.array_equal(mask, _bits(layer.last_bounding).numpy().astype(np.uint32))


@pytest.mark.parametrize('name', list(cases.WINDOW_CASES))
def test_window_contract_bit_exact(emul, name):
    case = cases.build_window(name)
    layer = O.ProjectLayerWithMask(case['image_size'], case['heatmap_size'])
    ref, off = layer(*case['args'])
    got, off2, cg = E.run_window(emul, case)
    assert np.array_equal(got, ref.numpy()), float(np.abs(got - ref.numpy()).max())
    assert torch.equal(off, off2)
    assert torch.equal(cg, layer.center_grid)


@pytest.mark.parametrize('V', list(range(1, 16)))
def test_view_sum_order_is_torchs(emul, V):
    order = (ctypes.c_int * V)()
    emul.emul_view_order(V, order)
    order = list(order)
    assert sorted(order) == list(range(V))
    torch.manual_seed(V)
    x = torch.rand(4000, V)
    acc = np.zeros(4000, np.float32)
    for v in order:
        acc = (acc + x[:, v].numpy()).astype(np.float32)
    assert np.array_equal(acc, torch.sum(x, dim=-1).numpy())


def test_voxel_on_principal_plane_is_nan_like_the_reference(emul):
    """z == 0 in the camera frame: the reference divides by zero, the sample is
    NaN, NaN * 0 poisons the masked mean.  Reproduced, not 'fixed'."""
    fms = [torch.rand(1, 3, 8, 8)]
    cam = {'R': np.eye(3).tolist(), 'T': [[0.0], [0.0], [0.0]],
           'K': [[10.0, 0.0, 4.0], [0.0, 10.0, 4.0]]}
    centers = [[np.array([4.0, 4.0], np.float32)]]
    scales = [[np.array([8.0, 8.0], np.float32)]]
    # 3 z-planes: -100, 0, 100 -> the middle one sits on the principal plane
    args = (fms, [200, 200, 200], [[0.0, 0.0, 0.0]], [2, 2, 3], centers, scales, [[cam]])
    case = dict(args=args, image_size=[8, 8], heatmap_size=[8, 8])
    ref, _ = O.ProjectLayer([8, 8], [8, 8])(*args)
    got, _ = E.run_root(emul, case)
    assert torch.isnan(ref).any()
    assert np.array_equal(np.isnan(got), torch.isnan(ref).numpy())
    ok = ~np.isnan(got)
    assert np.array_equal(got[ok], ref.numpy()[ok])


def test_unfused_flavour_matches_explicit_products(emul):
    """the 'unfused' flavour is (a0*b0 + a1*b1) + a2*b2 with every step rounded"""
    from tempo_amd import synthetic as S
    from tempo_amd.camera import pack_cameras
    cfg = S.PANOPTIC
    cams = S.ring_cameras(2, seed=9)
    centers, scales, cameras = S.frame_meta(cams, 1, cfg['raw_size'], cfg['image_size'])
    table = pack_cameras(centers, scales, cameras, cfg['image_size'], 'unfused')[0]
    grid = O.compute_grid(cfg['space_size'], cfg['space_center'], [12, 12, 6])
    n = grid.shape[0]
    f = ctypes.c_float
    for v in range(2):
        rec = table[v].numpy().copy()
        pix = np.zeros((n, 2), np.float32)
        ins = np.zeros(n, np.int32)
        g = np.zeros((n, 2), np.float32)
        emul.emul_project(rec.ctypes.data_as(ctypes.c_void_p),
                          grid.numpy().ctypes.data_as(ctypes.c_void_p), ctypes.c_long(n),
                          f(240), f(128), f(960), f(512), 1,
                          pix.ctypes.data_as(ctypes.c_void_p),
                          ins.ctypes.data_as(ctypes.c_void_p),
                          g.ctypes.data_as(ctypes.c_void_p))
        R = torch.from_numpy(rec[0:9].reshape(3, 3))
        T = torch.from_numpy(rec[9:12])
        K = torch.from_numpy(rec[12:18].reshape(2, 3))
        xc = (grid[:, 0:1] * R[0] + grid[:, 1:2] * R[1]) + grid[:, 2:3] * R[2] + T
        x = xc / xc[:, 2:]
        k, p = torch.from_numpy(rec[18:21]), torch.from_numpy(rec[21:23])
        r2 = x[:, 0] * x[:, 0] + x[:, 1] * x[:, 1]
        radial = 1 + ((k[0] * r2 + k[1] * (r2 * r2)) + k[2] * ((r2 * r2) * r2))
        tang = 2 * (p[1] * x[:, 0] + p[0] * x[:, 1])
        xd = x[:, 0] * (radial + tang) + r2 * p[1]
        yd = x[:, 1] * (radial + tang) + r2 * p[0]
        px = (xd * K[0, 0] + yd * K[0, 1]) + x[:, 2] * K[0, 2]
        py = (xd * K[1, 0] + yd * K[1, 1]) + x[:, 2] * K[1, 2]
        assert np.array_equal(pix[:, 0], px.numpy()) and np.array_equal(pix[:, 1], py.numpy())


def test_flavour_probe_is_consistent():
    m = numerics.probe()
    assert set(m) == set(numerics.SITES)
    assert all(v in ('fma', 'unfused') for v in m.values())
    assert numerics.probe(n=4096) == m
    assert numerics.abi_flag('fma') == 0 and numerics.abi_flag('unfused') == 7


def _extreme_maps(case):
    """Heatmap channels scaled into the fp32 denormal range, up to the overflow threshold, with
    a NaN and an infinity planted: the masked mean's division then runs its denormal-quotient
    and NaN paths (normalise_tile / tempo_div_r64 in the kernels)."""
    fms = [f.clone() for f in case['args'][0]]
    scales = [1e-38, 3e-41, 2e-44, 1.0, 3e38, 1e-45, 7e-40]
    for f in fms:
        for c in range(f.shape[1]):
            f[:, c] *= scales[c % len(scales)]
    H, W = fms[0].shape[2:]
    fms[0][0, 3, H // 3:H // 2, W // 3:W // 2] = float('nan')
    fms[-1][0, 10, H // 2:2 * H // 3, W // 2:2 * W // 3] = float('inf')
    out = dict(case)
    out['args'] = (fms,) + tuple(case['args'][1:])
    return out


def test_root_contract_extreme_values_bit_exact(emul):
    case = _extreme_maps(cases.build_root('root_panoptic_noise'))
    layer = O.ProjectLayer(case['image_size'], case['heatmap_size'])
    ref, _ = layer(*case['args'])
    got, _ = E.run_root(emul, case)
    r = ref.numpy()
    assert np.isnan(r).any() and (np.abs(r[np.isfinite(r)]) < 1.2e-38).any()  # both paths hit
    assert np.array_equal(np.isnan(got), np.isnan(r))
    assert np.array_equal(np.nan_to_num(got, nan=-1.0).view(np.uint32),
                          np.nan_to_num(r, nan=-1.0).view(np.uint32))
