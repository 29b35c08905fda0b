"""Host-side flattening of the reference's per-view camera dicts into the packed
``CamParams[B,V,32]`` fp32 table the HIP kernels read (layout:
``include/tempo_unproject.h``).

Replaces, per (batch item, view), the host work the reference redoes inside
its python loops (citations relative to /root/reference):

* ``tempo/models/project_layer.py:97-102`` / ``:240-245`` -- ``width, height =
  center * 2`` and ``get_affine_transform(center, scale / 200, 0, image_size)``
  (mmpose + ``cv2.getAffineTransform``; restated here, cv2 is not required);
* ``project_layer.py:104-110`` / ``:247-254`` -- ``SimpleCameraTorch(param)``:
  ``R``, ``T_w2c = -(T^T @ R)``, ``K`` (or ``f``/``c``), optional ``k``/``p``.

Every value is produced with the same fp32/fp64 host operations the reference
would run (torch CPU ops for the tensors, a float64 6x6 solve for the affine),
so the table is bit-identical to what the reference's camera objects hold.
"""
import numpy as np
import torch

from . import numerics

CAM_R, CAM_T, CAM_K, CAM_KD, CAM_PD, CAM_HASD, CAM_TRANS = 0, 9, 12, 18, 21, 23, 24
CAM_WIDTH, CAM_HEIGHT, CAM_FLOATS = 30, 31, 32


def _third_point(a, b):
    d = a - b
    return b + np.stack([-d[..., 1], d[..., 0]], axis=-1).astype(np.float32)


def affine_transforms(centers, scales, output_size):
    """Batched ``get_affine_transform(center, scale, rot=0, output_size)``.

    ``centers``/``scales``: float32 ``[M,2]`` (``scales`` already divided by
    200 like the call sites do).  Returns float64 ``[M,2,3]``: the solution of
    cv2's 6x6 three-point system, solved in double precision like OpenCV does.
    """
    centers = np.asarray(centers, dtype=np.float32)
    scale_tmp = np.asarray(scales, dtype=np.float32) * 200.0  # stays float32
    m = centers.shape[0]
    dst_w, dst_h = output_size[0], output_size[1]
    src_w = scale_tmp[:, 0]
    # rot = 0: rotate_point([0, -src_w/2], 0) in float64, shift = (0, 0)
    src_dir = np.stack([np.zeros_like(src_w), src_w * -0.5], axis=-1)
    zshift = scale_tmp * np.zeros(2)  # scale_tmp * shift (float64 zeros)
    src = np.zeros((m, 3, 2), dtype=np.float32)
    src[:, 0] = centers + zshift
    src[:, 1] = centers + src_dir + zshift
    src[:, 2] = _third_point(src[:, 0], src[:, 1])
    dst = np.zeros((m, 3, 2), dtype=np.float32)
    dst[:, 0] = [dst_w * 0.5, dst_h * 0.5]
    dst[:, 1] = np.array([dst_w * 0.5, dst_h * 0.5]) + np.array([0., dst_w * -0.5])
    dst[:, 2] = _third_point(dst[:, 0], dst[:, 1])

    a = np.zeros((m, 6, 6), dtype=np.float64)
    b = np.zeros((m, 6), dtype=np.float64)
    for i in range(3):
        a[:, 2 * i, 0:2] = src[:, i]
        a[:, 2 * i, 2] = 1.0
        a[:, 2 * i + 1, 3:5] = src[:, i]
        a[:, 2 * i + 1, 5] = 1.0
        b[:, 2 * i] = dst[:, i, 0]
        b[:, 2 * i + 1] = dst[:, i, 1]
    return np.linalg.solve(a, b[..., None])[..., 0].reshape(m, 2, 3)


def _t_w2c(T, R, mm_mode):
    """``-(T^T @ R)`` of SimpleCameraTorch.  'auto': the torch op itself, i.e.
    whatever the host BLAS does (exactly the reference on this machine).  A
    pinned flavour is evaluated in float64, where every float32 product is
    exact (an FMA is then one float64 add + one rounding to float32; the
    double rounding can differ from a true FMA with probability ~2^-29)."""
    if mm_mode is None or mm_mode == 'auto':
        return (-(T.T) @ R).reshape(3)
    flavour = numerics.resolve(mm_mode)['R']
    t = T.reshape(3).numpy().astype(np.float64)
    r = R.numpy().astype(np.float64)
    # torch negates T first (exact), then multiplies: (-T)^T @ R
    p = [(-t[k]) * r[k, :] for k in range(3)]
    f = np.float32
    if flavour == 'fma':
        out = f(f(f(p[0]).astype(np.float64) + p[1]).astype(np.float64) + p[2])
    else:
        out = f(f(f(p[0]) + f(p[1])) + f(p[2]))
    return torch.from_numpy(out.astype(np.float32))


def _camera_record(param, out, mm_mode='auto'):
    """One ``SimpleCameraTorch.__init__`` worth of fields into ``out[0:24]``."""
    if 'intrinsics' in param:
        # the reference selects FisheyeCameraTorch here, a name defined nowhere
        # in its tree (project_layer.py:105-107)
        raise NotImplementedError(
            "camera dicts with 'intrinsics' (FisheyeCameraTorch) are not supported")
    R = torch.as_tensor(np.asarray(param['R'], dtype=np.float32))
    T = torch.as_tensor(np.asarray(param['T'], dtype=np.float32))
    if R.shape != (3, 3) or T.shape != (3, 1):
        raise ValueError(f'camera R/T must be 3x3 / 3x1, got {tuple(R.shape)} / {tuple(T.shape)}')
    out[CAM_R:CAM_R + 9] = R.reshape(9)
    out[CAM_T:CAM_T + 3] = _t_w2c(T, R, mm_mode)
    if 'K' in param:
        K = torch.as_tensor(np.asarray(param['K'], dtype=np.float32))
        if K.shape != (2, 3):
            raise ValueError(f'camera K must be 2x3, got {tuple(K.shape)}')
        out[CAM_K:CAM_K + 6] = K.reshape(6)
    elif 'f' in param and 'c' in param:
        f = np.asarray(param['f'], dtype=np.float32)
        c = np.asarray(param['c'], dtype=np.float32)
        if f.shape != (2, 1) or c.shape != (2, 1):
            raise ValueError('camera f/c must be 2x1 each')
        out[CAM_K + 0], out[CAM_K + 2] = float(f[0, 0]), float(c[0, 0])
        out[CAM_K + 4], out[CAM_K + 5] = float(f[1, 0]), float(c[1, 0])
    else:
        raise ValueError('Camera intrinsic parameters are missing. '
                         'Either "K" or "f"&"c" should be provided.')
    if 'k' in param and 'p' in param:
        k = np.asarray(param['k'], dtype=np.float32).reshape(-1)
        p = np.asarray(param['p'], dtype=np.float32).reshape(-1)
        if len(k) not in (3, 6) or len(p) != 2:
            raise ValueError('camera k must have 3 (or 6) and p 2 coefficients')
        # with 6 coefficients mmpose still applies only k[:3] (its rational
        # branch tests ``k.size == 6`` on a tensor and never runs)
        out[CAM_KD:CAM_KD + 3] = torch.from_numpy(k[:3].copy())
        out[CAM_PD:CAM_PD + 2] = torch.from_numpy(p.copy())
        out[CAM_HASD] = 1.0


def pack_cameras(centers, scales, cameras, image_size, mm_mode='auto'):
    """``centers[b][v]``, ``scales[b][v]`` (np.float32 (2,)), ``cameras[b][v]``
    (dict) -> CPU fp32 tensor ``[B,V,32]``.  ``mm_mode``: rounding flavour used
    for ``T_w2c = -(T^T @ R)``, see ``tempo_amd.numerics`` (default: the torch
    op itself, i.e. as the reference on this host)."""
    B = len(cameras)
    V = len(cameras[0])
    table = torch.zeros(B, V, CAM_FLOATS, dtype=torch.float32)
    cen = np.zeros((B * V, 2), dtype=np.float32)
    scl = np.zeros((B * V, 2), dtype=np.float32)
    seen = {}
    for b in range(B):
        if len(cameras[b]) != V or len(centers[b]) != V or len(scales[b]) != V:
            raise ValueError('every batch item needs one center/scale/camera per view')
        for v in range(V):
            cen[b * V + v] = centers[b][v]
            scl[b * V + v] = np.asarray(scales[b][v], dtype=np.float32) / 200.0
            param = cameras[b][v]
            hit = seen.get(id(param))
            if hit is None:
                _camera_record(param, table[b, v], mm_mode)
                seen[id(param)] = (b, v)
            else:
                table[b, v, :CAM_TRANS] = table[hit[0], hit[1], :CAM_TRANS]
    # identical (center, scale) pairs share one solve
    key = np.concatenate([cen, scl], axis=1)
    uniq, inv = np.unique(key, axis=0, return_inverse=True)
    trans = affine_transforms(uniq[:, :2], uniq[:, 2:], image_size)
    trans32 = torch.as_tensor(trans, dtype=torch.float).reshape(-1, 6)
    flat = table.view(B * V, CAM_FLOATS)
    flat[:, CAM_TRANS:CAM_TRANS + 6] = trans32[torch.as_tensor(inv.reshape(-1))]
    wh = torch.from_numpy(cen * 2)  # width, height = center * 2 (float32)
    flat[:, CAM_WIDTH] = wh[:, 0]
    flat[:, CAM_HEIGHT] = wh[:, 1]
    return table
