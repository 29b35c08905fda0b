"""Seeded synthetic inputs of the shapes the unprojection path consumes.

Nothing here is read from the reference at run time.  The formats follow the
reference's dataset code (citations relative to /root/reference):

* Panoptic camera dict ``{R (c2w) 3x3 list, T (c2w, mm) 3x1 list, K 2x3 list,
  k [3], p [2]}``: ``tempo/datasets/body3d_mview_temporal_panoptic_dataset.py:166-198``
* H36M camera dict ``{R 3x3, T 3x1, f 2x1, c 2x1, k [3], p [2]}`` (float32
  ndarrays): ``tempo/datasets/body3d_mview_temporal_human36m.py:574-593``; the
  four S1 calibrations below are the public Human3.6M constants that file
  embeds at ``:25-126`` (restated as data).
* ``center = (w/2, h/2)`` float32, ``scale = _get_scale((w, h))`` float32:
  ``tempo/datasets/kpt3d_mview_rgb_temporal_dataset.py:169-185``
* proposals ``[B,7] = (x, y, z, id/valid, conf, bbox_w, bbox_h)``:
  ``tempo/models/tempo_detector.py:216-218``
"""
import math

import numpy as np
import torch

# ------------------------------------------------------------------ configs
PANOPTIC = dict(
    space_size=[8000, 8000, 2000], space_center=[0, -500, 800],
    cube_size=[80, 80, 20], sub_space_size=[2000, 2000, 2000],
    sub_cube_size=[64, 64, 64], image_size=[960, 512], heatmap_size=[240, 128],
    raw_size=(1920, 1080), num_views=5, num_joints=15)
H36M = dict(
    space_size=[8000, 8000, 2000], space_center=[300, 300, 800],
    cube_size=[80, 80, 20], sub_space_size=[2000, 2000, 2000],
    sub_cube_size=[64, 64, 64], image_size=[384, 384], heatmap_size=[384, 384],
    raw_size=(1000, 1002), num_views=4, num_joints=17)

_H36M_INTRINSICS = [
    dict(center=[512.54150390625, 515.4514770507812],
         focal_length=[1145.0494384765625, 1143.7811279296875],
         radial=[-0.20709891617298126, 0.24777518212795258, -0.0030751503072679043],
         tangential=[-0.0009756988729350269, -0.00142447161488235]),
    dict(center=[508.8486328125, 508.0649108886719],
         focal_length=[1149.6756591796875, 1147.5916748046875],
         radial=[-0.1942136287689209, 0.2404085397720337, 0.006819975562393665],
         tangential=[-0.0016190266469493508, -0.0027408944442868233]),
    dict(center=[519.8158569335938, 501.40264892578125],
         focal_length=[1149.1407470703125, 1148.7989501953125],
         radial=[-0.2083381861448288, 0.25548800826072693, -0.0024604974314570427],
         tangential=[0.0014843869721516967, -0.0007599993259645998]),
    dict(center=[514.9682006835938, 501.88201904296875],
         focal_length=[1145.5113525390625, 1144.77392578125],
         radial=[-0.198384091258049, 0.21832367777824402, -0.008947807364165783],
         tangential=[-0.0005872055771760643, -0.0018133620033040643]),
]
_H36M_S1_EXTRINSICS = [
    dict(orientation=[0.1407056450843811, -0.1500701755285263, -0.755240797996521,
                      0.6223280429840088],
         translation=[1841.1070556640625, 4955.28466796875, 1563.4454345703125]),
    dict(orientation=[0.6157187819480896, -0.764836311340332, -0.14833825826644897,
                      0.11794740706682205],
         translation=[1761.278564453125, -5078.0068359375, 1606.2650146484375]),
    dict(orientation=[0.14651472866535187, -0.14647851884365082, 0.7653023600578308,
                      -0.6094175577163696],
         translation=[-1846.7777099609375, 5215.04638671875, 1491.972412109375]),
    dict(orientation=[0.5834008455276489, -0.7853162288665771, 0.14548823237419128,
                      -0.14749594032764435],
         translation=[-1794.7896728515625, -3722.698974609375, 1574.8927001953125]),
]


def _quat_inv_to_rotmat_T(q):
    """``quat_to_rotmat(qinverse(q)).T`` of the H36M loader, float32."""
    q = torch.tensor(q, dtype=torch.float32)
    q = torch.cat((q[:1], -q[1:]))
    q = q / q.norm(p=2)
    w, x, y, z = q
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz = w * x, w * y, w * z
    xy, xz, yz = x * y, x * z, y * z
    m = torch.stack([
        w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
        2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
        2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2
    ]).view(3, 3)
    return m.numpy().T.copy()


def h36m_cameras():
    """The four S1 cameras as H36M-format dicts (float32 ndarrays)."""
    cams = []
    for intr, ext in zip(_H36M_INTRINSICS, _H36M_S1_EXTRINSICS):
        cams.append({
            'R': _quat_inv_to_rotmat_T(ext['orientation']),
            'T': np.array(ext['translation'], dtype=np.float32).reshape(3, 1),
            'f': np.array(intr['focal_length'], dtype=np.float32).reshape(2, 1),
            'c': np.array(intr['center'], dtype=np.float32).reshape(2, 1),
            'k': np.array(intr['radial'], dtype=np.float32),
            'p': np.array(intr['tangential'], dtype=np.float32),
        })
    return cams


def _look_at(pos, target):
    """Camera->world rotation (columns = camera x/right, y/down, z/forward)."""
    f = target - pos
    f = f / np.linalg.norm(f)
    up = np.array([0.0, 0.0, 1.0])
    r = np.cross(f, up)
    r = r / np.linalg.norm(r)
    d = np.cross(f, r)
    return np.stack([r, d, f], axis=1)  # R_c2w


def ring_cameras(num_views, seed=0, target=(0.0, -500.0, 800.0), radius=3000.0,
                 raw_size=(1920, 1080), distortion=True):
    """Panoptic-like ring(s) of cameras looking at ``target`` (mm, z up)."""
    rng = np.random.RandomState(seed)
    target = np.asarray(target, dtype=np.float64)
    cams = []
    for v in range(num_views):
        ring = v // 5
        ang = 2 * math.pi * (v % 5) / 5 + 0.35 * ring + rng.uniform(-0.15, 0.15)
        rad = radius * (1.0 + 0.25 * ring) * rng.uniform(0.92, 1.08)
        height = rng.uniform(1500.0, 2500.0)
        pos = np.array([target[0] + rad * math.cos(ang),
                        target[1] + rad * math.sin(ang), height])
        aim = target + rng.uniform(-150.0, 150.0, size=3)
        R = _look_at(pos, aim)
        fx = rng.uniform(1350.0, 1450.0)
        fy = fx * rng.uniform(0.995, 1.005)
        cx = raw_size[0] / 2 + rng.uniform(-20.0, 20.0)
        cy = raw_size[1] / 2 + rng.uniform(-20.0, 20.0)
        cam = {
            'R': R.tolist(),
            'T': pos.reshape(3, 1).tolist(),
            'K': [[fx, 0.0, cx], [0.0, fy, cy]],
        }
        cam['Kfull'] = cam['K'] + [[0.0, 0.0, 1.0]]
        if distortion:
            cam['k'] = [rng.uniform(-0.33, -0.23), rng.uniform(0.10, 0.20),
                        rng.uniform(-0.05, 0.05)]
            cam['p'] = [rng.uniform(-0.003, 0.003), rng.uniform(-0.003, 0.003)]
        cams.append(cam)
    return cams


def get_scale(raw_size, image_size):
    """``_get_scale`` of the reference's dataset base class."""
    w, h = raw_size
    w_r, h_r = image_size
    if w / w_r < h / h_r:
        w_pad, h_pad = h / h_r * w_r, h
    else:
        w_pad, h_pad = w, w / w_r * h_r
    return np.array([w_pad, h_pad], dtype=np.float32)


def frame_meta(cams, batch, raw_size, image_size):
    """``centers``, ``scales``, ``cameras`` as the callers build them
    (``tempo_detector.py:183-185``): one entry per batch item, per view."""
    center = np.array((raw_size[0] / 2, raw_size[1] / 2), dtype=np.float32)
    scale = get_scale(raw_size, image_size)
    centers = [[center.copy() for _ in cams] for _ in range(batch)]
    scales = [[scale.copy() for _ in cams] for _ in range(batch)]
    cameras = [[dict(c) for c in cams] for _ in range(batch)]
    return centers, scales, cameras


def make_heatmaps(batch, num_views, channels, height, width, kind='blobs',
                  seed=0, device='cpu'):
    """List of ``num_views`` fp32 ``[B,C,H,W]`` tensors with values in [0,1].

    ``blobs``: 1-4 Gaussian blobs per channel (sigma 2-5 px) -- realistic;
    ``noise``: U[0,1) -- worst case for coordinate-rounding sensitivity.
    """
    g = torch.Generator(device='cpu').manual_seed(seed)
    maps = []
    for _ in range(num_views):
        if kind == 'noise':
            hm = torch.rand(batch, channels, height, width, generator=g)
        elif kind == 'blobs':
            nb = 4
            cx = torch.rand(batch, channels, nb, 1, 1, generator=g) * width
            cy = torch.rand(batch, channels, nb, 1, 1, generator=g) * height
            sg = 2.0 + 3.0 * torch.rand(batch, channels, nb, 1, 1, generator=g)
            amp = 0.4 + 0.6 * torch.rand(batch, channels, nb, 1, 1, generator=g)
            keep = (torch.rand(batch, channels, nb, 1, 1, generator=g) <
                    0.75).float()
            keep[:, :, 0] = 1.0
            ys = torch.arange(height, dtype=torch.float32).view(1, 1, 1, height, 1)
            xs = torch.arange(width, dtype=torch.float32).view(1, 1, 1, 1, width)
            hm = (amp * keep * torch.exp(-((xs - cx)**2 + (ys - cy)**2) /
                                         (2 * sg * sg))).sum(2).clamp_(0.0, 1.0)
        else:
            raise ValueError(kind)
        maps.append(hm.contiguous().to(device))
    return maps


def make_proposals(num, space_center, space_size, seed=0, invalid_every=10,
                   margin=0.05):
    """``[num,7]`` proposals: centres uniform in the space, ``[3]`` id (>=0) or
    -1 for every ``invalid_every``-th row, ``[4]`` conf, ``[5:7]`` bbox in [0.3,1]."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    sc = torch.tensor(space_center, dtype=torch.float32)
    ss = torch.tensor(space_size, dtype=torch.float32)
    u = torch.rand(num, 3, generator=g) * (1 - 2 * margin) + margin
    out = torch.zeros(num, 7)
    out[:, :3] = sc - ss / 2 + u * ss
    out[:, 3] = torch.arange(num, dtype=torch.float32)
    if invalid_every:
        out[invalid_every - 1::invalid_every, 3] = -1.0
    out[:, 4] = torch.rand(num, generator=g)
    out[:, 5:7] = 0.3 + 0.7 * torch.rand(num, 2, generator=g)
    return out
