"""Pin the CPU oracle against the reference module itself and write the golden
fixtures from the REFERENCE's outputs.  Build container only.

    python tests/golden/check_against_reference.py            # check + rewrite *.npz
    python tests/golden/check_against_reference.py --check    # check only

What runs: ``/root/reference/tempo/models/project_layer.py`` is loaded where it
lies (``importlib``; nothing is copied).  Its four third-party imports are
absent from this image and from the reference tree (mmcv 1.6.1, mmpose @
dd3c4941, cv2), so ``sys.modules`` entries supply them for the duration of the
load:

* ``mmcv.runner.force_fp32``  -> identity decorator (the module never sets
  ``fp16_enabled``, so mmcv's decorator is a no-op there too);
* ``mmpose.core.camera.SimpleCameraTorch``,
  ``mmpose.core.post_processing.post_transforms.{affine_transform_torch,
  get_affine_transform}``, ``mmpose.models.utils.misc.torch_meshgrid_ij``
  -> the oracle's restatements (``oracle/project_layer_ref.py``).

What this pins: every line of the reference's own file -- ``compute_grid``
(:43-63), ``ProjectLayer.get_voxel`` (:65-152), ``set_space_size`` (:184-201),
``project_grid`` (:225-293), ``ProjectLayerWithMask.get_voxel`` (:295-378) --
i.e. SURVEY section 8 rows a2-a6: the oracle must equal the reference module
BIT FOR BIT (cubes, grids, offset, fine_grid, center_grid) on every case of
``tests/cases.py`` and on BASELINE config 1 at full size.  What it cannot pin:
rows a7-a9, the mmpose/cv2 helpers, which are the same restated code on both
sides here (their only independent evidence is the float64 evaluation in
``tests/test_oracle.py``).

The valid-view mask is an internal of the reference (``bounding``, :77-78,
:113-115), not an output.  It is read from the reference's own execution: the
module-global name ``torch`` of the loaded reference module is wrapped by a
proxy that forwards everything to torch and records the tensor handed to
``torch.sum(bounding, dim=-1)`` (:145).
"""
import argparse
import hashlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_FILE = '/root/reference/tempo/models/project_layer.py'
SOURCE = ('reference tempo/models/project_layer.py executed by '
          'tests/golden/check_against_reference.py (mmcv/mmpose helpers: oracle restatements)')
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

import cases  # noqa: E402
from oracle import project_layer_ref as O  # noqa: E402
from tempo_amd import numerics  # noqa: E402

# BASELINE.json configs[0]: Panoptic cam5, one frame, root grid 80x80x20
CONFIG1 = dict(cfg='panoptic', B=1, cube=[80, 80, 20], hm='blobs', seed=101)
CONFIG1_STRIDE = 5  # the fixture keeps cubes[:, :, ::5, ::5, :]
# BASELINE.json configs[2] geometry, two proposals of one frame pair: the full
# 253x253x64 lattice and 64^3 windows (the reference projects all 4.1 M points)
CONFIG3 = dict(cfg='panoptic', B=2, sub_cube=[64, 64, 64], hm='blobs', seed=103,
               invalid_every=0)
CONFIG3_STRIDE = 4  # the fixture keeps cubes[:, :, ::4, ::4, ::4]


def available():
    return os.path.isfile(REF_FILE)


def load_reference():
    """The reference module object, loaded from REF_FILE with stand-ins for
    its absent third-party imports.  The stand-ins are removed again."""
    def identity_decorator(*_args, **_kwargs):
        return lambda fn: fn

    fake = {}

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        fake[name] = m
        return m

    mod('mmcv')
    mod('mmcv.runner', force_fp32=identity_decorator)
    mod('mmpose')
    mod('mmpose.core')
    mod('mmpose.core.camera', SimpleCameraTorch=O.SimpleCameraTorch)
    mod('mmpose.core.post_processing')
    mod('mmpose.core.post_processing.post_transforms',
        affine_transform_torch=O.affine_transform_torch,
        get_affine_transform=O.get_affine_transform)
    mod('mmpose.models')
    mod('mmpose.models.utils')
    mod('mmpose.models.utils.misc', torch_meshgrid_ij=O.meshgrid_ij)
    saved = {k: sys.modules.get(k) for k in fake}
    sys.modules.update(fake)
    try:
        spec = importlib.util.spec_from_file_location('_tempo_reference_project_layer', REF_FILE)
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    # the reference drops into pdb on a RuntimeError (:285-286); a check must fail instead
    ref.pdb.set_trace = _no_pdb
    return ref


def _no_pdb(*_a, **_k):
    raise RuntimeError('reference reached its pdb.set_trace() trap')


def digest(tensors):
    h = hashlib.sha256()
    for t in tensors:
        h.update(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes())
    return h.hexdigest()


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b, equal_nan=True)


class _TorchProxy:
    """Forwards to torch; keeps the last tensor summed over dim=-1 that is not
    a product (``torch.sum(bounding, dim=-1)``, reference :145)."""

    def __init__(self):
        self.bounding = None

    def __getattr__(self, name):
        return getattr(torch, name)

    def sum(self, x, *args, **kwargs):
        if kwargs.get('dim') == -1 and x.dim() == 5 and x.shape[1] == 1:
            self.bounding = x.detach().clone()
        return torch.sum(x, *args, **kwargs)


def reference_root_with_mask(ref, case):
    """(cubes, grids, mask[B,N] uint32 with bit v = view v in bounds), all from
    one execution of the reference's ProjectLayer."""
    proxy = _TorchProxy()
    ref.torch = proxy
    try:
        layer = ref.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
        cubes, grids = layer(*case['args'])
    finally:
        ref.torch = torch
    b = proxy.bounding  # [B,1,1,N,V] float 0/1
    B, N, V = b.shape[0], b.shape[3], b.shape[4]
    bits = ((b.view(B, N, V) > 0).to(torch.int64) << torch.arange(V)).sum(-1)
    return cubes, grids, bits.numpy().astype(np.uint32)


def run_root(ref, case):
    r_cubes, r_grids, r_mask = reference_root_with_mask(ref, case)
    o_layer = O.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    o_cubes, o_grids = o_layer(*case['args'])
    V = o_layer.last_bounding.shape[-1]
    o_mask = (o_layer.last_bounding.to(torch.int64) << torch.arange(V)).sum(-1)
    o_mask = o_mask.numpy().astype(np.uint32)
    equal = dict(cubes=_same(r_cubes.numpy(), o_cubes.numpy()),
                 grids=_same(r_grids.numpy(), o_grids.numpy()),
                 mask=_same(r_mask, o_mask))
    out = dict(cubes=r_cubes.numpy(), grids=r_grids.numpy(), mask=r_mask,
               inputs_sha256=digest(case['args'][0]))
    return equal, out


def run_window(ref, case):
    r_layer = ref.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    r_cubes, r_offset = r_layer(*case['args'])
    o_layer = O.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    o_cubes, o_offset = o_layer(*case['args'])
    equal = dict(cubes=_same(r_cubes.numpy(), o_cubes.numpy()),
                 offset=_same(r_offset.numpy(), o_offset.numpy()),
                 center_grid=_same(r_layer.center_grid.numpy(), o_layer.center_grid.numpy()),
                 fine_grid=_same(r_layer.fine_grid.numpy(), o_layer.fine_grid.numpy()),
                 scale=_same(r_layer.scale.numpy(), o_layer.scale.numpy()),
                 bias=_same(r_layer.bias.numpy(), o_layer.bias.numpy()),
                 fine_voxels_per_axis=_same(r_layer.fine_voxels_per_axis.numpy(),
                                            o_layer.fine_voxels_per_axis.numpy()))
    out = dict(cubes=r_cubes.numpy(), offset=r_offset.numpy(),
               center_grid=r_layer.center_grid.numpy(),
               inputs_sha256=digest(list(case['args'][0]) + [case['args'][1]]))
    return equal, out


def check_all(write=False, verbose=True):
    """{case: {field: bit-equal?}} for every case; optionally rewrites the fixtures."""
    ref = load_reference()
    torch.manual_seed(0)
    mm = numerics.probe()
    flavour = np.array([mm[s] for s in numerics.SITES])
    report = {}
    for name in cases.ROOT_CASES:
        equal, out = run_root(ref, cases.build_root(name))
        report[name] = equal
        if write:
            np.savez_compressed(os.path.join(HERE, name + '.npz'), mm_mode=flavour,
                                source=np.array(SOURCE), **out)
    for name in cases.WINDOW_CASES:
        equal, out = run_window(ref, cases.build_window(name))
        report[name] = equal
        if write:
            np.savez_compressed(os.path.join(HERE, name + '.npz'), mm_mode=flavour,
                                source=np.array(SOURCE), **out)
    # BASELINE config 1 at full size: compared whole, kept as a strided sample + hash
    case = cases.build_root('config1_full', full=CONFIG1)
    equal, out = run_root(ref, case)
    report['config1_full'] = equal
    if write:
        s = CONFIG1_STRIDE
        np.savez_compressed(
            os.path.join(HERE, 'config1_full.npz'), mm_mode=flavour, source=np.array(SOURCE),
            cubes_strided=out['cubes'][:, :, ::s, ::s, :], stride=np.array(s),
            cubes_sha256=np.array(hashlib.sha256(out['cubes'].tobytes()).hexdigest()),
            cubes_sum=np.array(out['cubes'].astype(np.float64).sum()),
            mask=out['mask'], inputs_sha256=np.array(out['inputs_sha256']))
    case = cases.build_window('config3_windows', full=CONFIG3)
    equal, out = run_window(ref, case)
    report['config3_windows'] = equal
    if write:
        s = CONFIG3_STRIDE
        np.savez_compressed(
            os.path.join(HERE, 'config3_windows.npz'), mm_mode=flavour, source=np.array(SOURCE),
            cubes_strided=out['cubes'][:, :, ::s, ::s, ::s], stride=np.array(s),
            cubes_sha256=np.array(hashlib.sha256(out['cubes'].tobytes()).hexdigest()),
            cubes_sum=np.array(out['cubes'].astype(np.float64).sum()),
            offset=out['offset'], center_grid=out['center_grid'],
            inputs_sha256=np.array(out['inputs_sha256']))
    if verbose:
        for name, equal in report.items():
            print(name, 'bit-equal:', equal)
    return report


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--check', action='store_true', help='compare only, write nothing')
    a = ap.parse_args()
    if not available():
        sys.exit('no reference at ' + REF_FILE)
    report = check_all(write=not a.check)
    bad = {n: e for n, e in report.items() if not all(e.values())}
    if bad:
        sys.exit('oracle differs from the reference: %r' % bad)
    print('oracle == reference module, bit for bit, on %d cases%s'
          % (len(report), '' if a.check else '; fixtures rewritten from the reference outputs'))


if __name__ == '__main__':
    main()
