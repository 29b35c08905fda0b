"""Regenerate tests/golden/*.npz with the CPU oracle (oracle/project_layer_ref.py).

    python tests/golden/make_golden.py

The reference ships no golden vectors for this path and cannot be imported in
the build container (mmcv / mmpose / cv2 are not installed), so these fixtures
are outputs of the oracle restatement itself: PARITY UNPINNED against the
reference, pinned against regressions of the oracle.  Inputs are regenerated
from seeds by tests/cases.py; each file stores the expected outputs plus a
checksum of the inputs so a drift in the generators is detected.
"""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import cases  # noqa: E402
from oracle import project_layer_ref as O  # noqa: E402
from tempo_amd import numerics  # noqa: E402


def digest(tensors):
    h = hashlib.sha256()
    for t in tensors:
        h.update(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes())
    return h.hexdigest()


def run_root(name):
    case = cases.build_root(name)
    layer = O.ProjectLayer(case['image_size'], case['heatmap_size'], 'mean')
    cubes, grids = layer(*case['args'])
    mask = layer.last_bounding  # [B,N,V] bool
    V = mask.shape[-1]
    bits = (mask.to(torch.int32) << torch.arange(V, dtype=torch.int32)).sum(-1)
    return dict(cubes=cubes.numpy(), grids=grids.numpy(),
                mask=bits.to(torch.int32).numpy().astype(np.uint32),
                inputs_sha256=digest(case['args'][0]))


def run_window(name):
    case = cases.build_window(name)
    layer = O.ProjectLayerWithMask(case['image_size'], case['heatmap_size'], 'mean')
    cubes, offset = layer(*case['args'])
    return dict(cubes=cubes.numpy(), offset=offset.numpy(),
                center_grid=layer.center_grid.numpy(),
                inputs_sha256=digest(list(case['args'][0]) + [case['args'][1]]))


def main():
    torch.manual_seed(0)
    # the K=3 matmuls of the oracle go through this host's BLAS: record which
    # rounding flavour produced the files (tempo_amd/numerics.py)
    mm = numerics.probe()
    flavour = np.array([mm[s] for s in numerics.SITES])
    print('matmul flavour on this host:', mm)
    for name in cases.ROOT_CASES:
        out = run_root(name)
        np.savez_compressed(os.path.join(HERE, name + '.npz'), mm_mode=flavour, **out)
        print(name, out['cubes'].shape, float(out['cubes'].mean()))
    for name in cases.WINDOW_CASES:
        out = run_window(name)
        np.savez_compressed(os.path.join(HERE, name + '.npz'), mm_mode=flavour, **out)
        print(name, out['cubes'].shape, float(out['cubes'].mean()))


if __name__ == '__main__':
    main()
