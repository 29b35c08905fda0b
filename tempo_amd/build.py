"""Build ``tempo_amd/lib/libtempo_unproject.so`` in-tree with hipcc for gfx950.

    python -m tempo_amd.build [--force]

hipcc cross-compiles without a GPU.  ``-ffp-contract=off`` is part of the
numerical contract (see ``csrc/unproject_math.h``): every fused multiply-add
in the coordinate chain is written out as an explicit ``fmaf``.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'csrc', 'tempo_unproject.hip')
DEPS = [SRC, os.path.join(HERE, 'csrc', 'unproject_math.h'),
        os.path.join(HERE, '..', 'include', 'tempo_unproject.h')]
OUT = os.path.join(HERE, 'lib', 'libtempo_unproject.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-ffp-contract=off', '-fPIC', '-shared',
         '-std=c++17', '-Wall']


def hipcc_path():
    for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (need ROCm with gfx950 support)')


def stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=True):
    if not force and not stale():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [hipcc_path()] + FLAGS + [SRC, '-o', OUT]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(OUT)
