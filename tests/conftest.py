import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def emul():
    """gcc build of the kernels' per-voxel arithmetic (tests/emul), CPU only."""
    import ctypes
    d = os.path.join(ROOT, 'tests', 'emul')
    so = os.path.join(d, '_emul.so')
    deps = [os.path.join(d, 'emul.c'),
            os.path.join(ROOT, 'tempo_amd', 'csrc', 'unproject_math.h')]
    if not os.path.exists(so) or any(os.path.getmtime(p) > os.path.getmtime(so) for p in deps):
        subprocess.run(['gcc', '-O2', '-ffp-contract=off', '-mfma', '-shared', '-fPIC',
                        deps[0], '-o', so, '-lm'], check=True)
    return ctypes.CDLL(so)


@pytest.fixture(scope='session')
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail('a gpu-marked test was selected but no HIP device is visible')
    return torch.device('cuda:0')
