"""Which rounding flavour does the host's torch use for the path's K=3 matmuls?

The reference evaluates ``X @ R`` (mmpose ``world_to_camera``), ``_X @ K``
(``camera_to_pixel``) and ``torch.mm(trans, pts_homo.T)``
(``affine_transform_torch``) through the CPU BLAS.  With K = 3 the BLAS kernels
differ between machines in how they round (measured, torch 2.10 + oneMKL 2024.2):

* Intel Xeon (AVX-512):  ``fma(a2,b2, fma(a1,b1, a0*b0))``        -> 'fma'
* AMD EPYC:              ``(a0*b0 + a1*b1) + a2*b2`` (no fusion)   -> 'unfused'

A one-ulp difference in a source-pixel coordinate moves a sample on a
white-noise heatmap by up to ~1e-4, ten times the parity tolerance, so the
kernels implement both flavours (csrc/unproject_math.h, ``tempo_dot3``) and
every ABI entry point takes the flavour as ``mm_unfused``.  ``detect()`` measures
the host once (three small matmuls) so that the HIP result is the one the
reference's torch ops give on the same machine; ``TEMPO_MM_MODE=fma|unfused``
pins it instead.
"""
import os
import warnings

import numpy as np
import torch

MM_FMA, MM_UNFUSED = 0, 7  # TEMPO_MM_* of include/tempo_unproject.h
SITES = ('R', 'K', 'A')
_cache = None


def _candidates(a, b):
    """a [N,3], b [3] float32 -> (fma-chain, unfused) float32 [N]."""
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    p = [a64[:, k] * b64[k] for k in range(3)]  # exact: 24+24 bits fit in 53
    f = np.float32
    fma = f(f(f(p[0]).astype(np.float64) + p[1]).astype(np.float64) + p[2])
    unf = f(f(f(p[0]) + f(p[1])) + f(p[2]))
    return fma, unf


def _classify(got, a, b, site):
    fma, unf = _candidates(a, b)
    n = got.shape[0]
    miss_fma = int((got != fma).sum())
    miss_unf = int((got != unf).sum())
    best = 'fma' if miss_fma <= miss_unf else 'unfused'
    if min(miss_fma, miss_unf) > n // 1000:
        warnings.warn(
            f'tempo_amd.numerics: torch CPU matmul at site {site} matches neither known '
            f'rounding flavour (fma: {miss_fma}/{n} off, unfused: {miss_unf}/{n} off); '
            f'using {best!r}.  Results stay within rounding of the reference but the '
            'valid-view mask may differ on voxels that project onto an image border.')
    return best


def probe(n=32768, seed=1234):
    """Measure the three sites on this host; returns {'R','K','A': 'fma'|'unfused'}."""
    g = torch.Generator().manual_seed(seed)
    X = torch.rand(n, 3, generator=g) * 8000 - 4000
    R = torch.randn(3, 3, generator=g)
    out = {}
    got = (X @ R).numpy()
    out['R'] = _classify(got[:, 0], X.numpy(), R[:, 0].numpy(), 'R')
    K = torch.randn(2, 3, generator=g) * 1000
    got = (X @ K.T).numpy()
    out['K'] = _classify(got[:, 1], X.numpy(), K[1].numpy(), 'K')
    t = torch.randn(2, 3, generator=g)
    homo = torch.cat([X[:, :2], torch.ones(n, 1)], dim=1)
    got = torch.mm(t, torch.t(homo)).numpy()
    # site A multiplies t[i][k] * homo[n][k]; products commute exactly
    out['A'] = _classify(got[0], homo.numpy(), t[0].numpy(), 'A')
    return out


def detect():
    """Host flavour per site, cached; honours TEMPO_MM_MODE."""
    global _cache
    env = os.environ.get('TEMPO_MM_MODE', 'auto').lower()
    if env in ('fma', 'unfused'):
        return {s: env for s in SITES}
    if env != 'auto':
        raise ValueError(f'TEMPO_MM_MODE={env!r}: expected auto, fma or unfused')
    if _cache is None:
        _cache = probe()
    return dict(_cache)


def resolve(mode):
    """'auto' | 'fma' | 'unfused' | {'R':..,'K':..,'A':..} -> per-site dict."""
    if mode is None or mode == 'auto':
        return detect()
    if isinstance(mode, str):
        if mode not in ('fma', 'unfused'):
            raise ValueError(f'mm_mode={mode!r}: expected auto, fma or unfused')
        return {s: mode for s in SITES}
    return {s: mode[s] for s in SITES}


def abi_flag(mode):
    """``mm_unfused`` argument of the C ABI for a mode as accepted by resolve().
    The kernels are compiled for one flavour across the three sites (what every
    host measured so far shows); a mixed result falls back to the majority."""
    m = resolve(mode)
    n_unf = sum(1 for s in SITES if m[s] == 'unfused')
    if n_unf not in (0, len(SITES)):
        warnings.warn(f'tempo_amd.numerics: matmul sites disagree ({m}); using the majority '
                      'flavour for all three')
    return MM_UNFUSED if 2 * n_unf > len(SITES) else MM_FMA
