"""ctypes binding of ``libtempo_unproject.so`` (ABI: ``include/tempo_unproject.h``).

The library is built in-tree by ``tempo_amd.build`` (``hipcc --offload-arch=gfx950``)
and must be present: there is no CPU or PyTorch fallback for the kernels, a
missing or stale library raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('TEMPO_LIB') or os.path.join(_HERE, 'lib', 'libtempo_unproject.so')
ABI_VERSION = 14

_c_void_p, _c_int, _c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_float

# name -> (restype, argtypes); mirrors include/tempo_unproject.h one to one
SIGNATURES = {
    'tempo_version': (_c_int, []),
    'tempo_error_string': (ctypes.c_char_p, [_c_int]),
    'tempo_padded_channels': (_c_int, [_c_int]),
    'tempo_repack_nhwc': (_c_int, [ctypes.POINTER(_c_void_p)] + [_c_int] * 8 +
                          [_c_void_p, _c_void_p]),
    'tempo_repack_nhwc_half': (_c_int, [ctypes.POINTER(_c_void_p)] + [_c_int] * 9 +
                               [_c_void_p, _c_void_p]),
    'tempo_unproject_root': (_c_int, [_c_void_p] * 5 + [_c_int] + [_c_void_p] * 4 +
                             [_c_int] * 10 + [_c_float] * 4 + [_c_int, _c_void_p]),
    'tempo_unproject_window': (_c_int, [_c_void_p] * 11 + [_c_int] * 13 +
                               [_c_float] * 4 + [_c_int, _c_void_p]),
    'tempo_unproject_root_grouped': (_c_int, [_c_void_p] * 5 + [_c_int] + [_c_void_p] * 4 +
                                     [_c_int] * 10 + [_c_float] * 4 +
                                     [_c_int, _c_void_p, _c_int, _c_void_p]),
    'tempo_unproject_window_grouped': (_c_int, [_c_void_p] * 11 + [_c_int] * 13 +
                                       [_c_float] * 4 +
                                       [_c_int, _c_void_p, _c_int, _c_void_p]),
    'tempo_unproject_window_planes': (_c_int, [_c_void_p] * 14 + [_c_int] * 13 +
                                      [_c_float] * 4 + [_c_int, _c_void_p]),
    'tempo_clamp_mask_words': (_c_int, [_c_int, _c_int]),
    'tempo_unproject_root_train': (_c_int, [_c_void_p] * 5 + [_c_int] + [_c_void_p] * 4 +
                                   [_c_int] * 10 + [_c_float] * 4 +
                                   [_c_int, _c_void_p, _c_void_p]),
    'tempo_unproject_window_train': (_c_int, [_c_void_p] * 11 + [_c_int] * 13 +
                                     [_c_float] * 4 + [_c_int, _c_void_p, _c_void_p]),
    'tempo_unproject_root_backward': (_c_int, [_c_void_p] * 5 + [_c_int] + [_c_void_p] * 3 +
                                      [_c_int] * 9 + [_c_float] * 4 + [_c_int, _c_void_p]),
    'tempo_unproject_window_backward': (_c_int, [_c_void_p] * 12 + [_c_int] * 13 +
                                        [_c_float] * 4 + [_c_int, _c_int, _c_void_p]),
    'tempo_window_planes_claims': (_c_int, [_c_void_p] * 16 + [_c_int] * 13 +
                                   [_c_float] * 4 + [_c_int, _c_void_p]),
    'tempo_unproject_window_planes_backward': (_c_int, [_c_void_p] * 16 + [_c_int] * 13 +
                                               [_c_float] * 4 + [_c_int, _c_int, _c_void_p]),
    'tempo_unrepack_nchw': (_c_int, [_c_void_p, ctypes.POINTER(_c_void_p)] + [_c_int] * 8 +
                            [_c_void_p]),
    'tempo_root_step_workspace': (ctypes.c_longlong, [_c_int] * 6),
    'tempo_root_step': (_c_int, [ctypes.POINTER(_c_void_p), _c_int, _c_int] + [_c_void_p] * 4 +
                        [_c_int] + [_c_void_p] * 4 + [_c_int] * 10 + [_c_float] * 4 +
                        [_c_int, _c_void_p, _c_int, _c_void_p]),
    'tempo_root_step_zmax': (_c_int, [ctypes.POINTER(_c_void_p), _c_int, _c_int] +
                             [_c_void_p] * 4 + [_c_int] + [_c_void_p] * 4 + [_c_int] * 10 +
                             [_c_float] * 4 + [_c_int, _c_void_p, _c_void_p, _c_int, _c_void_p]),
    'tempo_root_step_static': (_c_int, [ctypes.POINTER(_c_void_p), _c_int, _c_int] +
                               [_c_void_p] * 4 + [_c_int] + [_c_void_p] * 4 + [_c_int] * 10 +
                               [_c_float] * 4 + [_c_int, _c_int, _c_void_p]),
    'tempo_unproject_root_views': (_c_int, [ctypes.POINTER(_c_void_p), ctypes.c_longlong, _c_int] +
                                   [_c_void_p] * 4 + [_c_int] + [_c_void_p] * 4 + [_c_int] * 10 +
                                   [_c_float] * 4 + [_c_int, _c_void_p]),
    'tempo_unproject_root_views_zmax': (_c_int, [ctypes.POINTER(_c_void_p), ctypes.c_longlong,
                                                 _c_int] + [_c_void_p] * 4 + [_c_int] +
                                        [_c_void_p] * 4 + [_c_int] * 10 + [_c_float] * 4 +
                                        [_c_int, _c_void_p, _c_void_p]),
    'tempo_unproject_window_views': (_c_int, [ctypes.POINTER(_c_void_p), ctypes.c_longlong,
                                              _c_int] + [_c_void_p] * 10 + [_c_int] * 13 +
                                     [_c_float] * 4 + [_c_int, _c_void_p]),
    'tempo_unproject_window_planes_views': (_c_int, [ctypes.POINTER(_c_void_p), ctypes.c_longlong,
                                                     _c_int] + [_c_void_p] * 13 + [_c_int] * 13 +
                                            [_c_float] * 4 + [_c_int, _c_void_p]),
    'tempo_unproject_root_zmax': (_c_int, [_c_void_p] * 5 + [_c_int] + [_c_void_p] * 4 +
                                  [_c_int] * 10 + [_c_float] * 4 +
                                  [_c_int, _c_void_p, _c_void_p]),
    'tempo_conv1x1_nhwc': (_c_int, [_c_void_p] * 4 + [_c_int] * 8 + [_c_void_p]),
    'tempo_gather_zcolumns': (_c_int, [_c_void_p] * 3 + [_c_int] * 6 + [_c_void_p]),
    'tempo_unproject_root_columns': (_c_int, [_c_void_p] * 5 + [_c_int] + [_c_void_p] * 3 +
                                     [_c_int] * 10 + [_c_float] * 4 + [_c_int, _c_void_p]),
    'tempo_unproject_root_columns_views': (_c_int, [ctypes.POINTER(_c_void_p), ctypes.c_longlong,
                                                    _c_int] + [_c_void_p] * 4 + [_c_int] +
                                           [_c_void_p] * 3 + [_c_int] * 10 + [_c_float] * 4 +
                                           [_c_int, _c_void_p]),
    'tempo_root_step_columns': (_c_int, [ctypes.POINTER(_c_void_p), _c_int, _c_int] +
                                [_c_void_p] * 4 + [_c_int] + [_c_void_p] * 3 + [_c_int] * 10 +
                                [_c_float] * 4 + [_c_int, _c_void_p]),
    'tempo_assemble_zslabs': (_c_int, [_c_void_p, _c_void_p, _c_int, ctypes.c_longlong, _c_int,
                                       _c_void_p]),
    'tempo_window_geometry': (_c_int, [_c_void_p, _c_int, _c_int] + [_c_void_p] * 4 +
                              [_c_int] * 6 + [_c_void_p] * 6),
    'tempo_project_points': (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int] +
                             [_c_float] * 4 + [_c_int] + [_c_void_p] * 4),
}

_lib = None


class TempoLibraryError(RuntimeError):
    pass


def load():
    """Load (once) and return the shared library; raise if it is unusable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TempoLibraryError(
            f'{LIB_PATH} is missing: build it with `python -m tempo_amd.build` '
            '(hipcc, gfx950).  tempo_amd has no fallback path.')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a symbol is not exported
        fn.restype = res
        fn.argtypes = args
    got = lib.tempo_version()
    if got != ABI_VERSION:
        raise TempoLibraryError(f'{LIB_PATH}: ABI version {got}, expected {ABI_VERSION}; rebuild')
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().tempo_error_string(code).decode()
        raise TempoLibraryError(f'{what} failed ({code}): {msg}')
