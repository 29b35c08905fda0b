"""The C-ABI library loads without a GPU, exports every symbol the header
declares, and rejects bad arguments before touching the device."""
import ctypes
import os
import re

from tempo_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'tempo_unproject.h')


def _declared():
    text = re.sub(r'/\*.*?\*/', '', open(HEADER).read(), flags=re.S)
    return sorted(set(re.findall(r'\b(tempo_[a-z0-9_]+)\s*\(', text)))


def test_header_and_binding_agree():
    names = _declared()
    assert names == sorted(_lib.SIGNATURES), (names, sorted(_lib.SIGNATURES))
    text = re.sub(r'/\*.*?\*/', '', open(HEADER).read(), flags=re.S)
    for name, (_, args) in _lib.SIGNATURES.items():
        decl = re.search(name + r'\s*\((.*?)\)\s*;', text, flags=re.S).group(1).strip()
        n = 0 if decl == 'void' else len(decl.split(','))
        assert n == len(args), (name, n, len(args))


def test_library_exports_every_symbol():
    lib = _lib.load()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert getattr(raw, name) is not None
    assert lib.tempo_version() == _lib.ABI_VERSION
    assert lib.tempo_error_string(0) == b'ok'
    assert b'NULL' in lib.tempo_error_string(-1)


def test_padded_channels():
    lib = _lib.load()
    assert [lib.tempo_padded_channels(c) for c in (1, 15, 16, 17, 20, 21, 32)] == \
        [16, 16, 16, 20, 20, 24, 32]
    assert lib.tempo_padded_channels(0) < 0 and lib.tempo_padded_channels(33) < 0


def test_bad_arguments_are_rejected_before_any_launch():
    lib = _lib.load()
    P = ctypes.c_void_p
    one = P(16)  # never dereferenced: every call below must fail validation first
    root = lambda **kw: lib.tempo_unproject_root(
        kw.get('hm', one), one, one, one, one, 0, None, kw.get('cubes', one), None,
        kw.get('mask', None), kw.get('mask_bytes', 0), kw.get('B', 1), kw.get('V', 5),
        kw.get('C', 15), kw.get('CP', 16), 128, 240, 80, 80, kw.get('Z', 20), 240.0, 128.0,
        960.0, 512.0, kw.get('mm', 0), None)
    assert root(hm=None) == -1 and root(cubes=None) == -1
    assert root(V=0) == -2 and root(V=33) == -2 and root(C=33) == -2 and root(Z=0) == -2
    assert root(CP=12) == -3 and root(CP=18) == -3 and root(C=17, CP=16) == -3
    assert root(mask=one, mask_bytes=0) == -5 and root(mask_bytes=1) == -5
    assert root(mask=one, mask_bytes=1, V=9, CP=16) == -5
    assert root(mm=3) == -5
    win = lambda **kw: lib.tempo_unproject_window(
        one, one, kw.get('wf', None), one, one, one, kw.get('tl', one), one, one, None, one,
        kw.get('NW', 4), kw.get('F', 4), 5, 15, 16, 128, 240, 253, 253, 64, 64, 64, 64,
        240.0, 128.0, 960.0, 512.0, 0, None)
    assert win(tl=None) == -1
    assert win(NW=8, F=4) == -5  # more windows than frames needs win_frame
    assert win(NW=0) == -2
    ptrs = (ctypes.c_void_p * 2)(16, 16)
    assert lib.tempo_repack_nhwc(ptrs, 2, 1, 15, 0, 15, 8, 8, 16, None, None) == -1
    assert lib.tempo_repack_nhwc(ptrs, 2, 1, 15, 4, 15, 8, 8, 16, one, None) == -2
    # the 16-bit repack: dtype 1 (fp16) or 2 (bf16) only, pointers 2-byte aligned
    half = lambda dtype, **kw: lib.tempo_repack_nhwc_half(
        kw.get('ptrs', ptrs), dtype, 2, 1, 15, kw.get('c0', 0), 15, 8, 8, kw.get('CP', 16),
        kw.get('out', one), None)
    assert half(0) == -5 and half(3) == -5 and half(-1) == -5
    assert half(1, out=None) == -1 and half(2, c0=4) == -2 and half(1, CP=12) == -3
    assert half(1, ptrs=(ctypes.c_void_p * 2)(16, 17)) == -5
    assert half(2, ptrs=(ctypes.c_void_p * 2)(16, None)) == -1
    assert half(1, out=ctypes.c_void_p(24)) == -5  # out is written with 128-bit stores
    assert lib.tempo_project_points(one, one, 0, 5, 1., 1., 1., 1., 0, one, one, one, None) == -2


def test_new_entry_points_reject_bad_arguments():
    lib = _lib.load()
    P = ctypes.c_void_p
    one = P(16)
    planes = lambda **kw: lib.tempo_unproject_window_planes(
        one, one, None, one, one, one, one, one, one, None, kw.get('cubes', None),
        kw.get('xy', one), None, None, kw.get('NW', 4), 4, 5, 15, kw.get('CP', 16), 128, 240, 253,
        253, 64, 64, 64, 64, 240.0, 128.0, 960.0, 512.0, kw.get('mm', 0), None)
    assert planes(xy=None) == -1            # no output at all
    assert planes(NW=0) == -2 and planes(CP=18) == -3 and planes(mm=3) == -5
    grouped = lambda **kw: lib.tempo_unproject_root_grouped(
        one, one, one, one, one, 0, None, one, None, None, 0, 4, 5, 15, 16, 128, 240, 80, 80, 20,
        240.0, 128.0, 960.0, 512.0, 0, kw.get('groups', one), kw.get('ng', 2), None)
    assert grouped(groups=None, ng=2) == -5 and grouped(ng=0) == -5 and grouped(ng=5) == -5
    train = lambda **kw: lib.tempo_unproject_root_train(
        one, one, one, one, one, 0, None, one, None, None, 0, 1, 5, 15, 16, 128, 240, 80, 80, 20,
        240.0, 128.0, 960.0, 512.0, 0, kw.get('cmask', one), None)
    assert train(cmask=None) == -1
    bwd = lambda **kw: lib.tempo_unproject_root_backward(
        kw.get('g', one), one, one, one, one, 0, None, kw.get('cmask', one), kw.get('ghm', one),
        1, kw.get('V', 5), 15, 16, 128, 240, 80, 80, 20, 240.0, 128.0, 960.0, 512.0,
        kw.get('mm', 0), None)
    assert bwd(g=None) == -1 and bwd(cmask=None) == -1 and bwd(ghm=None) == -1
    assert bwd(mm=1) == -5
    assert bwd(V=33) == -2                  # beyond TEMPO_MAX_VIEWS (13+ views: several launches)
    wbwd = lambda **kw: lib.tempo_unproject_window_backward(
        one, one, None, one, one, one, one, one, one, None, one, kw.get('ghm', one),
        kw.get('NW', 4), 4, 5, 15, 16, 128, 240, 253, 253, 64, 64, 64, 64, 240.0, 128.0, 960.0,
        512.0, 0, kw.get('tile', 0), None)
    assert wbwd(ghm=None) == -1 and wbwd(NW=8) == -5
    assert wbwd(tile=100) == -5 and wbwd(tile=-1) == -5  # 0 (default), 121 or 144
    # training through the planes: all six plane / claim pointers, the tile, the usual limits
    def claims(**kw):
        a = dict(hm=one, xy=one, xz=one, yz=one, cxy=one, cxz=one, cyz=one, NW=4, CP=16, VZ=64,
                 mm=0)
        a.update(kw)
        return lib.tempo_window_planes_claims(
            a['hm'], one, None, one, one, one, one, one, one, None, a['xy'], a['xz'], a['yz'],
            a['cxy'], a['cxz'], a['cyz'], a['NW'], 4, 5, 15, a['CP'], 128, 240, 253, 253, 64, 64,
            64, a['VZ'], 240.0, 128.0, 960.0, 512.0, a['mm'], None)
    for k in ('hm', 'xy', 'xz', 'yz', 'cxy', 'cxz', 'cyz'):
        assert claims(**{k: None}) == -1, k
    assert claims(NW=0) == -2 and claims(NW=8) == -5 and claims(CP=18) == -3
    assert claims(mm=3) == -5 and claims(VZ=1 << 23) == -2 and claims(hm=P(20)) == -5

    def pbwd(**kw):
        a = dict(gxy=one, gxz=one, gyz=one, cxy=one, cxz=one, cyz=one, ghm=one, NW=4, tile=0,
                 W=240, mm=0)
        a.update(kw)
        return lib.tempo_unproject_window_planes_backward(
            a['gxy'], a['gxz'], a['gyz'], a['cxy'], a['cxz'], a['cyz'], one, None, one, one, one,
            one, one, one, None, a['ghm'], a['NW'], 4, 5, 15, 16, 128, a['W'], 253, 253, 64, 64,
            64, 64, 240.0, 128.0, 960.0, 512.0, a['mm'], a['tile'], None)
    for k in ('gxy', 'gxz', 'gyz', 'cxy', 'cxz', 'cyz', 'ghm'):
        assert pbwd(**{k: None}) == -1, k
    assert pbwd(tile=100) == -5 and pbwd(tile=64) == -5  # 0 (= 81), 81, 121 or 144
    assert pbwd(NW=8) == -5 and pbwd(mm=1) == -5
    assert pbwd(ghm=P(20)) == -5 and pbwd(NW=0) == -2
    assert lib.tempo_clamp_mask_words(128000, 16) == 2000 * 16
    assert lib.tempo_clamp_mask_words(65, 20) == 2 * 24
    assert lib.tempo_clamp_mask_words(0, 16) == -2 and lib.tempo_clamp_mask_words(64, 18) == -3
    ptrs = (ctypes.c_void_p * 2)(16, 16)
    assert lib.tempo_unrepack_nchw(None, ptrs, 2, 1, 15, 0, 15, 8, 8, 16, None) == -1
    assert lib.tempo_unrepack_nchw(one, ptrs, 2, 1, 15, 4, 15, 8, 8, 16, None) == -2


def test_misaligned_channel_last_buffers_are_rejected():
    """hm / workspace / the repack's out are accessed with 128-bit loads and stores."""
    lib = _lib.load()
    P = ctypes.c_void_p
    one, odd = P(16), P(20)
    ptrs = (ctypes.c_void_p * 2)(16, 16)
    assert lib.tempo_repack_nhwc(ptrs, 2, 1, 15, 0, 15, 8, 8, 16, odd, None) == -5
    assert lib.tempo_unproject_root(odd, one, one, one, one, 0, None, one, None, None, 0, 1, 5,
                                    15, 16, 128, 240, 8, 8, 8, 240.0, 128.0, 960.0, 512.0, 0,
                                    None) == -5
    assert lib.tempo_unproject_window(odd, one, None, one, one, one, one, one, one, None, one,
                                      4, 4, 5, 15, 16, 128, 240, 253, 253, 64, 64, 64, 64,
                                      240.0, 128.0, 960.0, 512.0, 0, None) == -5
    assert lib.tempo_root_step(ptrs, 15, 0, one, one, one, one, 0, None, one, None, None, 0, 1,
                               2, 15, 16, 8, 8, 4, 4, 4, 8.0, 8.0, 8.0, 8.0, 0, odd, 1,
                               None) == -5


def test_window_geometry_rejects_bad_arguments():
    lib = _lib.load()
    P = ctypes.c_void_p
    one = P(16)
    geo = lambda **kw: lib.tempo_window_geometry(
        kw.get('gc', one), kw.get('B', 4), kw.get('cols', 7), one, one, one, one,
        kw.get('FX', 253), 253, 64, 64, 64, 64, kw.get('tl', one), one, one, one, None, None)
    assert geo(gc=None) == -1 and geo(tl=None) == -1
    assert geo(B=0) == -2 and geo(cols=5) == -2 and geo(FX=1) == -2


def test_fused_consumer_entry_points_reject_bad_arguments():
    """tempo_root_step_zmax, tempo_unproject_root_views_zmax and
    tempo_unproject_window_planes_views: missing outputs / inputs are refused before anything
    touches the GPU."""
    lib = _lib.load()
    P = ctypes.c_void_p
    one = P(16)
    ptrs = (ctypes.c_void_p * 5)(*([16] * 5))
    step = lambda zmax, ws=one: lib.tempo_root_step_zmax(
        ptrs, 15, 0, one, one, one, one, 0, None, one, None, None, 0, 2, 5, 15, 16, 8, 8, 4, 4, 4,
        8.0, 8.0, 32.0, 32.0, 0, zmax, ws, 2, None)
    assert step(None) == -1           # no z-max plane: use tempo_root_step
    assert step(one, ws=None) == -1   # no workspace
    views = lambda zmax, v=ptrs: lib.tempo_unproject_root_views_zmax(
        v, 8 * 8 * 64, 64, one, one, one, one, 0, None, one, None, None, 0, 2, 5, 15, 16, 8, 8,
        4, 4, 4, 8.0, 8.0, 32.0, 32.0, 0, zmax, None)
    assert views(None) == -1 and views(one, v=None) == -1
    planes = lambda v=ptrs, cubes=None, pxy=one: lib.tempo_unproject_window_planes_views(
        v, 8 * 8 * 64, 64, one, None, one, one, one, one, one, one, None, cubes, pxy, pxy, pxy,
        2, 2, 5, 15, 16, 8, 8, 9, 9, 9, 4, 4, 4, 8.0, 8.0, 32.0, 32.0, 0, None)
    assert planes(v=None) == -1
    assert planes(pxy=None) == -1     # neither the volume nor a plane asked for


def test_volume_free_root_entry_points_reject_bad_arguments():
    """cubes may be NULL only together with the z-max planes; the z-column entry points
    refuse missing inputs / outputs and bad sizes before anything touches the GPU."""
    lib = _lib.load()
    P = ctypes.c_void_p
    one = P(16)
    ptrs = (ctypes.c_void_p * 5)(*([16] * 5))
    zmax = lambda cubes, z: lib.tempo_unproject_root_zmax(
        one, one, one, one, one, 0, None, cubes, None, None, 0, 2, 5, 15, 16, 8, 8, 4, 4, 4,
        8.0, 8.0, 32.0, 32.0, 0, z, None)
    assert zmax(None, None) == -1 and zmax(one, None) == -1
    step = lambda cubes, z: lib.tempo_root_step_zmax(
        ptrs, 15, 0, one, one, one, one, 0, None, cubes, None, None, 0, 2, 5, 15, 16, 8, 8, 4, 4,
        4, 8.0, 8.0, 32.0, 32.0, 0, z, one, 2, None)
    assert step(None, None) == -1
    plain = lib.tempo_root_step(ptrs, 15, 0, one, one, one, one, 0, None, None, None, None, 0, 2,
                                5, 15, 16, 8, 8, 4, 4, 4, 8.0, 8.0, 32.0, 32.0, 0, one, 2, None)
    assert plain == -1                # no volume and no z-max: nothing to compute
    cols = lambda **kw: lib.tempo_unproject_root_columns(
        kw.get('hm', one), one, one, one, one, 0, None, kw.get('index', one), kw.get('out', one),
        2, kw.get('K', 3), 5, 15, kw.get('CP', 16), 8, 8, 4, 4, 4, 8.0, 8.0, 32.0, 32.0,
        kw.get('mm', 0), None)
    assert cols(hm=None) == -1 and cols(index=None) == -1 and cols(out=None) == -1
    assert cols(K=0) == -2 and cols(CP=18) == -3 and cols(mm=3) == -5
    assert cols(hm=P(20)) == -5       # the repacked buffer is read with 128-bit loads
    colv = lambda v=ptrs, tb=64: lib.tempo_unproject_root_columns_views(
        v, 8 * 8 * 64, tb, one, one, one, one, 0, None, one, one, 2, 3, 5, 15, 16, 8, 8, 4, 4,
        4, 8.0, 8.0, 32.0, 32.0, 0, None)
    assert colv(v=None) == -1 and colv(tb=56) == -5   # texels narrower than the C channels
    coln = lambda v=ptrs, ct=15, c0=0: lib.tempo_root_step_columns(
        v, ct, c0, one, one, one, one, 0, None, one, one, 2, 3, 5, 15, 16, 8, 8, 4, 4, 4, 8.0,
        8.0, 32.0, 32.0, 0, None)
    assert coln(v=None) == -1 and coln(ct=14) == -2 and coln(ct=16, c0=2) == -2
