"""Where one wave of the root kernel spends its time: per-phase time stamps (s_memtime)
written by a measurement build of the library.

    hipcc ... -DTEMPO_TIMELINE tempo_amd/csrc/tempo_unproject.hip -o build_dev/timeline.so
    TEMPO_LIB=$PWD/build_dev/timeline.so python tools/timeline.py [--layout nchw|nhwc] [--frames 8]

Stamps per wave (lane 0): 0 entry, 1 voxel coordinates landed, 2 first view projected; per
view slot s: 3+3s fetches issued, 4+3s next view projected, 5+3s mixed; 27 loop done,
28 normalised + clamped + transposed, 29 stores issued, 30 stores drained; 31 = HW ids.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--layout', default='nchw')
    ap.add_argument('--frames', type=int, default=8)
    a = ap.parse_args()
    args = bench.parse(['--workload', 'root', '--batch', str(a.frames), '--input-layout',
                             a.layout, '--chunk', str(a.frames), '--no-cpu-baseline'])
    dev = torch.device('cuda:0')
    wl = bench.RootWorkload(args, dev, 0)
    ops, cfg = wl.ops, wl.cfg
    B, N, V = wl.B, wl.N, wl.V
    grids = torch.zeros(B, N, 3, device=dev)
    if not wl.nhwc:
        wl.repack(0, B)

    def launch(g):
        if wl.nhwc:  # (allocates its own grids: zero-filled by the caching allocator's reuse)
            lay = ops.channels_last_layout(wl.fms)
            return ops.unproject_root_views(lay, wl.cam, wl.axes, wl.J, cfg['heatmap_size'],
                                            cfg['image_size'], out=wl.out, want_grids=True)[1]
        return ops.unproject_root(wl.hm[:B], wl.cam, wl.axes, wl.J, cfg['heatmap_size'],
                                  cfg['image_size'], out=wl.out, want_grids=True,
                                  grids_out=g)[1]
    for _ in range(3):
        launch(grids)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    grids.zero_()
    if not wl.nhwc:
        wl.repack(0, B)  # the state the chunked step launches in: maps fresh in the cache
    e0.record(); grids = launch(grids); e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    t = grids.reshape(-1).view(torch.int64).view(B * N // 64, 96)[:, :32].cpu().numpy().astype(np.int64)
    hw = t[:, 31]
    xcc = ((hw >> 32) & 0xf).astype(np.int64)
    # every XCD counts on its own clock: times relative to the XCD's first wave
    for x in np.unique(xcc):
        m = xcc == x
        t[m, :31] -= t[m, 0].min()
    t0 = t[:, 0]
    span = int(t[:, 30].max())
    tick_ns = ms * 1e6 / span
    print(f'launch {ms * 1e3:.1f} us by HIP events; longest XCD span {span} ticks -> '
          f'{tick_ns:.3f} ns per tick ({1e3 / tick_ns:.0f} MHz)')
    life = t[:, 30] - t0
    seg = {}
    seg['prologue (index, axis loads)'] = t[:, 1] - t[:, 0]
    seg['project view 0 (nothing in flight)'] = t[:, 2] - t[:, 1]
    prev = t[:, 2]
    fi, pa, mx = 0, 0, 0
    for s in range(V):
        fi = fi + (t[:, 3 + 3 * s] - prev)
        pa = pa + (t[:, 4 + 3 * s] - t[:, 3 + 3 * s])
        mx = mx + (t[:, 5 + 3 * s] - t[:, 4 + 3 * s])
        prev = t[:, 5 + 3 * s]
    seg[f'issue fetches ({V} views)'] = fi
    seg['project next view under the fetches'] = pa
    seg['wait for texels + mix'] = mx
    seg['normalise, clamp, transpose'] = t[:, 28] - t[:, 27]
    seg['issue stores'] = t[:, 29] - t[:, 28]
    seg['stores drain'] = t[:, 30] - t[:, 29]
    print(f'waves {len(t)}; wave lifetime mean {life.mean():.0f} ticks = '
          f'{life.mean() * tick_ns / 1e3:.2f} us (p10 {np.percentile(life, 10):.0f}, '
          f'p90 {np.percentile(life, 90):.0f})')
    tot = 0
    for k, v in seg.items():
        tot += v.mean()
        print(f'  {k:42s} {v.mean():8.0f} ticks  {100 * v.mean() / life.mean():5.1f} %')
    print(f'  {"(sum)":42s} {tot:8.0f}')
    # residency: waves alive (all XCDs, each on its own clock) at sampled instants
    samples = np.linspace(0, span, 42)[1:-1]
    alive = [int(((t0 <= x) & (t[:, 30] > x)).sum()) for x in samples]
    print('waves alive at 40 instants across the launch (4096 slots at 4 per SIMD):', alive)
    fetching = [int(sum((((t[:, 2] if s == 0 else t[:, 5 + 3 * (s - 1)]) <= x) &
                         (t[:, 5 + 3 * s] > x)).sum() for s in range(V))) for x in samples]
    print('waves inside the view loop at those instants:', fetching)
    print('waves per XCC id:', np.bincount(xcc).tolist())
    print('wave start (ticks after its XCD\'s first): p25 %d p50 %d p75 %d p99 %d' % tuple(
        np.percentile(t0, [25, 50, 75, 99])))
    per = []
    for s in range(V):
        prev = t[:, 2] if s == 0 else t[:, 5 + 3 * (s - 1)]
        per.append('%d/%d/%d' % ((t[:, 3 + 3 * s] - prev).mean(),
                                 (t[:, 4 + 3 * s] - t[:, 3 + 3 * s]).mean(),
                                 (t[:, 5 + 3 * s] - t[:, 4 + 3 * s]).mean()))
    print('per slot, issue / project next / wait+mix:', '  '.join(per))


if __name__ == '__main__':
    main()
