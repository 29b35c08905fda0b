#!/usr/bin/env python3
"""Headline benchmark: voxel-view heatmap samples/s of the root-grid unprojection.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload root|window]

Workload (BASELINE.json configs[1]): Panoptic cam5 root grid 80x80x20, batch 32
frames, V=5 views, J=15 joints, 128x240 heatmaps, fp32, per GPU.  One step = one
pass of the hot path over one batch: NCHW -> channel-last repack of the V
heatmap tensors + the unprojection kernel, through the C ABI.  Inputs (heatmaps,
packed camera table, grid axes) are resident in HBM before the timed region.
One sample = one (voxel, view) pair producing all J channel values: B*N*V.

With N > 1 every rank processes its own 32 frames (frames x proposals partition, no
data-path collective): weak scaling; value = samples of all ranks / max-over-ranks time.
`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself:
the parent, which never touches the GPU, runs `python -m torch.distributed.run
--nproc-per-node N bench.py ...` as a child process (what the reference's
tools/dist_train.sh:12-20 does with torch.distributed.launch) and relays rank 0's line.
When the driver has already launched the ranks (WORLD_SIZE set) nothing is spawned.

The timed region is `--blocks` blocks (default 5) of exactly `--steps` steps, each
bracketed by barrier + synchronize; the MEDIAN block (max over ranks within a block) is
reported -- one block of the default 50 steps lasts only ~15 ms.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--workload', default='root',
                    choices=['root', 'window', 'window_h36m', 'dense'])
    ap.add_argument('--timesteps', type=int, default=None,
                    help='window workloads: frames per sequence; the T windows of one proposal '
                         'share geometry and calibration (default 1; window_h36m: 4)')
    ap.add_argument('--batch', type=int, default=32, help='frames per GPU')
    ap.add_argument('--proposals', type=int, default=10, help='window workload only')
    ap.add_argument('--calibrations', type=int, default=1,
                    help='distinct camera calibrations in the batch (recordings the frames '
                         'come from); frames of one calibration are consecutive')
    ap.add_argument('--share', action='store_true',
                    help='grouped launch: one projection per run of frames / timesteps with '
                         'one calibration (measured slower than per-item launches on MI355X)')
    ap.add_argument('--group-max', type=int, default=None, help='frames per shared projection')
    ap.add_argument('--planes', default='off', choices=['off', 'with', 'only'],
                    help='window workloads: fused axis-max planes (tempo_pose.py:172-176) '
                         'next to the volume, or instead of it')
    ap.add_argument('--zmax', default='off', choices=['off', 'with', 'only'],
                    help='root workload: the detector\'s consumers fused in -- the z-max planes '
                         '(center_net.py:230) next to the volume, or `only`: z-max planes with '
                         'cubes = NULL plus the K z-columns (tempo_detector.py:204-206) computed '
                         'from the heatmaps, so that the volume is never written')
    ap.add_argument('--columns', type=int, default=10,
                    help='--zmax only: z-columns per frame (the detector\'s top-K cells)')
    ap.add_argument('--min-seconds', type=float, default=3.0,
                    help='the number of timed blocks is raised until the timed region lasts at '
                         'least this long (steps per block stay as given)')
    ap.add_argument('--backward', action='store_true',
                    help='also time the training step (forward with clamp mask + backward + '
                         'un-repack) and report it beside the headline')
    ap.add_argument('--tile', type=int, default=None,
                    help='window backward: texels per LDS tile (default: chosen from the lattice '
                         'pitch like the drop-in module does)')
    ap.add_argument('--chunk', type=int, default=None,
                    help='root workload: frames per repack+unproject chunk (one reused '
                         'channel-last buffer that stays in the Infinity Cache)')
    ap.add_argument('--window-chunk', type=int, default=None,
                    help='window workloads: frames per chunk (default: cache-sized, about 40 '
                         'windows per launch)')
    ap.add_argument('--channels', type=int, default=None,
                    help='heatmap channels instead of the config\'s joint count (TEMPO\'s own '
                         'configs unproject 32 learned features)')
    ap.add_argument('--input-layout', default='nchw', choices=['nchw', 'nhwc', 'nhwc-tight'],
                    help='nchw: V [B,C,H,W] tensors as the reference has them, repacked inside '
                         'the step; nhwc: torch.channels_last tensors whose channel count is '
                         'padded to a multiple of 4 (what a channel-last 2-D head such as '
                         'tempo_conv1x1_nhwc writes), read in place, no repack; nhwc-tight: '
                         'channels_last with exactly C channels (60-byte texels at 15 joints)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--blocks', type=int, default=5,
                    help='timed blocks of --steps steps each; the median block is reported')
    ap.add_argument('--dry-run', action='store_true',
                    help='no GPU: ranks rendezvous over gloo and run an empty workload '
                         '(exercises the launcher and the timing protocol on a CPU box)')
    ap.add_argument('--no-module', action='store_true',
                    help='skip the module-level (drop-in nn.Module.forward) timing')
    return ap.parse_args(argv)


def make_inputs(args, dev, seed, cfg=None):
    """Synthetic, seeded: U[0,1) heatmaps generated on the device, ring cameras
    (Panoptic-like) or the four H36M S1 cameras."""
    from tempo_amd import synthetic as S
    from tempo_amd.camera import pack_cameras
    cfg = S.PANOPTIC if cfg is None else cfg
    if getattr(args, 'channels', None):
        import copy
        h36m = cfg is S.H36M
        cfg = copy.deepcopy(cfg)
        cfg['num_joints'] = args.channels
        cfg['_h36m'] = h36m
    B, V, J = args.batch, cfg['num_views'], cfg['num_joints']
    w, h = cfg['heatmap_size']
    g = torch.Generator(device=dev).manual_seed(seed)
    fms = [torch.rand(B, J, h, w, device=dev, generator=g) for _ in range(V)]
    ncal = max(1, min(args.calibrations, B))
    centers, scales, cameras = [], [], []
    for k in range(ncal):
        cams = (S.h36m_cameras() if (cfg is S.H36M or cfg.get('_h36m')) else
                S.ring_cameras(V, seed=seed + 1000 * k, target=cfg['space_center']))
        nb = (B * (k + 1)) // ncal - (B * k) // ncal
        c, s, m = S.frame_meta(cams, nb, cfg['raw_size'], cfg['image_size'])
        centers += c
        scales += s
        cameras += m
    cam_host = pack_cameras(centers, scales, cameras, cfg['image_size'])
    make_inputs.meta = (centers, scales, cameras)  # the lists of dicts the module is called with
    return cfg, fms, cam_host.to(dev), cam_host


def gpu_blocker(dev, ms=12.0):
    """Keeps the GPU busy for about `ms` so that the launches timed next are all enqueued
    before the first one starts (event intervals then hold GPU time only, no host gaps)."""
    buf = torch.empty(1 << 28, dtype=torch.float32, device=dev)  # 1 GiB
    for _ in range(max(1, int(ms / 0.3))):
        buf.zero_()
    return buf


class RootWorkload:
    name = 'panoptic_cam5_root_80x80x20_b32'
    limiter = ('texture path (TA/TCP), saturated in steady state: 4 x 64-byte texel accesses '
               'per sample at 0.87 (L1 hit) / 0.41 (L1 miss, L2 hit) accesses per clock and CU; '
               'see roofline.attainable and DESIGN.md section 4')

    def __init__(self, args, dev, seed):
        from tempo_amd import ops, project_layer as PL
        self.ops = ops
        self.dev = dev
        self.cfg, self.fms, self.cam, cam_host = make_inputs(args, dev, seed)
        self.meta = make_inputs.meta
        cfg = self.cfg
        self.B, self.V, self.J = args.batch, cfg['num_views'], cfg['num_joints']
        # frames with bit-identical camera records share one projection per voxel
        self.groups = None
        if args.share:
            g = ops.share_groups(ops.camera_keys(cam_host), args.group_max)
            self.groups = None if g is None else g.to(dev)
        self.axes = [(PL._base_axis(cfg['space_size'][i], cfg['cube_size'][i]) +
                      cfg['space_center'][i]).to(dev) for i in range(3)]
        X, Y, Z = cfg['cube_size']
        self.N = X * Y * Z
        w, h = cfg['heatmap_size']
        cp = ops.padded_channels(self.J)
        self.input_layout = args.input_layout
        self.nhwc = args.input_layout != 'nchw'
        if self.nhwc:
            if self.groups is not None:
                raise SystemExit('--share works on repacked maps only')
            # what a channel-last 2-D head hands over: the same values in NHWC memory; the
            # padded form carries cp - J zero channels per texel
            pad = cp - self.J if args.input_layout == 'nhwc' else 0
            self.fms = [torch.nn.functional.pad(f, (0, 0, 0, 0, 0, pad)).contiguous(
                memory_format=torch.channels_last) for f in self.fms]
        # frames go through in chunks that reuse one channel-last buffer (ops.root_step);
        # the grouped launch works on the whole batch
        # (maps read in place: nothing to keep cache-resident between two kernels, one launch
        # over the whole batch is fastest -- 0.258 ms against 0.282 in 8-frame launches)
        auto = self.B if self.nhwc else ops.chunk_frames(self.V, h, w, cp)
        self.chunk = self.B if self.groups is not None else min(args.chunk or auto, self.B)
        self.hm = None if self.nhwc else torch.empty(self.chunk, self.V, h, w, cp, device=dev)
        self.layouts = {c0: ops.channels_last_layout([f[c0:c0 + n] for f in self.fms])
                        for c0, n in self._chunks()} if self.nhwc else None
        self.zmax_mode = getattr(args, 'zmax', 'off')
        if self.zmax_mode != 'off' and (self.groups is not None or self.nhwc):
            raise SystemExit('--zmax times the repacked root step')
        self.out = (torch.empty(self.B, self.J, X, Y, Z, device=dev)
                    if self.zmax_mode != 'only' else None)
        self.zmax = (torch.empty(self.B, self.J, X, Y, device=dev)
                     if self.zmax_mode != 'off' else None)
        self.K = args.columns if self.zmax_mode == 'only' else 0
        if self.K:
            g = torch.Generator().manual_seed(seed + 5)
            self.col_index = torch.randint(0, X * Y, (self.B, self.K), generator=g).to(dev)
            self.cols = torch.empty(self.B, self.K, self.J, Z, device=dev)
        self.samples = self.B * self.N * self.V
        # outputs written once + every heatmap element read once + cameras
        # (--zmax only: the outputs are the z-max planes and the K z-columns, no volume)
        out_elems = ((self.B * self.J * self.N if self.zmax_mode != 'only' else 0) +
                     (self.B * self.J * X * Y if self.zmax_mode != 'off' else 0) +
                     self.B * self.K * self.J * Z)
        self.kernel_bytes = (4 * out_elems + 4 * self.B * self.V * self.J * h * w
                             + 128 * self.B * self.V)
        if self.zmax_mode == 'only':
            self.name += '_zmax_columns_no_volume'
        elif self.zmax_mode == 'with':
            self.name += '_with_zmax'
        self.kernel_name = ('unproject_group_kernel<ROOT>' if self.groups is not None
                            else 'unproject_kernel<ROOT>')

    def _chunks(self):
        return [(c0, min(self.chunk, self.B - c0)) for c0 in range(0, self.B, self.chunk)]

    def repack(self, c0=0, n=None):
        if self.nhwc:
            return
        n = self.chunk if n is None else n
        self.ops.repack_nhwc([f[c0:c0 + n] for f in self.fms], out=self.hm[:n])

    def kernel(self, c0=0, n=None):
        n = self.chunk if n is None else n
        cfg = self.cfg
        if self.nhwc:
            self.ops.unproject_root_views(self.layouts[c0], self.cam[c0:c0 + n], self.axes, self.J,
                                          cfg['heatmap_size'], cfg['image_size'],
                                          out=self.out[c0:c0 + n])
        else:
            self.ops.unproject_root(self.hm[:n], self.cam[c0:c0 + n], self.axes, self.J,
                                    cfg['heatmap_size'], cfg['image_size'],
                                    out=None if self.out is None else self.out[c0:c0 + n],
                                    groups=self.groups,
                                    zmax_out=None if self.zmax is None else self.zmax[c0:c0 + n],
                                    want_cubes=self.out is not None)

    def columns(self):
        """the K z-columns per frame straight from the NCHW maps (tempo_root_step_columns)"""
        cfg = self.cfg
        self.ops.unproject_root_columns(self.fms, self.cam, self.axes, self.col_index,
                                        cfg['heatmap_size'], cfg['image_size'], out=self.cols)

    def step(self):
        cfg = self.cfg
        if self.groups is not None:
            self.repack(0, self.B)
            self.kernel(0, self.B)
            return
        if self.nhwc:  # maps read in place: one launch per chunk of frames, no repack
            for c0, n in self._chunks():
                self.kernel(c0, n)
            return
        # one library call: tempo_root_step runs the chunk loop (repack + unprojection per
        # chunk, one reused channel-last workspace)
        self.ops.root_step(self.fms, self.cam, self.axes, cfg['heatmap_size'],
                           cfg['image_size'], out=self.out, chunk=self.chunk,
                           hm_buf=self.hm.view(-1), zmax_out=self.zmax,
                           want_cubes=self.out is not None)
        if self.K:
            self.columns()

    def parts(self, steps):
        """(kernel ms, repack ms) per step and kernel launches per step: HIP events around
        every launch, on the launch stream, enqueued behind a blocker so that no interval
        contains host launch latency."""
        steps = min(max(steps, 10), 30)
        E = lambda: torch.cuda.Event(enable_timing=True)
        ks, rs = [], []
        for c0, n in self._chunks():  # (first use of these entry points: not timed)
            self.repack(c0, n)
            self.kernel(c0, n)
        torch.cuda.synchronize()
        keep = gpu_blocker(self.dev)
        for _ in range(steps):
            for c0, n in self._chunks():
                r0, r1, k0, k1 = E(), E(), E(), E()
                r0.record(); self.repack(c0, n); r1.record()
                k0.record(); self.kernel(c0, n); k1.record()
                rs.append((r0, r1)); ks.append((k0, k1))
        torch.cuda.synchronize()
        del keep
        # per step: the sum over its launches; the median step (a clock dip of a few ms --
        # seen on these boxes -- then costs one sample, not the mean)
        per = len(self._chunks())
        ksum = sorted(sum(a.elapsed_time(b) for a, b in ks[i * per:(i + 1) * per])
                      for i in range(steps))
        rsum = sorted(sum(a.elapsed_time(b) for a, b in rs[i * per:(i + 1) * per])
                      for i in range(steps))
        return ksum[steps // 2], rsum[steps // 2], per


class WindowWorkload:
    """BASELINE configs[2]: 64^3 windows, 10 proposals x 32 frames, Panoptic cam5.
    With --timesteps T the frames are B/T sequences of T frames and every proposal
    is unprojected at each of its sequence's T frames (tempo_pose.py:145-156):
    those T windows share geometry and calibration, hence one projection."""
    name = 'panoptic_cam5_window_64cubed_10prop_b32'
    cfg_src = 'PANOPTIC'
    default_T = 1

    def __init__(self, args, dev, seed):
        from tempo_amd import ops, project_layer as PL, synthetic as S
        self.ops = ops
        self.dev = dev
        self.cfg, self.fms, self.cam, self.cam_host = make_inputs(args, dev, seed,
                                                                  getattr(S, self.cfg_src))
        self.meta = make_inputs.meta
        cfg = self.cfg
        self.B, self.V, self.J = args.batch, cfg['num_views'], cfg['num_joints']
        P = args.proposals
        T = self.T = args.timesteps or self.default_T
        if self.B % T:
            raise SystemExit('--batch must be a multiple of --timesteps')
        self.NW = self.B * P
        layer = PL.ProjectLayerWithMask(cfg['image_size'], cfg['heatmap_size'], 'mean')
        layer.set_space_size(cfg['space_center'], cfg['space_size'], cfg['sub_space_size'],
                             cfg['sub_cube_size'], dev)
        # one proposal list per sequence; windows ordered (sequence, proposal, timestep)
        props = S.make_proposals(self.NW // T, cfg['space_center'], cfg['space_size'], seed=seed,
                                 invalid_every=0, margin=0.15).to(dev)
        props[:, 5:7] = 1.0  # full windows: every voxel is sampled
        props = props.repeat_interleave(T, 0)
        self.props = props
        tl, _, st, en = layer.window_geometry(props)
        layer._ensure_grids(dev)
        self.geo = [t.to(torch.int32).contiguous() for t in (tl, st, en)]
        self.fine = layer._fine_axes
        self.fine_host = layer._fine_axes_host
        self.sub = layer._sub_cube
        i = torch.arange(self.NW, device=dev, dtype=torch.int32)
        self.win_frame = ((i // (P * T)) * T + i % T).to(torch.int32).contiguous()
        self.groups = None
        if T > 1 and args.share:
            self.groups = torch.tensor([[k * T, T] for k in range(self.NW // T)],
                                       dtype=torch.int32, device=dev)
        w, h = cfg['heatmap_size']
        cp = ops.padded_channels(self.J)
        self.input_layout = args.input_layout
        self.nhwc = args.input_layout != 'nchw'
        if self.nhwc:
            if self.groups is not None:
                raise SystemExit('--input-layout nhwc: not with --share')
            pad = cp - self.J if args.input_layout == 'nhwc' else 0
            self.fms = [torch.nn.functional.pad(f, (0, 0, 0, 0, 0, pad)).contiguous(
                memory_format=torch.channels_last) for f in self.fms]
        # regular batches (window j looks at frame j // P) go through in frame chunks that
        # reuse one channel-last buffer (ops.window_step); sequences / grouped launches do not
        self.P = P
        self.regular = self.groups is None and (T == 1 or P == 1)
        self.chunk = self.B
        whole = self.nhwc or args.planes != 'off'  # (planes: one launch unless asked otherwise)
        if (T == 1 and self.groups is None) or (self.regular and not self.nhwc):
            self.chunk = min(self.B, args.window_chunk or
                             (self.B if whole else ops.chunk_frames(self.V, h, w, cp)))
        self.hm = None if self.nhwc else torch.empty(self.chunk, self.V, h, w, cp, device=dev)
        # in-place layouts of every chunk's maps, built once (views of the callers' tensors)
        self.layouts = {f0: ops.channels_last_layout([f[f0:f0 + n] for f in self.fms])
                        for f0, n in self._chunks()} if self.nhwc else None
        self.N = self.sub[0] * self.sub[1] * self.sub[2]
        self.planes_mode = args.planes
        self.out = (torch.empty(self.NW, self.J, *self.sub, device=dev)
                    if args.planes != 'only' else None)
        sx, sy, sz = self.sub
        self.planes = [torch.empty(self.NW, self.J, *sh, device=dev)
                       for sh in ((sx, sy), (sx, sz), (sy, sz))] if args.planes != 'off' else None
        self.samples = self.NW * self.N * self.V
        # algorithmic bytes: outputs written once + every heatmap element read once + cameras
        out_elems = (self.NW * self.J * self.N if args.planes != 'only' else 0) + \
                    (self.NW * self.J * (sx * sy + sx * sz + sy * sz) if args.planes != 'off' else 0)
        self.kernel_bytes = (4 * out_elems + 4 * self.B * self.V * self.J * h * w
                             + 128 * self.B * self.V)
        self.kernel_name = ('unproject_group_kernel<WINDOW>' if self.groups is not None
                            else 'unproject_planes_kernel' if args.planes != 'off'
                            else 'unproject_kernel<WINDOW>')

    def _chunks(self):
        return [(f0, min(self.chunk, self.B - f0)) for f0 in range(0, self.B, self.chunk)]

    def repack(self, f0=0, n=None):
        if self.nhwc:
            return
        n = self.chunk if n is None else n
        self.ops.repack_nhwc([f[f0:f0 + n] for f in self.fms], out=self.hm[:n])

    def _chunk_args(self, f0, n):
        key = (f0, n)
        if key not in self._memo:
            whole = n == self.B
            ws = slice(0, self.NW) if whole else slice(f0 * self.P, (f0 + n) * self.P)
            # inside a chunk the frames are numbered from 0
            wf = self.win_frame if whole else (self.win_frame[ws] - f0).contiguous()
            self._memo[key] = (ws, [t[ws] for t in self.geo], wf,
                               self.cam if whole else self.cam[f0:f0 + n])
        return self._memo[key]

    _memo = None

    def kernel(self, f0=0, n=None):
        n = self.chunk if n is None else n
        if self._memo is None:
            self._memo = {}
        ws, (tl, st, en), wf, cam = self._chunk_args(f0, n)
        if self.nhwc and self.planes_mode != 'off':  # planes from maps read in place
            self.ops.unproject_window_planes_views(
                self.layouts[f0], cam, self.fine, tl, st, en, self.sub, self.J,
                self.cfg['heatmap_size'], self.cfg['image_size'], win_frame=wf,
                want_cubes=self.planes_mode == 'with',
                out=None if self.out is None else self.out[ws],
                planes_out=[p[ws] for p in self.planes])
            return
        if self.nhwc:  # maps read in place
            self.ops.unproject_window_views(
                self.layouts[f0], cam, self.fine, tl, st, en, self.sub, self.J,
                self.cfg['heatmap_size'], self.cfg['image_size'], win_frame=wf, out=self.out[ws])
            return
        if self.planes_mode != 'off':
            self.ops.unproject_window_planes(
                self.hm[:n], cam, self.fine, tl, st, en, self.sub, self.J,
                self.cfg['heatmap_size'], self.cfg['image_size'], win_frame=wf,
                want_cubes=self.planes_mode == 'with',
                out=None if self.out is None else self.out[ws],
                planes_out=[p[ws] for p in self.planes])
            return
        self.ops.unproject_window(self.hm[:n], cam, self.fine, tl, st, en, self.sub, self.J,
                                  self.cfg['heatmap_size'], self.cfg['image_size'],
                                  win_frame=wf, out=self.out[ws], groups=self.groups)

    def step(self):
        if self.regular and not self.nhwc and self.planes_mode == 'off':
            # the library's own chunk loop (repack + unprojection per chunk of frames)
            tl, st, en = self.geo
            self.ops.window_step(self.fms, self.cam, self.fine, tl, st, en, self.sub,
                                 self.cfg['heatmap_size'], self.cfg['image_size'],
                                 windows_per_frame=self.P, out=self.out, chunk=self.chunk,
                                 hm_buf=self.hm)
            return
        for f0, n in self._chunks():
            self.repack(f0, n)
            self.kernel(f0, n)

    parts = RootWorkload.parts


class WindowH36MWorkload(WindowWorkload):
    """BASELINE configs[3], one GPU's slice: H36M 4 cameras, 17 joints, 384x384 maps,
    64^3 window, 1 person, T = 4 frames per sequence (run with --batch 128 for the
    128 windows per GPU of the 8-GPU split)."""
    name = 'h36m_cam4_window_64cubed_T4'
    cfg_src = 'H36M'
    default_T = 4


class DenseWorkload:
    """BASELINE configs[4]: one frame, 10 views, 15 joints, one 128^3 grid.  With N ranks the
    grid is z-slab-sharded (rank r computes planes [r*128/N, (r+1)*128/N) from replicated
    heatmaps) and ONE all-gather over RCCL/xGMI plus a device permute reassembles it: the total
    work is fixed, so this workload reports strong scaling."""
    name = 'dense_128cubed_10views_zslab'
    scaling = 'strong'

    def __init__(self, args, dev, seed):
        import copy
        from tempo_amd import ops, parallel, project_layer as PL, synthetic as S
        self.ops, self.parallel = ops, parallel
        cfg = copy.deepcopy(S.PANOPTIC)
        cfg['num_views'], cfg['cube_size'] = 10, [128, 128, 128]
        args = copy.copy(args)
        args.batch, args.calibrations = 1, 1
        # every rank holds the same frame (replicated heatmaps): same seed on all ranks
        self.cfg, self.fms, self.cam, _ = make_inputs(args, dev, 0, cfg)
        self.meta = make_inputs.meta
        self.B, self.V, self.J = 1, 10, cfg['num_joints']
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        rank = int(os.environ.get('RANK', '0'))
        X, Y, Z = cfg['cube_size']
        z0, z1 = parallel.zslab_range(Z, rank, self.world)
        self.axes = [(PL._base_axis(cfg['space_size'][i], cfg['cube_size'][i]) +
                      cfg['space_center'][i]).to(dev) for i in range(3)]
        self.axes[2] = self.axes[2][z0:z1].contiguous()
        self.N = X * Y * (z1 - z0)
        w, h = cfg['heatmap_size']
        self.hm = torch.empty(1, self.V, h, w, ops.padded_channels(self.J), device=dev)
        self.out = torch.empty(1, self.J, X, Y, z1 - z0, device=dev)
        self.full = None
        self.samples = self.N * self.V  # per rank; the bench multiplies by the world size
        self.kernel_bytes = 4 * self.J * self.N + 4 * self.V * self.J * h * w + 128 * self.V
        self.kernel_name = 'unproject_kernel<ROOT>'
        self.groups = None

    def repack(self):
        self.ops.repack_nhwc(self.fms, out=self.hm)

    def kernel(self):
        self.ops.unproject_root(self.hm, self.cam, self.axes, self.J, self.cfg['heatmap_size'],
                                self.cfg['image_size'], out=self.out)

    def step(self):
        self.repack()
        self.kernel()
        if self.world > 1:  # (one rank holds the whole grid; its collective is timed apart)
            self.full = self.parallel.gather_z_slabs(self.out)


def make_workload(args, dev, seed):
    if args.workload == 'window_h36m' and args.proposals == 10:
        args.proposals = 1  # one person per sequence
    return {'root': RootWorkload, 'window': WindowWorkload, 'window_h36m': WindowH36MWorkload,
            'dense': DenseWorkload}[args.workload](args, dev, seed)


def cpu_baseline(args):
    """The oracle (reference-structured torch CPU path) on BASELINE.json configs[0]:
    Panoptic cam5, 1 frame, root grid 80x80x20 -> 640 000 samples per call."""
    from oracle import project_layer_ref as O
    from tempo_amd import synthetic as S
    cfg = S.PANOPTIC
    V, J = cfg['num_views'], cfg['num_joints']
    w, h = cfg['heatmap_size']
    fms = S.make_heatmaps(1, V, J, h, w, kind='noise', seed=0)
    cams = S.ring_cameras(V, seed=0, target=cfg['space_center'])
    centers, scales, cameras = S.frame_meta(cams, 1, cfg['raw_size'], cfg['image_size'])
    layer = O.ProjectLayer(cfg['image_size'], cfg['heatmap_size'], 'mean')
    cores = min(os.cpu_count() or 1, 16)  # the GPU box grants a 16-core share
    call = lambda: layer(fms, cfg['space_size'], [cfg['space_center']], cfg['cube_size'],
                         centers, scales, cameras)

    def median_call(threads, budget_s, max_calls):
        torch.set_num_threads(threads)
        for _ in range(3):
            call()
        times = []
        t_end = time.perf_counter() + budget_s
        while len(times) < 10 or (time.perf_counter() < t_end and len(times) < max_calls):
            t0 = time.perf_counter()
            call()
            times.append(time.perf_counter() - t0)
        times.sort()
        return times[len(times) // 2], len(times)

    n = cfg['cube_size'][0] * cfg['cube_size'][1] * cfg['cube_size'][2]
    med, calls = median_call(cores, 12.0, 200)
    med1, calls1 = median_call(1, 8.0, 50)  # also at one thread
    torch.set_num_threads(cores)
    # the same reference-structured torch code with its tensors on the GPU (python loop over
    # views, stock ATen kernels, F.grid_sample): what the reference's own path costs on this
    # MI355X, forward and forward+backward -- a baseline beside the CPU one, not the product
    gpu = {}
    if torch.cuda.is_available():
        dev = torch.device('cuda', torch.cuda.current_device())
        gfms = [f.to(dev).requires_grad_(True) for f in fms]

        def gcall(backward):
            with torch.set_grad_enabled(backward):
                cubes, _ = layer(gfms, cfg['space_size'], [cfg['space_center']],
                                 cfg['cube_size'], centers, scales, cameras)
                if backward:
                    cubes.sum().backward()
            torch.cuda.synchronize()
            return cubes

        for backward in (False, True):
            for _ in range(3):
                gcall(backward)
            ts = []
            for _ in range(10):
                t0 = time.perf_counter()
                gcall(backward)
                ts.append(time.perf_counter() - t0)
            ts.sort()
            gpu['fwd_bwd_ms' if backward else 'fwd_ms'] = round(ts[len(ts) // 2] * 1e3, 3)
        gpu['fwd_gsamp_s'] = round(n * V / (gpu['fwd_ms'] * 1e-3) / 1e9, 4)
        gpu['what'] = ('oracle ProjectLayer (reference-structured torch ops) with tensors on '
                       'this GPU, same 640000-sample call, median of 10, host-synchronised')
        cpu_baseline.on_gpu_check = oracle_on_gpu_check(layer, cfg, dev, centers, scales, cameras)
    return {
        'torch_ops_on_gpu': gpu,
        'value': round(n * V / med / 1e9, 6), 'unit': 'Gsamp/s', 'cores': cores,
        'kind': 'port',
        'sample': f'oracle ProjectLayer, Panoptic cam5 root 80x80x20, B=1 (640000 samples/call), '
                  f'median of {calls} calls, {med * 1e3:.1f} ms/call',
        'value_1_thread': round(n * V / med1 / 1e9, 6),
        'sample_1_thread': f'same call, torch.set_num_threads(1), median of {calls1} calls, '
                           f'{med1 * 1e3:.1f} ms/call',
    }


def oracle_on_gpu_check(ref_layer, cfg, dev, centers, scales, cameras):
    """How far the HIP path is from the reference's torch ops EXECUTED ON THIS GPU (report
    only: the bar of the parity tests is torch on the CPU; ATen's GPU kernels round the K=3
    matmuls, grid_sample and the view sum their own way).  Same frame through the oracle with
    device tensors and through the drop-in module; blob and noise maps."""
    import tempo_amd
    from tempo_amd import synthetic as S
    V, J = cfg['num_views'], cfg['num_joints']
    w, h = cfg['heatmap_size']
    mine = tempo_amd.ProjectLayer(cfg['image_size'], cfg['heatmap_size'], 'mean')
    mine.keep_mask = True
    mine.materialize_grids = False
    res = {'what': 'HIP module vs oracle ProjectLayer run with tensors on this GPU (ATen GPU '
                   'kernels), config 1 frame; report only -- the parity bar is torch on the CPU'}
    for kind in ('blobs', 'noise'):
        fms = [f.to(dev) for f in S.make_heatmaps(1, V, J, h, w, kind=kind, seed=0)]
        with torch.no_grad():
            ref, _ = ref_layer(fms, cfg['space_size'], [cfg['space_center']], cfg['cube_size'],
                               centers, scales, cameras)
            got, _ = mine(fms, cfg['space_size'], [cfg['space_center']], cfg['cube_size'],
                          centers, scales, cameras)
        bits = (ref_layer.last_bounding.to(torch.int32) <<
                torch.arange(V, dtype=torch.int32, device=ref_layer.last_bounding.device)).sum(-1)
        mask_diff = int((bits.view(-1).to(dev) != mine.last_mask.view(-1).to(torch.int32)).sum())
        d = (torch.nan_to_num(got) - torch.nan_to_num(ref)).abs()
        res[kind] = {'max_abs': float(d.max()),
                     'differing_elements': int((d != 0).sum()), 'elements': d.numel(),
                     'nan_positions_equal': bool(torch.equal(got.isnan(), ref.isnan())),
                     'differing_mask_words': mask_diff, 'mask_words': bits.numel()}
    return res


def measured_traffic(args, wl):
    """HBM bytes per launch of the dominant kernel from the committed PMC pass of this very
    workload (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs: tools/profile_round.sh).
    bench.py cannot collect counters about itself, so the number comes from profiles/ and is
    reported only for the configuration that was profiled.  FETCH_SIZE is in KB and counts
    64 B per fabric request; this kernel's texel fetches are 64-byte requests, so it is taken at
    face value (calibration: profiles/r01/fetch_size_calibration.txt)."""
    import csv
    if (args.workload in ('window_h36m', 'dense') or (args.timesteps or 1) != 1
            or args.calibrations != 1
            or args.planes == 'with' or args.share):
        return None, None
    tag = 'root' if args.workload == 'root' else ('planes' if args.planes == 'only' else 'window')
    zmode = getattr(args, 'zmax', 'off')
    if zmode == 'with' or (zmode == 'only' and args.columns != 10):
        return None, None
    if zmode == 'only':
        tag = 'root_zmax_only'
    if args.channels == 32:
        tag += '_c32'
    layout = getattr(wl, 'input_layout', 'nchw')
    if layout == 'nhwc':
        tag += '_nhwc'
    here = os.path.dirname(os.path.abspath(__file__))
    path = src = None
    for rnd in ('r03', 'r02', 'r01'):  # the newest committed profile of this configuration
        cand = os.path.join(here, 'profiles', rnd, tag + '_pmc_summary.csv')
        if os.path.exists(cand):
            path, src = cand, f'profiles/{rnd}/{tag}_pmc_summary.csv'
            break
    if (path is None or layout == 'nhwc-tight' or args.batch != 32 or args.chunk is not None
            or (not tag.startswith('root') and args.proposals != 10)
            or (args.channels and args.channels != 32)):
        return None, None
    vals = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if row['kernel'].startswith(wl.kernel_name.split('<')[0] + '<'):
                vals[row['counter']] = float(row['mean_per_dispatch'])
    if 'FETCH_SIZE' not in vals or 'WRITE_SIZE' not in vals:
        return None, None
    if 'TCC_HIT_sum' in vals and 'TCC_MISS_sum' in vals:  # L2 hit rate of the same passes
        measured_traffic.l2_hit = round(
            vals['TCC_HIT_sum'] / (vals['TCC_HIT_sum'] + vals['TCC_MISS_sum']), 4)
    if 'TCP_TOTAL_CACHE_ACCESSES_sum' in vals and 'TCP_TCC_READ_REQ_sum' in vals:
        # What the texture path allows for this launch's own access mix: its 64-byte L1
        # accesses priced at the rates the gather microbenchmark sustains on this chip
        # (profiles/r01/gather_rate_microbenchmark.txt: 0.87 L1 hits, 0.41 L1 misses served by
        # L2 per clock and CU), 256 CUs at 2.4 GHz, the path busy for the whole launch.
        miss = vals['TCP_TCC_READ_REQ_sum']
        hit = max(0.0, vals['TCP_TOTAL_CACHE_ACCESSES_sum'] - miss)
        measured_traffic.texture_path_ms = (hit / 0.87 + miss / 0.41) / 256 / 2.4e9 * 1e3
    return int((vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024), src


def attainable(wl, launches):
    ms = getattr(measured_traffic, 'texture_path_ms', None)
    if ms is None:
        return None
    gbs = wl.kernel_bytes / launches / (ms * 1e-3) / 1e9
    return {'kernel_launch_ms': round(ms, 4), 'achieved': round(gbs, 1),
            'frac': round(gbs / HBM_PEAK_GBS, 4),
            'bound': 'texture path (TA/TCP): L1 accesses of the profiled launch at the '
                     'microbenchmarked hit and miss rates'}


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def launch_ranks(args):
    """--gpus N > 1 outside a torch.distributed launch: start the N ranks as a child
    process group (one process per GPU) and exit with its code.  The parent has not
    touched the GPU and replaces nothing: it waits for the child."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    return subprocess.run(cmd, env=env).returncode


class DryWorkload:
    """--dry-run: no device work at all; what is left is the launcher, the rendezvous
    and the block timing protocol (CPU boxes, tests/test_bench_launcher.py)."""
    name = 'dry_run'
    B, V, J, samples, kernel_bytes, kernel_name = 0, 0, 0, 0, 0, 'none'

    def step(self):
        time.sleep(0.001)


def oracle_check(args, wl, dev):
    """max-abs error of one item of the timed output against the CPU oracle run on the
    same inputs (item 0: a whole frame of the root grid / the first window)."""
    from oracle import project_layer_ref as O
    cfg = wl.cfg
    centers, scales, cameras = wl.meta
    fms = [f[0:1, :wl.J].contiguous().cpu() for f in wl.fms]  # (nhwc: drop the pad channels)
    one = ([centers[0]], [scales[0]], [cameras[0]])
    if args.workload in ('root', 'dense'):
        if args.workload == 'dense' and getattr(wl, 'world', 1) > 1:
            return None
        layer = O.ProjectLayer(cfg['image_size'], cfg['heatmap_size'], 'mean')
        ref, _ = layer(fms, cfg['space_size'], [cfg['space_center']], cfg['cube_size'], *one)
        if wl.out is None:  # --zmax only: the z-max planes and the K z-columns of frame 0
            C, Z = ref.shape[1], ref.shape[4]
            idx = wl.col_index[0:1].cpu()
            ref_cols = torch.gather(torch.flatten(ref, 2, 3).permute(0, 2, 1, 3), dim=1,
                                    index=idx.view(1, -1, 1, 1).repeat(1, 1, C, Z))
            ref = torch.cat([ref.max(dim=4)[0].flatten(), ref_cols.flatten()])
            got = torch.cat([wl.zmax[0:1].cpu().flatten(), wl.cols[0:1].cpu().flatten()])
            what = ('z-max planes and z-columns of frame 0 vs torch.max / the torch.gather of '
                    'tempo_detector.py:204-206 on the oracle volume')
        else:
            got = wl.out[0:1].cpu()
            what = 'frame 0 of the timed output vs oracle ProjectLayer on the same inputs'
    else:
        if wl.out is None:
            return None
        layer = O.ProjectLayerWithMask(cfg['image_size'], cfg['heatmap_size'], 'mean')
        ref, _ = layer(fms, wl.props[0:1].cpu(), cfg['space_center'], cfg['space_size'],
                       cfg['sub_space_size'], cfg['sub_cube_size'], *one)
        got = wl.out[0:1].cpu()
        what = 'window 0 of the timed output vs oracle ProjectLayerWithMask on the same inputs'
    same_nan = bool(torch.equal(got.isnan(), ref.isnan()))
    err = float((torch.nan_to_num(got) - torch.nan_to_num(ref)).abs().max())
    return {'max_abs_err': err, 'nan_positions_equal': same_nan,
            'differing_elements': int((torch.nan_to_num(got) != torch.nan_to_num(ref)).sum()),
            'elements': ref.numel(), 'what': what}


def module_timing(args, wl, dev):
    """The drop-in nn.Module called the way tempo_detector.py:187-190 calls it: lists of V
    [B,C,H,W] tensors and lists of per-(frame, view) dicts, B = 32.  First call (cold
    caches: camera packing, axes) and steady state (median of 20), host-synchronised wall
    time -- what a caller of the module sees, host packing included."""
    import tempo_amd
    cfg = wl.cfg
    centers, scales, cameras = wl.meta
    layer = tempo_amd.ProjectLayer(cfg['image_size'], cfg['heatmap_size'], 'mean')
    layer.materialize_grids = False  # the caller drops `grids` (tempo_detector.py:187)
    call = lambda: layer(wl.fms, cfg['space_size'], [cfg['space_center']], cfg['cube_size'],
                         centers, scales, cameras)
    times = []
    with torch.no_grad():
        for _ in range(23):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            cubes, _ = call()
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
    same = bool(torch.equal(torch.nan_to_num(cubes, nan=-1.0),
                            torch.nan_to_num(wl.out, nan=-1.0)))
    steady = sorted(times[3:])
    # and as a caller's loop issues it: calls back to back, one synchronisation at the end
    with torch.no_grad():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            call()
        host = (time.perf_counter() - t0) / 50 * 1e3
        torch.cuda.synchronize()
        piped = (time.perf_counter() - t0) / 50 * 1e3
    return {'first_call_ms': round(times[0], 3), 'steady_ms': round(steady[len(steady) // 2], 4),
            'pipelined_ms': round(piped, 4), 'host_ms': round(host, 4),
            'equals_timed_output': same,
            'what': 'tempo_amd.ProjectLayer.forward(list of [B,C,H,W], ..., lists of dicts), '
                    'host-synchronised wall time incl. camera packing; grids not materialised; '
                    'pipelined_ms = 50 calls back to back, one synchronisation (host_ms of it '
                    'is python: argument checks, the content-keyed camera-table cache)'}


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args))  # before anything touches the GPU
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    if args.dry_run:
        dev = torch.device('cpu')
    else:
        if not torch.cuda.is_available():
            raise SystemExit('bench.py needs a HIP device (tempo_amd has no CPU path)')
        torch.cuda.set_device(local)
        dev = torch.device('cuda', local)
    # the dense workload's data path has a collective: at one rank it still runs (a one-rank
    # RCCL group), so that the all-gather and the slab assembly are timed on hardware
    if world > 1 or (args.workload == 'dense' and not args.dry_run):
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if world == 1:
            os.environ.setdefault('MASTER_PORT', str(free_port()))
        if args.dry_run:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    def sync():
        if not args.dry_run:
            torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            if args.dry_run:
                dist.barrier()
            else:
                dist.barrier(device_ids=[local])
        sync()

    # ranks the collective library actually connected (an all-reduce of ones)
    seen = 1
    if dist is not None:
        one = torch.ones(1, device=dev, dtype=torch.float32)
        dist.all_reduce(one)
        seen = int(round(float(one.item())))

    if args.input_layout != 'nchw' and (args.backward or args.share or
                                        (args.planes != 'off' and args.workload == 'root')):
        raise SystemExit('--input-layout nhwc times the forward kernels only')
    wl = DryWorkload() if args.dry_run else make_workload(args, dev, seed=rank)

    for _ in range(args.warmup):
        wl.step()

    def timed_block():
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            wl.step()
        barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    # Every block is exactly --steps steps between two barrier + synchronize pairs.  One block
    # of the default workload lasts a few ms, too short for the clocks to settle or for an
    # outside observer (the driver samples GPU activity) to see the GPU working: the number
    # of blocks is raised until the timed region lasts --min-seconds (every rank computes the
    # same count from the max-over-ranks times of two probe blocks; the first block of a
    # process can run several times slower than the steady state -- clocks ramping up -- so the
    # faster of the two sizes the region).
    block_s = [timed_block()]
    blocks = max(1, args.blocks)
    if not args.dry_run and args.min_seconds > 0:
        block_s.append(timed_block())
        est = min(block_s)
        if est > 0:
            blocks = max(blocks, min(20000, int(args.min_seconds / est) + 1))
    for _ in range(blocks - len(block_s)):
        block_s.append(timed_block())
    elapsed = sorted(block_s)[len(block_s) // 2]

    if args.dry_run:
        if rank == 0:
            print(json.dumps({'metric': 'voxel-view heatmap samples/sec', 'value': 0.0,
                              'unit': 'Gsamp/s', 'n_gpus': seen, 'steps': args.steps,
                              'warmup': args.warmup,
                              'ms_per_step': round(elapsed / args.steps * 1e3, 4),
                              'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
                              'dtype': 'f32', 'data': 'none', 'dry_run': True,
                              'config': {'workload': 'dry_run', 'ranks_launched': world}}),
                  flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return

    # dominant kernel alone, HIP events on the launch stream
    if hasattr(wl, 'parts'):
        k_ms, r_ms, launches = wl.parts(args.steps)
    else:
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(args.steps)]
        rp = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(args.steps)]
        for (a, b), (c, d) in zip(ev, rp):
            c.record()
            wl.repack()
            d.record()
            a.record()
            wl.kernel()
            b.record()
        torch.cuda.synchronize()
        k_ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
        r_ms = sum(c.elapsed_time(d) for c, d in rp) / len(rp)
        launches = 1
    comm_ms = comm = None
    if args.workload == 'dense':  # the all-gather and the slab assembly alone, separately
        def ev_ms(fn, n=max(10, args.steps)):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            barrier()
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n

        slabs = wl.parallel.allgather_slabs(wl.out)
        full = torch.empty(wl.out.shape[:4] + (wl.out.shape[4] * world,), device=dev)
        ag_ms = ev_ms(lambda: wl.parallel.allgather_slabs(wl.out, out=slabs))
        as_ms = ev_ms(lambda: wl.parallel.assemble_z_slabs(slabs, out=full))
        comm_ms = ev_ms(lambda: wl.parallel.gather_z_slabs(wl.out))
        comm = {'allgather_ms': round(ag_ms, 4), 'assemble_ms': round(as_ms, 4),
                'allgather_plus_assemble_ms': round(comm_ms, 4), 'ranks': world,
                'bytes_per_rank': wl.out.numel() * 4}
        if world == 1:
            # the 8-rank assembly's traffic on this GPU: the same bytes seen as 8 slabs of
            # Z/8 planes (values meaningless, the copy pattern is the 8-GPU run's), beside
            # the torch permute + reshape it replaces
            B_, C_, X_, Y_, Z_ = wl.out.shape
            fake = wl.out.view(8, B_, C_, X_, Y_, Z_ // 8)
            comm['assemble_8_slabs_ms'] = round(ev_ms(
                lambda: wl.parallel.assemble_z_slabs(fake, out=full)), 4)
            comm['torch_permute_8_slabs_ms'] = round(ev_ms(
                lambda: fake.permute(1, 2, 3, 4, 0, 5).reshape(B_, C_, X_, Y_, Z_)), 4)
        del slabs, full

    cols_ms = None
    if getattr(wl, 'K', 0):  # --zmax only: the z-column launch alone
        for _ in range(3):
            wl.columns()
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(20):
            wl.columns()
        c1.record()
        torch.cuda.synchronize()
        cols_ms = c0.elapsed_time(c1) / 20

    # parity of what was just timed, and the module-level cost of the same call
    check = module = None
    if rank == 0 and world == 1:
        wl.step()
        torch.cuda.synchronize()
        check = oracle_check(args, wl, dev)
        if args.workload == 'root' and not args.no_module and wl.groups is None \
                and not args.channels and not wl.nhwc and wl.zmax_mode == 'off':
            module = module_timing(args, wl, dev)

    # fused epilogue: what the reference's caller does instead (tempo_pose.py:172-176) --
    # the volume kernel followed by torch.max over each axis, on the same inputs
    unfused = None
    if rank == 0 and getattr(wl, 'planes_mode', 'off') != 'off' and not wl.nhwc:
        vol = wl.out if wl.out is not None else torch.empty(wl.NW, wl.J, *wl.sub, device=dev)
        tl, st, en = wl.geo
        wl.hm = torch.empty(wl.B, *wl.hm.shape[1:], device=dev)
        wl.chunk = wl.B
        wl.ops.repack_nhwc(wl.fms, out=wl.hm)

        def unfused_step():
            wl.ops.unproject_window(wl.hm, wl.cam, wl.fine, tl, st, en, wl.sub, wl.J,
                                    wl.cfg['heatmap_size'], wl.cfg['image_size'],
                                    win_frame=wl.win_frame, out=vol)
            return vol.max(dim=4)[0], vol.max(dim=3)[0], vol.max(dim=2)[0]

        for _ in range(3):
            unfused_step()
        u0, u1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        u0.record()
        for _ in range(10):
            ref_planes = unfused_step()
        u1.record()
        torch.cuda.synchronize()
        wl.kernel()
        torch.cuda.synchronize()
        same = all(torch.equal(torch.nan_to_num(p, nan=-1.0), torch.nan_to_num(q, nan=-1.0))
                   for p, q in zip(wl.planes, ref_planes))
        unfused = {'ms': round(u0.elapsed_time(u1) / 10, 4), 'planes_bit_identical': same,
                   'what': 'unproject_window (volume) + torch.max over z, y, x'}
        del vol, ref_planes

    # training step: forward that records the clamp mask, backward into channel-last heatmap
    # gradients, un-repack to per-view NCHW gradients (reported beside the headline)
    train = None
    if rank == 0 and args.backward and args.planes == 'off':
        ops = wl.ops
        cp = wl.hm.shape[-1]
        root = args.workload == 'root'
        if wl.hm.shape[0] != wl.B:  # the training step works on the whole batch at once
            wl.hm = torch.empty(wl.B, *wl.hm.shape[1:], device=dev)
            wl.chunk = wl.B
            ops.repack_nhwc(wl.fms, out=wl.hm)
        items = wl.B if root else wl.NW
        cmask = ops.new_clamp_mask(items, wl.N, cp, dev)
        gout = torch.rand(wl.out.shape, device=dev)
        grads = [torch.zeros_like(f) for f in wl.fms]

        # the window backward's LDS tile, chosen like the drop-in module does (lattice pitch)
        tile = 0 if root else ops.backward_tile_texels(
            wl.cam_host, wl.fine_host, wl.cfg['heatmap_size'], wl.cfg['image_size'])

        def fwd():
            if root:
                ops.unproject_root(wl.hm, wl.cam, wl.axes, wl.J, wl.cfg['heatmap_size'],
                                   wl.cfg['image_size'], out=wl.out, clamp_mask=cmask)
            else:
                tl, st, en = wl.geo
                ops.unproject_window(wl.hm, wl.cam, wl.fine, tl, st, en, wl.sub, wl.J,
                                     wl.cfg['heatmap_size'], wl.cfg['image_size'],
                                     win_frame=wl.win_frame, out=wl.out, clamp_mask=cmask)

        def bwd():
            if root:
                return ops.unproject_root_backward(gout, wl.cam, wl.axes, cmask, wl.hm.shape,
                                                   wl.cfg['heatmap_size'], wl.cfg['image_size'])
            tl, st, en = wl.geo
            return ops.unproject_window_backward(gout, wl.cam, wl.fine, tl, st, en, cmask,
                                                 wl.hm.shape, wl.cfg['heatmap_size'],
                                                 wl.cfg['image_size'], win_frame=wl.win_frame,
                                                 tile_texels=tile)

        def timed(fn, n=10):
            for _ in range(2):
                r = fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                r = fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n, r

        f_ms, _ = timed(fwd)
        b_ms, ghm = timed(bwd)
        u_ms, _ = timed(lambda: ops.unrepack_nchw(ghm, wl.J, out=grads))
        train = {'forward_with_mask_ms': round(f_ms, 4), 'backward_ms': round(b_ms, 4),
                 'unrepack_ms': round(u_ms, 4),
                 'backward_gsamp_s': round(wl.samples / (b_ms * 1e-3) / 1e9, 3),
                 'backward_tile_texels': tile}
        del gout, grads, ghm, cmask

    # training step THROUGH the fused planes (--planes only --backward): planes + claim pass,
    # backward from the planes' gradients; no volume, no gradient volume
    if (rank == 0 and args.backward and args.planes == 'only' and
            args.workload.startswith('window') and not wl.nhwc):
        ops = wl.ops
        if wl.hm.shape[0] != wl.B:
            wl.hm = torch.empty(wl.B, *wl.hm.shape[1:], device=dev)
            wl.chunk = wl.B
        ops.repack_nhwc(wl.fms, out=wl.hm)
        tl, st, en = wl.geo
        hs, im = wl.cfg['heatmap_size'], wl.cfg['image_size']
        tile = args.tile if args.tile is not None else ops.backward_tile_texels(
            wl.cam_host, wl.fine_host, hs, im, planes=True)
        gpl = [torch.rand(p.shape, device=dev) for p in wl.planes]
        grads = [torch.zeros_like(f) for f in wl.fms]
        state = {}

        def pfwd():
            ops.unproject_window_planes(wl.hm, wl.cam, wl.fine, tl, st, en, wl.sub, wl.J, hs, im,
                                        win_frame=wl.win_frame, planes_out=wl.planes)

        def pclaims():
            state['claims'] = ops.window_planes_claims(wl.hm, wl.cam, wl.fine, tl, st, en, wl.sub,
                                                       wl.J, wl.planes, hs, im,
                                                       win_frame=wl.win_frame)

        def pbwd():
            return ops.unproject_window_planes_backward(
                gpl, state['claims'], wl.cam, wl.fine, tl, st, en, wl.hm.shape, hs, im,
                win_frame=wl.win_frame, tile_texels=tile)

        def ptimed(fn, n=10):
            for _ in range(2):
                r = fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                r = fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n, r

        f_ms, _ = ptimed(pfwd)
        c_ms, _ = ptimed(pclaims)
        b_ms, ghm = ptimed(pbwd)
        u_ms, _ = ptimed(lambda: ops.unrepack_nchw(ghm, wl.J, out=grads))
        train = {'through': 'fused planes (no volume, no gradient volume)',
                 'planes_forward_ms': round(f_ms, 4), 'claim_pass_ms': round(c_ms, 4),
                 'backward_ms': round(b_ms, 4), 'unrepack_ms': round(u_ms, 4),
                 'backward_gsamp_s': round(wl.samples / (b_ms * 1e-3) / 1e9, 3),
                 'backward_tile_texels': tile}
        del gpl, grads, ghm, state

    if rank == 0:
        traffic, traffic_src = measured_traffic(args, wl)
        ms_step = elapsed / args.steps * 1e3
        achieved = wl.kernel_bytes / (k_ms * 1e-3) / 1e9
        step_achieved = wl.kernel_bytes / (ms_step * 1e-3) / 1e9
        line = {
            'metric': 'voxel-view heatmap samples/sec',
            'value': round(wl.samples * world / (elapsed / args.steps) / 1e9, 4),
            'unit': 'Gsamp/s',
            'n_gpus': seen, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_step, 4),
            'higher_is_better': True, 'scaling': getattr(wl, 'scaling', 'weak'),
            'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': wl.name, 'frames_per_gpu': wl.B,
                       'calibrations_per_batch': max(1, min(args.calibrations, args.batch)),
                       'shared_projection': getattr(wl, 'groups', None) is not None,
                       'timesteps': getattr(wl, 'T', 1),
                       'planes': getattr(wl, 'planes_mode', 'off'),
                       'views': wl.V,
                       'joints': wl.J, 'samples_per_step_per_gpu': wl.samples,
                       'input_layout': getattr(wl, 'input_layout', 'nchw'),
                       'ranks_launched': world,
                       'partition': (f'z-slabs over {world} rank(s), one all-gather'
                                     if args.workload == 'dense' else
                                     f'frames x proposals over {world} rank(s), no collective')},
            'timing': {'blocks': len(block_s), 'steps_per_block': args.steps,
                       'timed_region_s': round(sum(block_s), 3),
                       'block_ms_per_step': (lambda v: {
                           'first': v[0], 'min': min(v), 'median': sorted(v)[len(v) // 2],
                           'max': max(v), 'last': v[-1]})(
                               [round(b / args.steps * 1e3, 4) for b in block_s]),
                       'reported': 'median block, max over ranks within a block'},
            'roofline': {
                # `bound` names the yardstick (SURVEY 8d prices the path by HBM bytes);
                # `limiter` says what the kernel actually waits on
                'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS,
                'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic,
                'traffic_source': traffic_src,
                # the same yardstick on the whole step (every launch of it: repack, gather,
                # the z-columns), which is what `value` measures
                'step': {'achieved': round(step_achieved, 1),
                         'frac': round(step_achieved / HBM_PEAK_GBS, 4),
                         'ms': round(ms_step, 4)},
                'step_frac': round(step_achieved / HBM_PEAK_GBS, 4),
                'limiter': getattr(wl, 'limiter', None),
                'l2_hit_rate': getattr(measured_traffic, 'l2_hit', None),
                # speed of light of this formulation: the launch's texel accesses at the
                # texture path's measured hit / miss rates; null when
                # the profiled counters of this configuration are not committed
                'attainable': attainable(wl, launches),
                'kernel': wl.kernel_name, 'kernel_ms': round(k_ms, 4),
                'launches_per_step': launches,
                'kernel_launch_ms': round(k_ms / launches, 4),
                'algorithmic_bytes': wl.kernel_bytes,
                'algorithmic_bytes_per_launch': wl.kernel_bytes // launches},
            'repack_ms': round(r_ms, 4),
            'kernel_gsamp_s': round(wl.samples / (k_ms * 1e-3) / 1e9, 3),
        }
        if comm_ms is not None:
            line['allgather_ms'] = round(comm_ms, 4)
            line['collective'] = comm
        if cols_ms is not None:
            line['zcolumns_ms'] = round(cols_ms, 4)
        if check is not None:
            line['oracle_check'] = check
        if module is not None:
            line['module_ms'] = module
        if unfused is not None:
            line['unfused_consumer'] = unfused
        if train is not None:
            line['training_step'] = train
        if world == 1 and not args.no_cpu_baseline:
            # (last: every GPU phase above is over before the host cores are timed)
            line['cpu_baseline'] = cpu_baseline(args)
            if getattr(cpu_baseline, 'on_gpu_check', None) is not None:
                line['oracle_on_gpu_check'] = cpu_baseline.on_gpu_check
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
