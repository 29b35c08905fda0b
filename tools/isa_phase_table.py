"""Per-phase instruction table of one kernel, from hipcc's assembly with line info.

    hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -g1 -S --cuda-device-only \
        tempo_amd/csrc/tempo_unproject.hip -o /tmp/unproj.s
    python tools/isa_phase_table.py /tmp/unproj.s unproject_kernelILi0ELi4ELi8ELi1E [--blocks]

Every instruction is attributed to a phase through the innermost source line its `.loc`
names (the inlined arithmetic of unproject_math.h keeps its own lines), and to a basic
block.  Static counts per (block, phase, unit) are printed; `--weights file` multiplies
each block by an execution count per wave (label -> float, one per line) so that the
table can be compared with SQ_INSTS_VALU / SQ_WAVES of a profile.
"""
import collections
import re
import sys

# Phases are found in the sources themselves: (file, regex of the line a phase starts at,
# phase), searched in file order, each after the previous match; a phase lasts until the
# next one starts.
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STARTS = [
    ('unproject_math.h', r'^TEMPO_HD tempo_f2 tempo_splat', 'pair helpers'),
    ('unproject_math.h', r'^// sum_k a_k', 'dot3 (X@R, K, affine)'),
    ('unproject_math.h', r'^// ---- exact IEEE fp32 division', 'div_r64 (image_size, hm-1, mean)'),
    ('unproject_math.h', r'^// \[xc, yc\] / zc', 'div_shared (perspective)'),
    ('unproject_math.h', r'^// per-call constants', 'geometry constants'),
    ('unproject_math.h', r'^TEMPO_HD float tempo_nan_clamp', 'clamp helpers'),
    ('unproject_math.h', r'^// world point -> source-image pixel', 'world_to_pixel: camera transform'),
    ('unproject_math.h', r'^  if \(cam->has_dist', 'world_to_pixel: distortion'),
    ('unproject_math.h', r'^  const float\* K = cam->K;', 'world_to_pixel: K, NaN patch, bounds'),
    ('unproject_math.h', r'^// source-image pixel -> grid_sample', 'pixel_to_grid: clamp, affine, scale'),
    ('unproject_math.h', r'^// grid_sample\(align_corners=False\) unnorm', 'bilinear tap: unnormalise, floor, weights'),
    ('unproject_math.h', r'^// The same footprint as', 'bilinear frac (other kernels)'),
    ('tempo_unproject.hip', r'^__device__ __forceinline__ int slot_view', 'slot -> view'),
    ('tempo_unproject.hip', r'^// host side: slot -> view table', 'host'),
    ('tempo_unproject.hip', r'^// buffer descriptor of one', 'map descriptor'),
    ('tempo_unproject.hip', r'^// wave-uniform read through', 'uniform loads'),
    ('tempo_unproject.hip', r'^__device__ __forceinline__ int texel_index', 'texel offset'),
    ('tempo_unproject.hip', r'^// one byte, wave-uniform', 'uniform loads'),
    ('tempo_unproject.hip', r'^__device__ __forceinline__ TempoCam load_cam', 'camera record (scalar loads)'),
    ('tempo_unproject.hip', r'^// maximum over the wave', 'planes epilogue (not run)'),
    ('tempo_unproject.hip', r"^// Masked mean of one wave's tile", 'normalise: division, class test'),
    ('tempo_unproject.hip', r'^template <int MODE>$', 'voxel index -> world'),
    ('tempo_unproject.hip', r"^// One wave's 64 voxels", 'wave prologue'),
    ('tempo_unproject.hip', r'^  auto phase_a = ', 'phase A: need / count / ballot'),
    ('tempo_unproject.hip', r'^    if \(active\) \{', 'phase A: offsets + LDS park'),
    ('tempo_unproject.hip', r"^  // views in torch's summation order", 'view loop control'),
    ('tempo_unproject.hip', r'^    auto fetch = ', 'fetch: LDS read + buffer loads'),
    ('tempo_unproject.hip', r'^    auto mix = ', 'mix + view sum'),
    ('tempo_unproject.hip', r'^    if \(act_cur\) \{', 'view loop control'),
    ('tempo_unproject.hip', r'^  // -+ phase C', 'normalise: reciprocal, LDS hand-over'),
    ('tempo_unproject.hip', r'^  if \(a.cmask != nullptr\) \{', 'clamp mask (training)'),
    ('tempo_unproject.hip', r'^    if \(lane_on && vsl < kWaveVox\) \{', 'clamp + transpose to LDS'),
    ('tempo_unproject.hip', r"^// One wave's finished tile -> the volume", 'volume store'),
    ('tempo_unproject.hip', r'^// One workgroup of the tile kernel', 'block prologue / dead items'),
    ('tempo_unproject.hip', r'^  unproject_wave<MODE, L, SB, UNF>\(a, b, frame', 'volume store'),
    ('tempo_unproject.hip', r'^// -+ window \+ axis-max planes, fused', 'other kernels'),
]


def phase_ranges():
    out = []  # (file suffix, first line, last line, phase)
    for suf in ('unproject_math.h', 'tempo_unproject.hip'):
        lines = open(os.path.join(REPO, 'tempo_amd', 'csrc', suf)).read().split('\n')
        at = 0
        marks = []
        for fsuf, rx, ph in STARTS:
            if fsuf != suf:
                continue
            r = re.compile(rx)
            for i in range(at, len(lines)):
                if r.match(lines[i]):
                    marks.append((i + 1, ph))
                    at = i + 1
                    break
            else:
                raise SystemExit('phase start not found in %s: %s' % (suf, rx))
        for k, (ln, ph) in enumerate(marks):
            end = marks[k + 1][0] - 1 if k + 1 < len(marks) else len(lines)
            out.append((suf, ln, end, ph))
    return out


PHASES = None


def unit(op):
    if op.startswith('v_'):
        return 'VALU'
    if op.startswith('s_load') or op.startswith('s_buffer_load'):
        return 'SMEM'
    if op.startswith('s_waitcnt') or op.startswith('s_nop') or op.startswith('s_barrier'):
        return 'WAIT'
    if op.startswith('s_cbranch') or op.startswith('s_branch') or op.startswith('s_endpgm'):
        return 'BRANCH'
    if op.startswith('s_'):
        return 'SALU'
    if op.startswith('ds_'):
        return 'LDS'
    if op.startswith('buffer_load') or op.startswith('global_load') or op.startswith('flat_load'):
        return 'VMEM_RD'
    if op.startswith('buffer_store') or op.startswith('global_store') or \
            op.startswith('flat_store') or 'atomic' in op:
        return 'VMEM_WR'
    return 'OTHER'


def phase_of(files, fno, line):
    global PHASES
    if PHASES is None:
        PHASES = phase_ranges()
    name = files.get(fno, '?')
    for suf, a, b, ph in PHASES:
        if name.endswith(suf) and a <= line <= b:
            return ph
    return 'other (%s:%d)' % (name.rsplit('/', 1)[-1], line)


def main():
    path, key = sys.argv[1], sys.argv[2]
    want_blocks = '--blocks' in sys.argv
    weights = None
    if '--weights' in sys.argv:
        weights = {}
        for ln in open(sys.argv[sys.argv.index('--weights') + 1]):
            ln = ln.split('#')[0].split()
            if len(ln) == 2:
                weights[ln[0]] = float(ln[1])
    files = {}
    inside = False
    block = 'entry'
    cur = ('?', 0)
    counts = collections.Counter()  # (block, phase, unit) -> n
    order = ['entry']
    nfall = collections.Counter()
    file_re = re.compile(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?')
    loc_re = re.compile(r'\s*\.loc\s+(\d+)\s+(\d+)')
    with open(path) as f:
        for ln in f:
            m = file_re.match(ln)
            if m:
                files[int(m.group(1))] = (m.group(3) or m.group(2))
                continue
            if not inside:
                if ln.startswith('_Z') and key in ln and ln.rstrip().split(':')[0].endswith('E'):
                    inside = True
                continue
            if ln.startswith('.Lfunc_end'):
                break
            m = loc_re.match(ln)
            if m:
                cur = (int(m.group(1)), int(m.group(2)))
                continue
            if ln.startswith('.LBB'):
                block = ln.split(':')[0]
                order.append(block)
                continue
            s = ln.strip()
            if not s or s.startswith(';') or s.startswith('.'):
                continue
            op = s.split()[0]
            if not re.match(r'^[a-z]', op):
                continue
            counts[(block, phase_of(files, *cur), unit(op))] += 1
            if unit(op) == 'BRANCH' and op != 's_endpgm':
                # a basic block ends at a branch: the fall-through gets a name of its own
                base = block.split('+')[0]
                nfall[base] += 1
                block = '%s+%d' % (base, nfall[base])
                order.append(block)
    units = ['VALU', 'SALU', 'SMEM', 'LDS', 'VMEM_RD', 'VMEM_WR', 'WAIT', 'BRANCH']
    if want_blocks:
        for b in order:
            tot = collections.Counter()
            per = collections.defaultdict(collections.Counter)
            for (bb, ph, u), n in counts.items():
                if bb == b:
                    tot[u] += n
                    per[ph][u] += n
            if not tot:
                continue
            w = '' if weights is None else '  x%g' % weights.get(b, 0.0)
            print('%s%s  ' % (b, w) + ' '.join('%s=%d' % (u, tot[u]) for u in units if tot[u]))
            for ph, c in sorted(per.items(), key=lambda kv: -sum(kv[1].values())):
                print('      %-52s ' % ph + ' '.join('%s=%d' % (u, c[u]) for u in units if c[u]))
    table = collections.defaultdict(collections.Counter)
    for (b, ph, u), n in counts.items():
        w = 1.0 if weights is None else weights.get(b, 0.0)
        table[ph][u] += n * w
    print('\n%-54s' % ('phase (static)' if weights is None else 'phase (per wave)') +
          ''.join('%9s' % u for u in units))
    tot = collections.Counter()
    for ph, c in sorted(table.items(), key=lambda kv: -kv[1]['VALU']):
        if sum(c.values()) == 0:
            continue
        print('%-54s' % ph + ''.join('%9.1f' % c[u] for u in units))
        for u in units:
            tot[u] += c[u]
    print('%-54s' % 'TOTAL' + ''.join('%9.1f' % tot[u] for u in units))


if __name__ == '__main__':
    main()
