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

# (file suffix, first line, last line) -> phase; first match wins
PHASES = [
    ('unproject_math.h', 82, 91, 'dot3 (X@R, K, affine)'),
    ('unproject_math.h', 101, 117, 'div_const (image_size, hm-1, mean)'),
    ('unproject_math.h', 141, 165, 'div_shared (perspective)'),
    ('unproject_math.h', 211, 225, 'world_to_pixel: camera transform'),
    ('unproject_math.h', 226, 237, 'world_to_pixel: distortion'),
    ('unproject_math.h', 238, 249, 'world_to_pixel: K, NaN patch, bounds'),
    ('unproject_math.h', 253, 276, 'pixel_to_grid: clamp, affine, scale'),
    ('unproject_math.h', 281, 308, 'bilinear tap: unnormalise, floor, weights'),
    ('unproject_math.h', 193, 208, 'clamp helpers'),
    ('unproject_math.h', 63, 80, 'pair helpers'),
    ('tempo_unproject.hip', 119, 121, 'slot -> view'),
    ('tempo_unproject.hip', 139, 148, 'map descriptor'),
    ('tempo_unproject.hip', 157, 165, 'texel offset'),
    ('tempo_unproject.hip', 176, 187, 'camera record (scalar loads)'),
    ('tempo_unproject.hip', 207, 278, 'planes epilogue (not run)'),
    ('tempo_unproject.hip', 281, 313, 'voxel index -> world'),
    ('tempo_unproject.hip', 319, 358, 'wave prologue'),
    ('tempo_unproject.hip', 362, 378, 'phase A: need / count / ballot'),
    ('tempo_unproject.hip', 379, 412, 'phase A: offsets + LDS park'),
    ('tempo_unproject.hip', 417, 429, 'view loop control'),
    ('tempo_unproject.hip', 430, 444, 'fetch: LDS read + buffer loads'),
    ('tempo_unproject.hip', 445, 476, 'mix + view sum'),
    ('tempo_unproject.hip', 477, 497, 'view loop control'),
    ('tempo_unproject.hip', 499, 539, 'normalise (reciprocal, LDS hand-over, division)'),
    ('tempo_unproject.hip', 540, 555, 'clamp mask (training)'),
    ('tempo_unproject.hip', 556, 577, 'clamp + transpose to LDS'),
    ('tempo_unproject.hip', 581, 619, 'block prologue / dead items'),
    ('tempo_unproject.hip', 620, 637, 'volume store'),
]


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
