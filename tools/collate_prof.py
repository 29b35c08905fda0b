"""Collate rocprofv3 output directories into the small summaries kept under profiles/.

usage: python tools/collate_prof.py <prof_dir> <out_prefix>
  <prof_dir>/kt/**/_kernel_stats.csv         -> <out_prefix>_kernel_stats.csv (our kernels only)
  <prof_dir>/p*/**/_counter_collection.csv   -> <out_prefix>_pmc_summary.csv  (mean per dispatch)
"""
import csv, glob, os, sys
from collections import defaultdict


def short(name):
    for key in ('window_claim_kernel', 'unproject_columns_kernel', 'assemble_zslabs_kernel',
                'conv1x1_nhwc_kernel',
                'window_brick_kernel', 'gather_kernel', 'footprint_kernel', 'zcolumns_kernel',
                'window_backward_brick_kernel', 'unproject_backward_kernel', 'unrepack_kernel',
                'unproject_planes_kernel', 'unproject_group_kernel', 'unproject_kernel',
                'repack_wide_kernel', 'repack_kernel', 'project_points_kernel'):
        if key in name:
            i = name.index(key)
            j = name.find('(', i)
            return name[i:j if j > 0 else None]
    return None


def main(prof, out):
    ks = glob.glob(os.path.join(prof, 'kt', '**', '*_kernel_stats.csv'), recursive=True)
    if ks:
        with open(ks[0]) as f, open(out + '_kernel_stats.csv', 'w', newline='') as g:
            r = csv.reader(f)
            w = csv.writer(g)
            w.writerow(next(r))
            for row in r:
                s = short(row[0])
                w.writerow([s if s else row[0][:60]] + row[1:])
    acc = defaultdict(lambda: [0, 0.0])
    for path in sorted(glob.glob(os.path.join(prof, 'p*', '**', '*_counter_collection.csv'),
                                 recursive=True)):
        per = defaultdict(float)  # (kernel, dispatch, counter) -> value summed over instances
        with open(path) as f:
            for row in csv.DictReader(f):
                s = short(row['Kernel_Name'])
                if s:
                    per[(s, row['Dispatch_Id'], row['Counter_Name'])] += float(row['Counter_Value'])
        for (s, _, c), v in per.items():
            acc[(s, c)][0] += 1
            acc[(s, c)][1] += v
    with open(out + '_pmc_summary.csv', 'w', newline='') as g:
        w = csv.writer(g)
        w.writerow(['kernel', 'counter', 'dispatches', 'mean_per_dispatch'])
        for (s, c), (n, v) in sorted(acc.items()):
            w.writerow([s, c, n, '%.1f' % (v / n)])


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
