#!/bin/bash
# kernel trace of tools/train_planes_time.py (run through gpurun from the repo root)
set -o pipefail
repo=$PWD; out=$repo/gpurun_out/prof_train_planes; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 $repo/tools/train_planes_time.py $1 > $out/run.log 2>&1 || exit 1
f=$(find $out/kt -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:14]:
    print('%-90s calls %5s avg %9.1f us total %8.2f ms' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
