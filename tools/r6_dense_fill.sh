#!/bin/bash
# tools/r6_dense_fill.sh: per-kernel durations of the C4-type step for B = 2 .. 16 images (see r6_dense_fill.py)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
for B in 2 4 8 16; do
  O=/tmp/df_$B; rm -rf $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python tools/r6_dense_fill.py $B 30 2>/dev/null | grep "^B="
  python - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name'].replace('void ssg::', '').replace('ssg::', '')
    if n.startswith('ssg_') and float(r['AverageNs']) > 8000:
        print("   %-64s avg %8.1f us  min %8.1f" % (n[:64], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
done
