#!/bin/bash
# kernel timeline of one step of a sparse configuration: tools/r5_timeline.sh <c1|b1|b4|c4>
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
cfg=$1; O=$PWD/gpurun_out/r5tl_$cfg; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -o tl -- python tools/sparse_step.py $cfg 20 > $O/log.txt 2>&1
grep ms/step $O/log.txt
python - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'band_count' in r['Kernel_Name'] or 'edge_count' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]['Start_Timestamp']); prev_end = t0
print("step span %.1f us, %d kernels" % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3, b - a))
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%8.1f us  dur %7.1f  gap %6.1f  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get('Queue_Id', '?'), r['Kernel_Name'][:64].replace('void ssg::', '')))
    prev_end = max(prev_end, e)
PY
