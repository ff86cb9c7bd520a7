#!/bin/bash
# Timeline of ONE loss step from a rocprofv3 kernel trace: start offset, duration and gap to the previous kernel's end,
# per stream -- shows where the step idles.   tools/step_timeline.sh [c2|c5]
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
R="$PWD"; cfg="${1:-c2}"; O="$R/gpurun_out/timeline_$cfg"; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$O" -o tl -- python "$R/bench.py" --config $cfg --steps 6 --warmup 3 --no-cpu-baseline --no-module --no-extra > "$O/log.txt" 2>&1
cd "$R"
python - "$O" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# one step = from an edge_count launch to the next one; take the 6th (inside the timed loop)
idx = [i for i, r in enumerate(rows) if 'band_count' in r['Kernel_Name'] or 'edge_count' in r['Kernel_Name']]
a, b = idx[5], idx[6]
t0 = int(rows[a]['Start_Timestamp']); prev_end = t0
print("step span %.1f us, %d kernels" % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3, b - a))
busy = 0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%8.1f us  dur %7.1f  gap %6.1f  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get('Queue_Id', '?'), r['Kernel_Name'][:70].replace('void ssg::', '')))
    prev_end = max(prev_end, e)
PY
