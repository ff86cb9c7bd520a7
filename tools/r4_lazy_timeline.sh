#!/bin/bash
# Kernel timeline of ONE deferred caller loop at C2 (rocprofv3 kernel trace of tools/r4_lazy_loop_run.py)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
R="$PWD"; O="$R/gpurun_out/lazy_tl"; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$O" -o tl -- python "$R/tools/r4_lazy_loop_run.py" > "$O/log.txt" 2>&1
cd "$R"; tail -3 "$O/log.txt"
python - "$O" <<'PY' | tee gpurun_out/lazy_tl/timeline.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
for g in glob.glob(sys.argv[1] + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(g)):
        r['Kernel_Name'] = 'MEMCPY ' + r.get('Direction', ''); rows.append(r)
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'edge_count' in r['Kernel_Name']]
# one loop = from the end of the previous loop's last kernel (grad flush) to this loop's last kernel
a0, a1 = idx[-3], idx[-2]
# walk back from a1 to the first kernel after the previous loop's backward
prev_last = max(i for i in range(a0, a1) if 'grad_fix_flush' in rows[i]['Kernel_Name'] or 'ssg_bwd' in rows[i]['Kernel_Name'])
nxt_last = max(i for i in range(a1, idx[-1]) if 'grad_fix_flush' in rows[i]['Kernel_Name'] or 'ssg_bwd' in rows[i]['Kernel_Name'])
t0 = int(rows[prev_last]['End_Timestamp']); prev_end = t0
busy = 0
for r in rows[prev_last + 1:nxt_last + 8]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%8.1f us  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r['Kernel_Name'][:100].replace('void ssg::', '')))
    prev_end = max(prev_end, e)
PY
