#!/bin/bash
# kernel timeline of one iteration of the drop-in module (SSGLoss forward + autograd backward) at C2
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r6modtl; rm -rf $O; mkdir -p $O
cat > /tmp/mod_step.py <<PY
import sys, numpy as np, torch
sys.path.insert(0, "$R")
from ssl_amd import synth, SSGLoss
dev = torch.device("cuda:0")
sr, gt, m = synth.make_batch(16, 256, 256)
a, b, mm = (torch.as_tensor(x, device=dev) for x in (sr, gt, m))
crit = SSGLoss(25, 9, 1.0, True, 1e3, 1e3, capacity=int(m.sum()) + 1024)
x = a.clone().requires_grad_(True)
def one():
    x.grad = None
    l1, kl = crit(x, b, mm)
    (l1 + kl).backward()
for _ in range(30): one()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): one()
e1.record(); e1.synchronize()
print("module ms/step %.4f" % (e0.elapsed_time(e1) / 50))
PY
python /tmp/mod_step.py 2>/dev/null
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O -o tl -- python /tmp/mod_step.py > $O/log.txt 2>&1)
python - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'band_count' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]['Start_Timestamp']); prev_end = t0
print("iteration span %.1f us, %d kernels" % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3, b - a))
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%8.1f us  dur %7.1f  gap %6.1f  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get('Queue_Id', '?'), r['Kernel_Name'][:70].replace('void ssg::', '')))
    prev_end = max(prev_end, e)
PY
find $O -name "*kernel_trace.csv" -delete
