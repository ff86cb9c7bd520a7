#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
O=$PWD/gpurun_out/r5scale; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/p -o t -- python tools/r5_strip_scale.py 72 144 288 512 > $O/log.txt 2>&1
cat $O/log.txt | grep "^H"
python - $O <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/p/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "strip" in k or "rows_tm" in k or "bwd_dense" in k:
        acc[(k.split("(")[0].replace("void ssg::", "")[:40], int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (k, g), v in sorted(acc.items()):
    print("%-42s workgroups %5d  mean %9.1f us (n=%d)" % (k, g, sum(v) / len(v), len(v)))
PY
rm -rf $O/p
