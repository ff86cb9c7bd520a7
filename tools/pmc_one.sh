#!/bin/bash
# One PMC pass (--kernel-trace only) over bench.py; usage: tools/pmc_one.sh "<counters>" [bench args...]
# prints per-dispatch means of the ssg_ kernels
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
R="$PWD"; mkdir -p gpurun_out/pmc; cd /tmp; export TMPDIR=/tmp
set_="$1"; shift
rm -rf "$R/gpurun_out/pmc/one"
timeout 600 rocprofv3 --pmc $set_ --kernel-trace --output-format csv -d "$R/gpurun_out/pmc/one" -o pmc -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$R/gpurun_out/pmc/one.log" 2>&1
echo "rc=$?"
cd "$R"
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc/one/pmc_counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        seen[k].add(r['Dispatch_Id'])
    for k, d in agg.items():
        if 'ssg_' not in k: continue
        n = len(seen[k])
        print(k.replace('void ssg::',''), ' '.join('%s=%.4g' % (c, v / n) for c, v in sorted(d.items())))
PY
find gpurun_out/pmc -name "*kernel_trace.csv" -delete
