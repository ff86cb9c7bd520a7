#!/bin/bash
# instruction-cache counters of the fused C5 step's kernels (one pass) -> stdout
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
R="$PWD"; O="$R/gpurun_out/r4_c5f_icache"; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp; cd /tmp
export SSG_OVERLAP=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU"; do
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$O/p" -o pmc -- python "$R/bench.py" --config ${1:-c5} ${2:---no-ssg-output} --no-kernel-table --steps 3 --warmup 1 --no-cpu-baseline --no-module --no-extra > "$O/p.log" 2>&1
  find "$O/p" -name "*kernel_trace.csv" -delete
  python - "$O/p" <<'PY'
import csv, sys, glob, collections
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'].split('(')[0][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        if sum(d.get('SQ_WAVE_CYCLES', [0])) / max(len(d.get('SQ_WAVE_CYCLES', [1])), 1) > 1e7:
            print(k, ' '.join('%s=%.4g' % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
PY
done
