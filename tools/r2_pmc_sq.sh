#!/bin/bash
# SQ counter passes (separate runs, --kernel-trace only) over the bench; prints per-dispatch means for ssg_ kernels
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
R="$PWD"; mkdir -p gpurun_out/pmc; cd /tmp; export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM"; do
  i=$((i+1))
  rm -rf "$R/gpurun_out/pmc/p$i"
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$R/gpurun_out/pmc/p$i" -o pmc -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/pmc/p$i.log" 2>&1
  echo "pass $i rc=$?"
done
cd "$R"
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc/p[12]/pmc_counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:70]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        seen[k].add(r['Dispatch_Id'])
    for k, d in agg.items():
        if 'ssg_' not in k: continue
        n = len(seen[k])
        print(k.replace('void ssg::',''), ' '.join('%s=%.4g' % (c.replace('SQ_',''), v / n) for c, v in sorted(d.items())))
PY
find gpurun_out/pmc -name "*kernel_trace.csv" -delete
