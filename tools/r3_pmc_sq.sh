#!/bin/bash
# SQ counters of every kernel of one config (two passes, counters only): tools/r3_pmc_sq.sh <tag> [c2|c5]
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
R="$PWD"; tag="${1:-pmc}"; cfg="${2:-c2}"; O="$R/gpurun_out/$tag"; mkdir -p "$O"; export TMPDIR=/tmp SSG_OVERLAP=0
cd /tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" \
           "SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_LEVEL_WAVES SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$O/p$i" -o pmc -- python "$R/bench.py" --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-module --no-extra > "$O/p$i.log" 2>&1
  echo "pmc pass $i rc=$?"
  find "$O/p$i" -name "*kernel_trace.csv" -delete
done
cd "$R"
python - "$O" <<'PY' > "$O/summary.txt"
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + '/p*/pmc_counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:72]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); seen[k].add(r['Dispatch_Id'])
    for k, d in agg.items():
        if 'ssg_' not in k: continue
        n = len(seen[k])
        print(k.replace('void ssg::', ''), ' '.join('%s=%.4g' % (c, v / n) for c, v in sorted(d.items())))
PY
cat "$O/summary.txt"
