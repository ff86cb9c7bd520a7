#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; R="$PWD"; mkdir -p gpurun_out/dpmc; cd /tmp; export TMPDIR=/tmp SSG_DENSE_THR=1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$R/gpurun_out/dpmc/p$i" -o pmc -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
cd "$R"
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/dpmc/p*/pmc_counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:44]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); seen[k].add(r['Dispatch_Id'])
    for k, d in agg.items():
        if 'dense' in k: print(k, {c: '%.4g' % (v/len(seen[k])) for c, v in d.items()})
PY
