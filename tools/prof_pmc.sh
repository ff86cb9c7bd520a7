#!/bin/bash
# PMC passes over the bench (separate runs, --kernel-trace only, as the node policy requires).
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
R="$PWD"; mkdir -p gpurun_out/pmc; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > "$R/gpurun_out/pmc/counters_list.txt" 2>&1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$R/gpurun_out/pmc/p$i" -o pmc -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/pmc/p$i.log" 2>&1
  echo "pass $i ($set) rc=$?"
done
cd "$R"
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc/p*/pmc_counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    # number of dispatches per kernel
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        seen[r['Kernel_Name'][:60]].add(r['Dispatch_Id'])
    print('==', f)
    for k, d in agg.items():
        if 'ssg_' not in k: continue
        n = len(seen[k])
        print(' ', k, 'dispatches', n)
        for c, v in d.items(): print('     %-28s %.4g per dispatch' % (c, v / n))
PY
