#!/bin/bash
# rocprofv3 kernel stats + two SQ counter passes of the fused C5 step (tile-major dense backward) -> gpurun_out/r4_c5f_pmc/
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
R="$PWD"; O="$R/gpurun_out/r4_c5f_pmc"; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp; cd /tmp
F="--config c5 --no-ssg-output --no-kernel-table --steps 10 --warmup 3 --no-cpu-baseline --no-module --no-extra"
SSG_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o bench -- python "$R/bench.py" $F > "$O/prof.log" 2>&1
f=$(find "$O/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-200
find "$O/prof" -name "*kernel_trace.csv" -delete
export SSG_OVERLAP=0
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$O/p$i" -o pmc -- python "$R/bench.py" --config c5 --no-ssg-output --no-kernel-table --steps 3 --warmup 1 --no-cpu-baseline --no-module --no-extra > "$O/p$i.log" 2>&1
  find "$O/p$i" -name "*kernel_trace.csv" -delete
  python - "$O/p$i" <<'PY'
import csv, sys, glob, collections
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if 'bwd_dense' in r['Kernel_Name'] and 'true' in r['Kernel_Name'].split('(')[0]: acc[r['Kernel_Name'].split('(')[0][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items(): print(k, ' '.join('%s=%.4g' % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
PY
done
