#!/bin/bash
# PMC passes (one counter set per run, --kernel-trace only beside them) of one bench configuration; per-kernel means.
#   tools/r5_pmc.sh <tag> "<bench flags>" [kernel substring filter]
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
tag=$1; flags=$2; filt=${3:-ssg_}
O=$PWD/gpurun_out/r5pmc/$tag; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o pmc -- python bench.py $flags --steps 3 --warmup 1 --no-cpu-baseline --no-module --no-extra --no-kernel-table > $O/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python - "$O" "$filt" <<'PY'
import csv, glob, sys, collections
O, filt = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if filt in k:
            acc[k.split("(")[0].replace("void ssg::", "")[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print("==", k)
    for c, v in sorted(cs.items()):
        print("   %-28s %14.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
