#!/bin/bash
# materialising C5 step: tile-major rows (ssg_rows_tm_mat writes the SSG tensors) vs row-major (SSG_TILE_MAJOR=0)
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
O=$R/gpurun_out/${1:-c5mat}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
for tmj in 1 0; do
  SSG_TILE_MAJOR=$tmj timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$tmj -o bench -- python $R/bench.py --config c5 --no-kernel-table --no-extra --no-module --steps 10 --warmup 3 --no-cpu-baseline > $O/c5_tm$tmj.json 2> $O/c5_tm$tmj.err || tail -5 $O/c5_tm$tmj.err
  python - $O/c5_tm$tmj.json $(find $O/prof$tmj -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, json, sys
d = json.load(open(sys.argv[1]))
print("ms_per_step %.3f  l1 %.6g kl %.6g" % (d["ms_per_step"], d["config"]["l1"], d["config"]["kl"]))
for r in list(csv.DictReader(open(sys.argv[2])))[:5]:
    print("   %-60s %4s x %8.1f us" % (r["Name"][10:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
