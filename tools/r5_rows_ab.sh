#!/bin/bash
# Same-box A/B of the materialising C5 step: round-3 head tree (gpurun_ab/r3tree) against the working tree, each under
# rocprofv3 --kernel-trace --stats (why did ssg_rows_tm_mat go 2.36 -> 3.15 ms with unchanged source?)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
O=$PWD/gpurun_out/r5ab; mkdir -p $O
run() {  # tag dir
  ( cd $2 && rocprofv3 --kernel-trace --stats --output-format csv -d $O/$1 -o $1 -- python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-module --no-extra > $O/$1.json 2> $O/$1.err )
  f=$(find $O/$1 -name "*kernel_stats.csv" | head -1)
  echo "== $1: $(grep -o '"ms_per_step": [0-9.]*' $O/$1.json | head -1)"; head -5 "$f" | cut -d, -f1-4 | cut -c1-150
}
for rep in 1 2; do
  run r3_$rep gpurun_ab/r3tree
  run head_$rep .
done
