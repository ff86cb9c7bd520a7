#!/bin/bash
# tile-major check + C5 fused bench under rocprofv3 --kernel-trace --stats: tile-major rows with the strip forward (11),
# with the tile forward only (10), row-major rows (00)
#   tools/tm_quick.sh <tag> [nocheck|checkonly]
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
O=$R/gpurun_out/${1:-tm}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
if [ "$2" != nocheck ]; then
  (cd $R && timeout 900 python tools/tm_check.py) > $O/check.txt 2>&1; echo "check rc=$?"; cat $O/check.txt
fi
[ "$2" = checkonly ] && exit 0
for tmj in 11 10 00; do
  SSG_TILE_MAJOR=${tmj:0:1} SSG_STRIPS=${tmj:1:1} timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tm$tmj -o bench -- python $R/bench.py --config c5 --no-ssg-output --no-extra --no-module --steps 10 --warmup 3 --no-cpu-baseline > $O/c5_fused_tm$tmj.json 2> $O/c5_fused_tm$tmj.err || tail -5 $O/c5_fused_tm$tmj.err
  grep -o '"ms_per_step": [0-9.]*' $O/c5_fused_tm$tmj.json | head -1
  f=$(find $O/prof_tm$tmj -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_tm$tmj.csv && head -9 $f | cut -c1-150
  find $O/prof_tm$tmj -name "*kernel_trace.csv" -delete
done
