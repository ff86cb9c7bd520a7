#!/bin/bash
# two-waves-per-tile (role split) tile-major k_s 49 dense backward against the one-wave kernel, same box:
# parity tests with the split on, then C5 fused / materialised steps alternating SSG_BWD_TM_SPLIT=1/0 -> gpurun_out/r4_c5_split.txt
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests -m gpu -x -q -k "49 or k49 or c5 or C5 or tile_major or fused" 2>&1 | tail -3
for rep in 1 2 3; do for sp in 1 0; do
  echo -n "split=$sp  "
  SSG_BWD_TM_SPLIT=$sp python bench.py --config c5 --no-ssg-output --steps 20 --warmup 5 --no-cpu-baseline --no-module --no-extra --no-kernel-table 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('c5 fused %.4f' % d['ms_per_step'], end='  ')"
  SSG_BWD_TM_SPLIT=$sp python bench.py --config c5 --steps 20 --warmup 5 --no-cpu-baseline --no-module --no-extra --no-kernel-table 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('c5 %.4f' % d['ms_per_step'])"
done; done
} 2>&1 | tee gpurun_out/r4_c5_split.txt
