#!/bin/bash
# dense threshold 20 vs 28 on the other headline shapes: C5 (k_s 49), C1, the unchanged caller loop -> gpurun_out/r4thr3.txt
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
for rep in 1 2; do
for thr in 20 28; do
  echo -n "thr=$thr  "
  for c in c5 c1; do
    SSG_DENSE_THR=$thr python bench.py --config $c --no-extra --no-cpu-baseline --no-module --no-kernel-table --steps 50 --warmup 10 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$c %.4f' % d['ms_per_step'], end='  ')"
  done
  SSG_DENSE_THR=$thr python bench.py --no-cpu-baseline --no-module --no-kernel-table --steps 30 --warmup 10 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read()); e = d['extra']; print('c5_fused %.3f c2_fused %.3f ref_api %s' % (e['c5_fused']['ms_per_step'], e['c2_fused']['ms_per_step'], {k: v for k, v in e['ref_api'].items() if 'ms' in k}))"
done; done | tee gpurun_out/r4thr3.txt
