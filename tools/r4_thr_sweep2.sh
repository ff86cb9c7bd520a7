#!/bin/bash
# dense threshold 16 .. 28 with the round-4 kernels: C2 (bench, 100 steps), C4, Bernoulli 4 %, 16 % -> gpurun_out/r4thr2.txt
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
for rep in 1 2; do
for thr in 16 20 24 28; do
  echo -n "thr=$thr  "
  SSG_DENSE_THR=$thr python bench.py --no-extra --no-cpu-baseline --no-module --no-kernel-table --steps 100 --warmup 20 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('c2 %.4f' % d['ms_per_step'], end='  ')"
  for c in c4 b4; do SSG_DENSE_THR=$thr python tools/sparse_step.py $c 100 2>&1 | grep ms/step | tr '\n' ' '; done; echo
done; done | tee gpurun_out/r4thr2.txt
