#!/bin/bash
# Dense threshold x backward offset split at the sparse configurations -> gpurun_out/r4thr.txt
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
for c in c4 b4 b1; do
  for thr in 6 10 14 18 22 28; do
    for qs in 0 1; do
      echo -n "thr=$thr qsplit=$qs  "; SSG_DENSE_THR=$thr SSG_BWD_QSPLIT=$qs python tools/sparse_step.py $c 100 2>&1 | grep ms/step
    done
  done
done | tee gpurun_out/r4thr.txt
