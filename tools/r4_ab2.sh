#!/bin/bash
# same-box A/B (base vs work) on C2 plus the sparse configurations
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r4ab gpurun_ab/work; cp ssl_amd/csrc/libssg_hip*.so gpurun_ab/work/
if [ -n "$1" ]; then timeout 1200 python -m pytest tests -m gpu -x -q --timeout 900 -k "$1" 2>&1 | tail -3; fi
for rep in 1 2; do
  for t in base work; do
    cp gpurun_ab/$t/libssg_hip.so gpurun_ab/$t/libssg_hip_prof.so ssl_amd/csrc/
    for c in b1 b4 c4; do python tools/sparse_step.py $c 100 2>&1 | grep ms/step | sed "s/^/$t  /"; done
  done
done | tee gpurun_out/r4ab/ab_sparse.txt
cp gpurun_ab/work/libssg_hip*.so ssl_amd/csrc/
tools/ab_run.sh base work --steps 100 --warmup 20 2>&1 | tee gpurun_out/r4ab/ab.txt
