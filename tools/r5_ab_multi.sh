#!/bin/bash
# alternate library builds (gpurun_ab/<tag>/) on one box: C2 bench line + sparse steps; three alternations
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_ab/work; cp ssl_amd/csrc/libssg_hip*.so gpurun_ab/work/
for rep in 1 2 3; do
  for t in "$@"; do
    cp gpurun_ab/$t/libssg_hip.so gpurun_ab/$t/libssg_hip_prof.so ssl_amd/csrc/
    c2=$(python bench.py --no-cpu-baseline --no-module --no-extra --no-kernel-table --steps 100 --warmup 20 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | cut -d' ' -f2 | cut -c1-6)
    echo "$t c2 $c2 $(for c in c1 b1 b4 c4 i1; do python tools/sparse_step.py $c 200 2>&1 | grep ms/step | sed 's/N=[0-9]* //; s/ ms\/step//'; done | tr '\n' ' ')"
  done
done
cp gpurun_ab/work/libssg_hip*.so ssl_amd/csrc/
