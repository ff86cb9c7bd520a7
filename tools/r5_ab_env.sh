#!/bin/bash
# same-box A/B of PROFILING-build environment switches: each argument is "tag:VAR=VAL,VAR=VAL" (or "tag:" for none);
# C2 bench line + sparse steps on libssg_hip_prof.so used as the product library; three alternations
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_ab/work; cp ssl_amd/csrc/libssg_hip.so gpurun_ab/work/
cp ssl_amd/csrc/libssg_hip_prof.so ssl_amd/csrc/libssg_hip.so
for rep in 1 2 3; do
  for a in "$@"; do
    tag=${a%%:*}; envs=$(echo "${a#*:}" | tr ',' ' ')
    c2=$(env $envs python bench.py --no-cpu-baseline --no-module --no-extra --no-kernel-table --steps 100 --warmup 20 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | cut -d' ' -f2 | cut -c1-6)
    echo "$tag c2 $c2 $(for c in c1 b1 b4 c4 i1; do env $envs python tools/sparse_step.py $c 200 2>&1 | grep ms/step | sed 's/N=[0-9]* //; s/ ms\/step//'; done | tr '\n' ' ')"
  done
done
cp gpurun_ab/work/libssg_hip.so ssl_amd/csrc/libssg_hip.so
