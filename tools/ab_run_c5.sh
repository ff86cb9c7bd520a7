#!/bin/bash
# same-box A/B of two library builds on the C5 steps (materialising and fused): tools/ab_run_c5.sh <tagA> <tagB>
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
A="$1"; B="$2"
mkdir -p gpurun_ab/work; cp ssl_amd/csrc/libssg_hip*.so gpurun_ab/work/
for rep in 1 2 3; do
  for t in "$A" "$B"; do
    cp gpurun_ab/$t/libssg_hip.so gpurun_ab/$t/libssg_hip_prof.so ssl_amd/csrc/
    m=$(python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-module --no-extra 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)
    f=$(python bench.py --config c5 --no-ssg-output --steps 10 --warmup 3 --no-cpu-baseline --no-module --no-extra 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)
    echo "$t materialising $m fused $f"
  done
done
cp gpurun_ab/work/libssg_hip*.so ssl_amd/csrc/
