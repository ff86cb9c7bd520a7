#!/bin/bash
# same-box A/B of the k_s 49 materialising step's schedule on the PROFILING build: the writer of the normalised rows as the
# row pass itself (SSG_MAT_BESIDE=0, rounds 3-5) against the fused step's row pass + the writer beside the backward (=1)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_ab/work; cp ssl_amd/csrc/libssg_hip.so gpurun_ab/work/
cp ssl_amd/csrc/libssg_hip_prof.so ssl_amd/csrc/libssg_hip.so
for rep in 1 2 3; do
  for v in 0 1; do
    ms=$(SSG_MAT_BESIDE=$v python bench.py --config c5 --no-cpu-baseline --no-module --no-extra --no-kernel-table --steps 20 --warmup 5 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | cut -d' ' -f2 | cut -c1-6)
    echo "SSG_MAT_BESIDE=$v c5 $ms ms"
  done
done
cp gpurun_ab/work/libssg_hip.so ssl_amd/csrc/libssg_hip.so
