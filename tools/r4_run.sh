#!/bin/bash
# run an arbitrary command line on the GPU box, output -> gpurun_out/r4run.txt
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; bash -c "$*" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4run.txt
