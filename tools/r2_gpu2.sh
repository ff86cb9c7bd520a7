#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/dense_bwd_check.py > gpurun_out/dense_bwd_check.txt 2>&1; echo "check rc=$?"; cat gpurun_out/dense_bwd_check.txt | tail -22
bash tools/r2_prof.sh b | head -12
