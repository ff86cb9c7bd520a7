#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/dense_bwd_check.py > gpurun_out/dense_bwd_check.txt 2>&1; echo "check rc=$?"; cat gpurun_out/dense_bwd_check.txt | tail -40
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -x > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.txt
