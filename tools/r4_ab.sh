#!/bin/bash
# same-box A/B of the working tree's build against gpurun_ab/base (+ a -k selection of the GPU tests first)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r4ab
if [ -n "$1" ]; then timeout 1200 python -m pytest tests -m gpu -x -q --timeout 900 -k "$1" 2>&1 | tail -3; fi
tools/ab_run.sh base work --steps 100 --warmup 20 2>&1 | tee gpurun_out/r4ab/ab.txt
