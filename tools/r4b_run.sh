#!/bin/bash
# round-4 (second session) iteration helper: GPU tests by -k expression, then same-box A/B of base vs work on C2 / sparse
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r4b
if [ -n "$1" ]; then timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 -k "$1" 2>&1 | tail -15; fi
[ "$2" = "noab" ] && exit 0
bash tools/r4_ab2.sh 2>&1 | tail -40
