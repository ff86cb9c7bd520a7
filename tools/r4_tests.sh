#!/bin/bash
# GPU test-suite (optionally a -k selection) -> gpurun_out/r4t/
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
O="gpurun_out/r4t"; mkdir -p "$O"
timeout 2000 python -m pytest tests -m gpu -q --timeout 900 ${1:+-k "$1"} > "$O/pytest.txt" 2>&1; echo "pytest rc=$?"
grep -n "^E \|^FAILED\|passed\|failed" "$O/pytest.txt" | head -40
