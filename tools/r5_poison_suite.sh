#!/bin/bash
# The GPU suite with the allocator's free memory poisoned before every test (tests/conftest.py), one pattern per run.
#   tools/gpu.sh 3000 'bash tools/r5_poison_suite.sh "int:3 nan int:1"'   -> gpurun_out/r5poison/<pattern>.txt
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
O=gpurun_out/r5poison; mkdir -p $O
for pat in ${1:-int:3 nan int:1 f32:1e30}; do
  f=$O/$(echo $pat | tr ':' '_').txt
  SSL_AMD_TEST_POISON=$pat timeout 1500 python -m pytest tests -m gpu -q --timeout 900 ${2:+-k "$2"} > $f 2>&1
  echo "== $pat rc=$? $(tail -1 $f)"; grep -n "^FAILED" $f | head -20
done
