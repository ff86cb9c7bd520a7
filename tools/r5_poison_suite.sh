#!/bin/bash
# The GPU suite under poison (tests/conftest.py), one pattern per run -> gpurun_out/r5poison/<pattern>.txt
#   global memory: the allocator's free blocks filled before every test      tools/r5_poison_suite.sh "int:3 nan int:1"
#   LDS: profiling build, every launch preceded by an LDS fill of every CU   tools/r5_poison_suite.sh "lds:00640064 lds:7fc00000"
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
O=gpurun_out/r5poison; mkdir -p $O
for pat in ${1:-int:3 nan int:1 f32:1e30}; do
  f=$O/$(echo $pat | tr ':' '_').txt
  if [ "${pat%%:*}" = lds ]; then
    SSL_AMD_TEST_LDS_POISON=${pat#lds:} timeout 2400 python -m pytest tests -m gpu -q --timeout 900 ${2:+-k "$2"} > $f 2>&1
  else
    SSL_AMD_TEST_POISON=$pat timeout 1500 python -m pytest tests -m gpu -q --timeout 900 ${2:+-k "$2"} > $f 2>&1
  fi
  echo "== $pat rc=$? $(tail -1 $f)"; grep -n "^FAILED" $f | head -20
done
