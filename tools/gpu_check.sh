#!/bin/bash
# One GPU-box pass: hardware microbench, smoke, GPU parity tests, bench.  Outputs -> gpurun_out/
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
what="${1:-all}"
if [[ "$what" == all || "$what" == micro ]]; then
  hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/microbench.hip -o /tmp/microbench 2>/dev/null \
    && timeout 300 /tmp/microbench > gpurun_out/microbench.txt 2>&1
  cat gpurun_out/microbench.txt
fi
if [[ "$what" == all || "$what" == smoke ]]; then
  timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.txt 2>&1; echo "smoke rc=$?"; tail -5 gpurun_out/smoke.txt
fi
if [[ "$what" == all || "$what" == test ]]; then
  timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest rc=$?"
  tail -40 gpurun_out/pytest_gpu.txt
fi
if [[ "$what" == all || "$what" == bench ]]; then
  timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.txt 2> gpurun_out/bench.err; echo "bench rc=$?"
  tail -3 gpurun_out/bench.err; cat gpurun_out/bench.txt
fi
