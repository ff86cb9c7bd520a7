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
  timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest rc=$?"
  tail -40 gpurun_out/pytest_gpu.txt
fi
if [[ "$what" == all || "$what" == bench ]]; then
  timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.txt 2> gpurun_out/bench.err; echo "bench rc=$?"
  tail -3 gpurun_out/bench.err; cat gpurun_out/bench.txt
fi
if [[ "$what" == micro2 || "$what" == all2 ]]; then
  hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/microbench2.hip -o /tmp/microbench2 2>/dev/null \
    && timeout 300 /tmp/microbench2 > gpurun_out/microbench2.txt 2>&1
  cat gpurun_out/microbench2.txt
fi
if [[ "$what" == prof || "$what" == all2 ]]; then
  rm -rf gpurun_out/prof; cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof_bench.txt" 2>&1
  cd "$GRAFT_REPO_ROOT"; find gpurun_out/prof -name "*stats*" | head; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
fi
