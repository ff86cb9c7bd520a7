#!/bin/bash
# Sparse regime: per-kernel stats of C1 / Bernoulli 1 % / 4 % / C4, eager vs HIP-graph replay -> gpurun_out/r4s/
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
R="$PWD"; O="$R/gpurun_out/r4s"; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
for c in c1 b1 b4 c4; do
  python tools/sparse_step.py $c 100; python tools/sparse_step.py $c 100 --graph
  SSG_BWD_QSPLIT=1 python tools/sparse_step.py $c 100 | sed 's/$/  (qsplit 1)/'
done 2>&1 | grep -v amdgpu.ids | tee "$O/sparse_times.txt"
cd /tmp
for c in b1 c4 c1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$c" -o t -- python "$R/tools/sparse_step.py" $c 30 > "$O/prof_$c.log" 2>&1
  f=$(find "$O/prof_$c" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/r4_sparse_${c}_kernel_stats.csv"
  find "$O/prof_$c" -name "*kernel_trace.csv" -delete
  echo "== $c"; cut -d, -f1-4 "$O/r4_sparse_${c}_kernel_stats.csv" | head -24
done
cd "$R"
python bench.py --no-extra --no-cpu-baseline --steps 50 --warmup 10 > "$O/bench_c2.json" 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4s/bench_c2.json"))
print("c2 ms_per_step %.4f" % d["ms_per_step"]); [print("  %-55s %.4f" % kv) for kv in d["roofline"]["kernel_ms"].items()]
PY
timeout 900 python -m pytest tests -m gpu -x -q --timeout 600 -k "dense or f10 or f9 or c2_full or deterministic or stress" 2>&1 | tail -3
