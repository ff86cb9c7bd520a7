#!/bin/bash
# Quick GPU look: bench line (+ per-kernel event times) of one config; optional pytest subset first.
#   tools/r3_quick.sh <tag> [c2|c5] [pytest -k expression]
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
tag="${1:-q}"; cfg="${2:-c2}"; sel="$3"
O="gpurun_out/$tag"; mkdir -p "$O"
if [ -n "$sel" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q --timeout 600 -k "$sel" > "$O/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest.txt"
fi
steps=50; [ "$cfg" = c5 ] && steps=10
python bench.py --config $cfg --steps $steps --warmup 5 --no-cpu-baseline > "$O/bench_$cfg.json" 2> "$O/bench_$cfg.err" || tail -5 "$O/bench_$cfg.err"
python - "$O/bench_$cfg.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("ms_per_step %.4f  value %.3f M  module %.4f" % (d["ms_per_step"], d["value"] / 1e6, d.get("module", {}).get("ms_per_step", 0)))
for k, v in d["roofline"].get("kernel_ms", {}).items():
    print("  %-55s %.4f" % (k, v))
PY
