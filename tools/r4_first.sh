#!/bin/bash
# Round-4 first look: the unchanged-loop tests, the whole GPU suite, the driver's bench line with the new extras
# (ref_api, operator, c4, c2_maskgen) and rocprofv3 kernel stats of the C4 step -> gpurun_out/r4a/
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
R="$PWD"; O="$R/gpurun_out/r4a"; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ref_api.py -m gpu -q --timeout 600 > "$O/pytest_ref_api.txt" 2>&1; echo "pytest ref_api rc=$?"; tail -15 "$O/pytest_ref_api.txt"
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 --deselect tests/test_gpu_ref_api.py > "$O/pytest_all.txt" 2>&1; echo "pytest all rc=$?"; tail -5 "$O/pytest_all.txt"
python bench.py > "$O/r4_bench_c2.json" 2> "$O/bench_c2.err"; echo "bench c2 rc=$?"; tail -3 "$O/bench_c2.err"
python bench.py --config c4 --no-extra > "$O/r4_bench_c4.json" 2> "$O/bench_c4.err"; echo "bench c4 rc=$?"; tail -3 "$O/bench_c4.err"
cd /tmp
SSG_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_c4" -o bench -- python "$R/bench.py" --config c4 --steps 20 --warmup 3 --no-cpu-baseline --no-module --no-extra --no-kernel-table > "$O/prof_c4.log" 2>&1
f=$(find "$O/prof_c4" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/r4_bench_c4_kernel_stats.csv"
find "$O/prof_c4" -name "*kernel_trace.csv" -delete
cd "$R"
python - <<'PY'
import json
for f in ("r4_bench_c2.json", "r4_bench_c4.json"):
    try:
        d = json.load(open("gpurun_out/r4a/" + f))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, "ms_per_step %.4f value %.3f M module %.4f" % (d["ms_per_step"], d["value"] / 1e6, d.get("module", {}).get("ms_per_step", 0)))
    for k, v in d["roofline"].get("kernel_ms", {}).items():
        print("  %-55s %.4f" % (k, v))
    ex = d.get("extra", {})
    for k, v in ex.items():
        if k == "ref_api":
            print("  ref_api floor %.3f" % v["caller_floor_ms"], {m: round(x["ms"], 3) for m, x in v.items() if isinstance(x, dict)})
        elif k == "operator":
            print("  operator", v["fwd_ms"], v["fwd_bwd_ms"], v["edge_px"])
        else:
            print("  extra %-12s %.4f ms %.2f M" % (k, v["ms_per_step"], v["value"] / 1e6))
PY
head -30 "$O/r4_bench_c4_kernel_stats.csv"
