#!/bin/bash
# full GPU suite + the driver's bench line -> gpurun_out/r5c/
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
O=gpurun_out/r5c; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.txt 2>&1; echo "pytest rc=$?"; grep -n "^FAILED\|passed\|failed\|^E " $O/pytest.txt | head -20
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.txt
python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?"; tail -3 $O/bench_c2.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5c/bench_c2.json"))
c = d["config"]
print("c2 ms %.4f block2 %.4f block3 %.4f prewarm %d steps %.0f ms sclk %s module %.4f" % (d["ms_per_step"], c["ms_per_step_block2"], c["ms_per_step_block3"], c["prewarm_steps"], c["prewarm_ms"], c["sclk_after_timed_region"], d["module"]["ms_per_step"]))
print("kernel_ms", {k.split("(")[0][:28]: round(v, 3) for k, v in d["roofline"].get("kernel_ms", {}).items()})
for k, v in d["extra"].items():
    if k == "ref_api": print("ref_api floor %.3f" % v["caller_floor_ms"], {m: round(x["ms"], 3) for m, x in v.items() if isinstance(x, dict)})
    elif k == "ref_api_dm": print("ref_api_dm", {m: round(x["ms"], 3) for m, x in v.items() if isinstance(x, dict)}, v["edge_px"])
    elif k == "operator": print("operator", round(v["fwd_ms"], 4), round(v["fwd_bwd_ms"], 4))
    elif k.endswith("step_share"): print(k, "%.2f -> %.2f ms (+%.3f = %.2f %%)" % (v["step_ms_without_ssl"], v["step_ms_with_ssl"], v["ssl_ms"], 100 * v["ssl_share"]))
    else: print("extra %-12s %.4f ms %.2f M" % (k, v["ms_per_step"], v["value"] / 1e6))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
