#!/bin/bash
# full GPU suite + the driver's bench line + C1/C4 quick numbers -> gpurun_out/r4c/
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
O=gpurun_out/r4c; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.txt 2>&1; echo "pytest rc=$?"; grep -n "^FAILED\|passed\|failed\|^E " $O/pytest.txt | head -20
python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4c/bench_c2.json"))
print("c2 ms %.4f blocks %s sclk %s module %.4f" % (d["ms_per_step"], [round(x, 4) for x in d["config"]["ms_per_step_blocks"]], d["config"]["sclk_after_timed_region"], d["module"]["ms_per_step"]))
ex = d["extra"]
for k, v in ex.items():
    if k == "ref_api": print("ref_api floor %.3f" % v["caller_floor_ms"], {m: round(x["ms"], 3) for m, x in v.items() if isinstance(x, dict)})
    elif k == "ref_api_dm": print("ref_api_dm", {m: round(x["ms"], 3) for m, x in v.items() if isinstance(x, dict)}, v["edge_px"])
    elif k == "operator": print("operator", round(v["fwd_ms"], 4), round(v["fwd_bwd_ms"], 4))
    else: print("extra %-12s %.4f ms %.2f M" % (k, v["ms_per_step"], v["value"] / 1e6))
PY
for c in c1 b1 c4; do python tools/sparse_step.py $c 100 2>&1 | grep ms/step; done
