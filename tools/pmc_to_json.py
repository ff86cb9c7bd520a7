"""profiles/r*_pmc_summary.txt (per-dispatch counter means written by tools/r*_final.sh) -> profiles/pmc_traffic.json:
HBM bytes, issued VALU wave-instructions and LDS-active cycles per launch of every SSG kernel.

    python tools/pmc_to_json.py [profiles/r4_pmc_summary.txt]

Units and corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): rocprofv3 reports FETCH_SIZE /
WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts 128-byte read requests as 64 bytes for wide coalesced streams (x2);
that factor is uncalibrated for narrow / scattered reads, so the raw figures are stored too and
`hbm_bytes_per_launch` = raw_write + 2 * raw_fetch, an upper bound for the scattered tile fills.

`launches_per_step` (round 6): how often a kernel runs in ONE step, from the call counts of the kernel-trace taken beside
the PMC passes (profiles/<round>_bench_<config>_kernel_stats.csv next to the summary: calls / the modal call count of the
configuration's ssg kernels) -- ssg_grad_rows runs once per chain, i.e. twice per C2 step; bench.py's per-step sums
(`roofline.step.traffic`, `valu.issued`) weight every kernel with it."""
import collections, json, os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "profiles", "r6_pmc_summary.txt")
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_* (separate passes, --kernel-trace only, "
                 "--no-overlap) over bench.py --config {c2,c4,c5} --steps 3 (tools/r*_final.sh -> " + os.path.relpath(src, root) +
                 ", per-dispatch means); units and gfx950 corrections per MI355X_MICROARCH.md (HBM section): counters "
                 "are KiB, FETCH_SIZE counts 128-byte read requests as 64 bytes for wide coalesced streams (x2; an upper "
                 "bound for scattered reads), WRITE_SIZE is uncalibrated for partial-line stores",
       "kernels": {}}
cfg, vals = None, collections.defaultdict(dict)
for line in open(src):
    m = re.match(r"===== (c\d\w*)", line)
    if m:
        cfg = m.group(1)
        continue
    if not line.startswith("ssg_"):
        continue
    name = line.split("(")[0].replace(" ", "")
    vals[(cfg, name)].update((k, float(v)) for k, v in re.findall(r"([A-Z_]+)=([0-9.e+-]+)", line))
for (cfg, name), d in vals.items():
    fetch, write = d.get("FETCH_SIZE", 0) * 1024, d.get("WRITE_SIZE", 0) * 1024
    out["kernels"][name if name not in out["kernels"] else name + "@" + cfg] = {"config": cfg, "fetch_bytes_raw": fetch, "write_bytes_raw": write,
                            "hbm_bytes_per_launch": write + 2 * fetch,
                            "valu_insts_per_launch": d.get("SQ_INSTS_VALU", 0.0),            # wave-instructions
                            "lds_active_cycles_per_launch": d.get("SQ_LDS_IDX_ACTIVE", 0.0),   # summed over the CUs
                            "lds_bank_conflict_cycles_per_launch": d.get("SQ_LDS_BANK_CONFLICT", 0.0),
                            "wave_cycles_quad_per_launch": d.get("SQ_WAVE_CYCLES", 0.0),
                            "grbm_gui_active_per_launch": d.get("GRBM_GUI_ACTIVE", 0.0)}
# launches per step from the kernel-trace call counts of the same round
import csv
prefix = os.path.basename(src).split("_")[0]          # "r5" / "r6"
for cfg in sorted({v["config"] for v in out["kernels"].values()}):
    path = os.path.join(os.path.dirname(src), "%s_bench_%s_kernel_stats.csv" % (prefix, cfg))
    if not os.path.exists(path):
        path = os.path.join(root, "profiles", "%s_bench_%s_kernel_stats.csv" % (prefix, cfg))
    if not os.path.exists(path):
        continue
    calls = {r["Name"].replace("void ssg::", "").replace("ssg::", "").split("(")[0].replace(" ", ""): int(r["Calls"])
             for r in csv.DictReader(open(path)) if "ssg" in r["Name"]}
    big = [c for n, c in calls.items() if n.startswith("ssg_")]
    if not big:
        continue
    steps = collections.Counter(big).most_common(1)[0][0]
    for name, v in out["kernels"].items():
        if v["config"] != cfg:
            continue
        n = calls.get(name.split("@")[0].replace("ssg::", ""))
        if n:
            v["launches_per_step"] = round(n / steps, 2) if abs(n / steps - round(n / steps)) > 0.02 else float(round(n / steps))
    out["source"] += "; launches_per_step[%s] = calls / %d (modal call count) of %s" % (cfg, steps, os.path.relpath(path, root))
json.dump(out, open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1)
tot = collections.defaultdict(float)
for k, v in out["kernels"].items():
    tot[v["config"]] += v["hbm_bytes_per_launch"] * v.get("launches_per_step", 1.0)
print({c: "%.3f GB per step (launches per step counted)" % (t / 1e9) for c, t in tot.items()})
