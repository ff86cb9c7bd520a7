"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/r2_final.sh (gpurun_out/r2/pmc_{c2,c5}/p*/) into
profiles/pmc_traffic.json.

Units and corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): rocprofv3 reports FETCH_SIZE /
WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts 128-byte read requests as 64 bytes for wide coalesced streams (x2);
that factor is uncalibrated for narrow / scattered reads, so the raw figures are stored too and
`hbm_bytes_per_launch` = raw_write + 2 * raw_fetch, an upper bound for the scattered tile fills."""
import collections, csv, glob, json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out", "r2")
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only, SSG_OVERLAP=0) "
                 "over bench.py --config {c2,c5} --steps 3 (tools/r2_final.sh); units and gfx950 corrections per "
                 "MI355X_MICROARCH.md (HBM section): counters are KiB, FETCH_SIZE counts 128-byte read requests as 64 "
                 "bytes for wide coalesced streams (x2; an upper bound for scattered reads), WRITE_SIZE is "
                 "uncalibrated for partial-line stores",
       "kernels": {}}
for cfg in ("c2", "c5"):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, "pmc_" + cfg, "p*", "pmc_counter_collection.csv")):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_LDS_IDX_ACTIVE") and "ssg_" in r["Kernel_Name"]:
                per[(r["Kernel_Name"], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        for (k, _), d in per.items():
            for c, v in d.items():
                vals[k][c].append(v)
    for k, d in vals.items():
        name = k.replace("void ssg::", "").split("(")[0].replace(" ", "")
        fetch = sum(d.get("FETCH_SIZE", [0])) / max(len(d.get("FETCH_SIZE", [1])), 1) * 1024
        write = sum(d.get("WRITE_SIZE", [0])) / max(len(d.get("WRITE_SIZE", [1])), 1) * 1024
        mean = lambda c: sum(d.get(c, [0])) / max(len(d.get(c, [1])), 1)
        out["kernels"][name] = {"config": cfg, "fetch_bytes_raw": fetch, "write_bytes_raw": write,
                                "hbm_bytes_per_launch": write + 2 * fetch,
                                "valu_insts_per_launch": mean("SQ_INSTS_VALU"),          # wave-instructions
                                "lds_active_cycles_per_launch": mean("SQ_LDS_IDX_ACTIVE")}  # summed over the CUs
json.dump(out, open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1)
tot = collections.defaultdict(float)
for k, v in out["kernels"].items():
    tot[v["config"]] += v["hbm_bytes_per_launch"]
print({c: "%.3f GB per step (one launch of each kernel)" % (t / 1e9) for c, t in tot.items()})
