"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/prof_pmc.sh into profiles/pmc_traffic.json.

Units and corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): rocprofv3 reports
FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts 128-byte read requests as 64 bytes for
wide coalesced streams (x2); that factor is uncalibrated for narrow/scattered reads, so both the raw
and the doubled read figures are stored and `hbm_bytes_per_launch` uses raw_write + 2*raw_fetch as the
guide prescribes for streaming reads (an upper bound for the scattered tile fills)."""
import collections, csv, glob, json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out", "pmc")
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "p*", "pmc_counter_collection.csv")):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE") and "ssg_" in r["Kernel_Name"]:
            per[(r["Kernel_Name"], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, _), d in per.items():
        for c, v in d.items():
            vals[k][c].append(v)
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py --steps 3",
       "kernels": {}}
for k, d in vals.items():
    name = k.replace("void ssg::", "").split("(")[0]
    fetch = sum(d.get("FETCH_SIZE", [0])) / max(len(d.get("FETCH_SIZE", [1])), 1) * 1024
    write = sum(d.get("WRITE_SIZE", [0])) / max(len(d.get("WRITE_SIZE", [1])), 1) * 1024
    out["kernels"][name] = {"fetch_bytes_raw": fetch, "write_bytes_raw": write,
                            "hbm_bytes_per_launch": write + 2 * fetch}
json.dump(out, open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
