#!/usr/bin/env python3
"""VGPR / AGPR / SGPR / scratch / static LDS / occupancy of every kernel in ssl_amd/csrc as the compiler reports them
(hipcc -Rpass-analysis=kernel-resource-usage, gfx950, the Makefile's flags) -> profiles/r6_resource_usage.txt.
Runs without a GPU (about a minute).  DESIGN.md section 4's register figures are read off this file.
   python tools/resource_usage.py [out.txt]"""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "ssl_amd", "csrc")
OUT = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r6_resource_usage.txt")
FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function".split()
EXTRA = {"ssg_dense": ["-fno-slp-vectorize"], "ssg_bwd_dense": ["-fno-slp-vectorize"], "ssg_degrade": ["-ffp-contract=off"]}
FILES = ["ssg_fwd", "ssg_dense", "ssg_bwd", "ssg_bwd_dense", "ssg_grow", "ssg_edges", "ssg_datapath", "ssg_degrade"]
KEYS = ["VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "VGPRs Spill", "LDS Size [bytes/block]", "Occupancy [waves/SIMD]"]


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), stdout=subprocess.PIPE, text=True)
    return p.stdout.split("\n")[:len(names)]


rows = []
for f in FILES:
    err = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + EXTRA.get(f, []) + ["-Rpass-analysis=kernel-resource-usage", "-c",
                         f + ".hip", "-o", os.devnull], cwd=SRC, stderr=subprocess.PIPE, text=True).stderr
    cur = None
    for line in err.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = {"file": f, "name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+(.+?): (\S+) \[-Rpass", line)
        if m and cur is not None:
            cur[m.group(1)] = m.group(2)
for r, d in zip(rows, demangle([r["name"] for r in rows])):
    r["pretty"] = re.sub(r"\((ssg::)?\w+Params\)$", "", d.replace("void ", "").replace("ssg::", ""))
ver = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], stdout=subprocess.PIPE, text=True).stdout.splitlines()[0]
with open(OUT, "w") as o:
    o.write(f"# kernel resource usage (hipcc -Rpass-analysis=kernel-resource-usage, gfx950; {ver})\n")
    o.write("# dynamic LDS is requested at launch and not shown here; occupancy is the compiler's register-based figure\n")
    o.write("%-14s %-78s %5s %5s %5s %8s %6s %8s %5s\n" % ("file", "kernel", "VGPR", "AGPR", "SGPR", "scratch", "spill", "LDS(st.)", "occ"))
    for r in rows:
        o.write("%-14s %-78s %5s %5s %5s %8s %6s %8s %5s\n" % ((r["file"], r["pretty"][:78]) + tuple(r.get(k, "?") for k in KEYS)))
print(f"{len(rows)} kernels -> {OUT}")
