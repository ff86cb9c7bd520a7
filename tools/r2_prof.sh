#!/bin/bash
# rocprofv3 kernel-trace stats of the bench loop -> gpurun_out/prof_<tag>/ ; usage: r2_prof.sh TAG [ENV=VAL ...]
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
tag=$1; shift
for kv in "$@"; do export "$kv"; done
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
rm -rf gpurun_out/prof_$tag; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_$tag" -o bench -- python "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$R/gpurun_out/prof_${tag}_bench.txt" 2>&1
cd "$R"; f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -16 "$f" | cut -c1-200
tail -1 gpurun_out/prof_${tag}_bench.txt | cut -c1-400
# keep only the stats (the trace is large)
find gpurun_out/prof_$tag -name "*kernel_trace.csv" -delete
