#!/bin/bash
# Round-6 evidence: the driver's bench line (with extra: C5, fused), rocprofv3 kernel stats of the same commands, PMC
# passes (SQ x2, FETCH, WRITE, GRBM: one pass each, never together with a trace domain) -> gpurun_out/r6/
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
R="$PWD"; O="$R/gpurun_out/r6"; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
python bench.py > "$O/r6_bench_c2.json" 2> "$O/bench_c2.err"; echo "bench c2 rc=$?"
python bench.py --config c5 --steps 10 --warmup 3 > "$O/r6_bench_c5.json" 2> "$O/bench_c5.err"; echo "bench c5 rc=$?"
python bench.py --config c4 --no-extra > "$O/r6_bench_c4.json" 2> "$O/bench_c4.err"; echo "bench c4 rc=$?"
python bench.py --config c1 --no-extra --steps 200 --warmup 50 > "$O/r6_bench_c1.json" 2> "$O/bench_c1.err"; echo "bench c1 rc=$?"
python bench.py --no-ssg-output --steps 30 --warmup 5 --no-cpu-baseline > "$O/r6_bench_c2_fused.json" 2>> "$O/bench_c2.err"; echo "bench c2 fused rc=$?"
python bench.py --no-ssg-output --config c5 --steps 10 --warmup 3 --no-cpu-baseline > "$O/r6_bench_c5_fused.json" 2>> "$O/bench_c5.err"; echo "bench c5 fused rc=$?"
python tools/sweep.py > "$O/r6_sweep_configs_densities.txt" 2>&1; echo "sweep rc=$?"
python tools/degrade_time.py > "$O/r6_degrade_time.txt" 2>&1; echo "degrade rc=$?"
cd /tmp
# kernel durations with --no-overlap (ssg_set_overlap(0)) (every launch alone on the caller's stream = what bench's masked runs measure);
# the default build's trace (direct kernels beside the dense ones) as *_overlap.csv
# (c5f = the fused C5 step, no SSG output: tile-major scratch rows, ssg_fwd_strip / ssg_rows_tm / ssg_bwd_dense<..., TM>;
#  its stats come from a run WITHOUT bench's per-kernel table, which launches the row-major kernels on their own)
flags() { case $1 in c1) echo "--config c1 --steps 200";; c2) echo "--config c2";; c4) echo "--config c4 --no-kernel-table";; c5) echo "--config c5 --no-kernel-table";; c5f) echo "--config c5 --no-ssg-output --no-kernel-table";; esac; }
for cfg in c1 c2 c4 c5 c5f; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$cfg" -o bench -- python "$R/bench.py" $(flags $cfg) --no-overlap --steps 10 --warmup 3 --no-cpu-baseline --no-module --no-extra > "$O/prof_$cfg.log" 2>&1
  f=$(find "$O/prof_$cfg" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/r6_bench_${cfg}_kernel_stats.csv"
  find "$O/prof_$cfg" -name "*kernel_trace.csv" -delete
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_c2_overlap" -o bench -- python "$R/bench.py" --config c2 --steps 10 --warmup 3 --no-cpu-baseline --no-module --no-extra > "$O/prof_c2_overlap.log" 2>&1
f=$(find "$O/prof_c2_overlap" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/r6_bench_c2_kernel_stats_overlap.csv"
find "$O/prof_c2_overlap" -name "*kernel_trace.csv" -delete
NOOV="--no-overlap"   # counters: one kernel at a time
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  for cfg in c2 c4 c5 c5f; do
    timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$O/pmc_$cfg/p$i" -o pmc -- python "$R/bench.py" $(flags $cfg) $NOOV --steps 3 --warmup 1 --no-cpu-baseline --no-module --no-extra > "$O/pmc_$cfg.p$i.log" 2>&1
    echo "pmc pass $i $cfg rc=$?"
    find "$O/pmc_$cfg/p$i" -name "*kernel_trace.csv" -delete
  done
done
cd "$R"
python - <<'PY' > gpurun_out/r6/r6_pmc_summary.txt
import csv, glob, collections
for cfg in ("c2", "c4", "c5", "c5f"):
    print("=====", cfg, "(bench.py %s --steps 3; per-dispatch means; rocprofv3 --pmc, one pass per counter set)" % {"c2": "--config c2", "c4": "--config c4", "c5": "--config c5", "c5f": "--config c5 --no-ssg-output (fused step, tile-major rows)"}[cfg])
    for f in sorted(glob.glob('gpurun_out/r6/pmc_%s/p*/pmc_counter_collection.csv' % cfg)):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:80]
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); seen[k].add(r['Dispatch_Id'])
        for k, d in agg.items():
            if 'ssg_' not in k: continue
            n = len(seen[k])
            print(k.replace('void ssg::', ''), ' '.join('%s=%.4g' % (c, v / n) for c, v in sorted(d.items())))
PY
cp gpurun_out/r6/r6_pmc_summary.txt profiles/ 2>/dev/null; python tools/pmc_to_json.py gpurun_out/r6/r6_pmc_summary.txt; cp profiles/pmc_traffic.json gpurun_out/r6/
# drop the bulky raw counter csv (the summary and pmc_traffic.json are what is kept)
find "$O" -name "pmc_counter_collection.csv" -delete; find "$O" -name "*agent_info.csv" -delete
rm -rf "$O"/prof_* "$O"/pmc_*/p*/ 2>/dev/null
ls "$O" | head -60
