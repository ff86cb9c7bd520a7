#!/bin/bash
# Round-2 evidence: bench lines (c2, c5), rocprofv3 kernel stats of the same commands, PMC passes -> gpurun_out/r2/
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
R="$PWD"; O="$R/gpurun_out/r2"; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
python bench.py --steps 30 --warmup 5 > "$O/r2_bench_c2.json" 2> "$O/bench_c2.err"; echo "bench c2 rc=$?"
python bench.py --config c5 --steps 10 --warmup 3 > "$O/r2_bench_c5.json" 2> "$O/bench_c5.err"; echo "bench c5 rc=$?"
# the fused step (no SSG output) is a separate metric (B_alg' = (12C+4)HW/N)
python bench.py --no-ssg-output --steps 30 --warmup 5 --no-cpu-baseline > "$O/r2_bench_c2_fused.json" 2>> "$O/bench_c2.err"; echo "bench c2 fused rc=$?"
python bench.py --no-ssg-output --config c5 --steps 10 --warmup 3 --no-cpu-baseline > "$O/r2_bench_c5_fused.json" 2>> "$O/bench_c5.err"; echo "bench c5 fused rc=$?"
cd /tmp
# Kernel durations: with SSG_OVERLAP=0 every launch runs alone on the caller's stream -- these are the durations
# bench.py's roofline line measures (one kernel at a time under the profile mask).  The default build runs the direct
# kernel of a pass on a side stream beside the dense one (k_s <= 25), which stretches both in a trace: that summary is
# kept next to it as *_overlap.csv.
for cfg in c2 c5; do
  SSG_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$cfg" -o bench -- python "$R/bench.py" --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-module > "$O/prof_$cfg.log" 2>&1
  f=$(find "$O/prof_$cfg" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/r2_bench_${cfg}_kernel_stats.csv"
  find "$O/prof_$cfg" -name "*kernel_trace.csv" -delete
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_c2_overlap" -o bench -- python "$R/bench.py" --config c2 --steps 10 --warmup 3 --no-cpu-baseline --no-module > "$O/prof_c2_overlap.log" 2>&1
f=$(find "$O/prof_c2_overlap" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/r2_bench_c2_kernel_stats_overlap.csv"
find "$O/prof_c2_overlap" -name "*kernel_trace.csv" -delete
export SSG_OVERLAP=0   # counters: one kernel at a time
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  for cfg in c2 c5; do
    timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$O/pmc_$cfg/p$i" -o pmc -- python "$R/bench.py" --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-module > "$O/pmc_$cfg.p$i.log" 2>&1
    echo "pmc pass $i $cfg rc=$?"
    find "$O/pmc_$cfg/p$i" -name "*kernel_trace.csv" -delete
  done
done
cd "$R"
python - <<'PY' > gpurun_out/r2/r2_pmc_summary.txt
import csv, glob, collections
for cfg in ("c2", "c5"):
    print("=====", cfg, "(bench.py --config %s --steps 3; per-dispatch means; rocprofv3 --pmc, one pass per counter set)" % cfg)
    for f in sorted(glob.glob('gpurun_out/r2/pmc_%s/p*/pmc_counter_collection.csv' % cfg)):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:72]
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); seen[k].add(r['Dispatch_Id'])
        for k, d in agg.items():
            if 'ssg_' not in k: continue
            n = len(seen[k])
            print(k.replace('void ssg::', ''), ' '.join('%s=%.4g' % (c, v / n) for c, v in sorted(d.items())))
PY
ls gpurun_out/r2 | head -30
python tools/pmc_to_json.py
