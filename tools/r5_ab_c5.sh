#!/bin/bash
# Same-box A/B of product-library variants (gpurun_ab/<tag>/libssg_hip.so) on the C5 steps under rocprofv3 stats:
#   tools/r5_ab_c5.sh [mat|fused|both] <tag> <tag> ...    -> top kernels per variant, two alternations
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
what=$1; shift
O=$PWD/gpurun_out/r5ab; mkdir -p $O gpurun_ab/work; cp ssl_amd/csrc/libssg_hip.so gpurun_ab/work/
for rep in 1 2; do
  for t in "$@"; do
    cp gpurun_ab/$t/libssg_hip.so ssl_amd/csrc/
    for m in mat fused; do
      [ $what != both ] && [ $what != $m ] && continue
      fl="--config c5 --no-kernel-table --steps 10 --warmup 3 --no-cpu-baseline --no-module --no-extra"; [ $m = fused ] && fl="$fl --no-ssg-output"
      rm -rf $O/p; rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o p -- python bench.py $fl > $O/$t.json 2> $O/$t.err
      f=$(find $O/p -name "*kernel_stats.csv" | head -1)
      echo "$t $m step $(grep -o '"ms_per_step": [0-9.]*' $O/$t.json | head -1 | cut -d' ' -f2 | cut -c1-6) $(head -4 "$f" | python -c "
import sys, csv
print(' | '.join('%s %.0f' % (r[0].split('<')[0].replace('void ssg::', ''), float(r[3]) / 1e3) for r in csv.reader(sys.stdin) if r[0] != 'Name'))")"
    done
  done
done
cp gpurun_ab/work/libssg_hip.so ssl_amd/csrc/
