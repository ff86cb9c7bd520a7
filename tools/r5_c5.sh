#!/bin/bash
# C5 iteration helper: [-k expr] GPU tests touching the k_s = 49 tile-major path, then the materialising and the fused
# C5 step under rocprofv3 --kernel-trace --stats (no side stream at k_s = 49 anyway) -> gpurun_out/r5c5/<tag>_*
#   tools/r5_c5.sh <tag> [notest|testonly] [pytest -k expression]
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
tag=${1:-x}; mode=${2:-all}; kexpr=${3:-"c5 or tile_major or k49 or stress or fused_step"}
O=$PWD/gpurun_out/r5c5; mkdir -p $O
if [ "$mode" != notest ]; then
  timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 900 -k "$kexpr" > $O/${tag}_pytest.txt 2>&1; echo "pytest rc=$?"
  tail -3 $O/${tag}_pytest.txt; grep -n "^FAILED\|^E " $O/${tag}_pytest.txt | head -20
fi
[ "$mode" = testonly ] && exit 0
for m in mat fused; do
  fl="--config c5 --no-kernel-table --steps 10 --warmup 3 --no-cpu-baseline --no-module --no-extra"; [ $m = fused ] && fl="$fl --no-ssg-output"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_$m -o p -- python bench.py $fl > $O/${tag}_$m.json 2> $O/${tag}_$m.err
  f=$(find $O/prof_${tag}_$m -name "*kernel_stats.csv" | head -1)
  echo "== $tag $m: $(grep -o '"ms_per_step": [0-9.]*' $O/${tag}_$m.json | head -1)"
  head -5 "$f" | python -c "
import sys, csv
for r in csv.reader(sys.stdin):
    if r[0] != 'Name': print('   %-70s calls %4s avg %9.1f us' % (r[0][:70], r[1], float(r[3]) / 1e3))"
  cp "$f" $O/${tag}_${m}_kernel_stats.csv; rm -rf $O/prof_${tag}_$m
done
