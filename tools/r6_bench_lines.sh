#!/bin/bash
# the bench lines of the round-6 evidence set only (tools/r6_final.sh without the rocprofv3 / PMC passes) -> gpurun_out/r6b/
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
O=gpurun_out/r6b; rm -rf $O; mkdir -p $O
python bench.py > $O/r6_bench_c2.json 2> $O/err.txt; echo "c2 rc=$?"
python bench.py --config c5 --steps 10 --warmup 3 > $O/r6_bench_c5.json 2>> $O/err.txt; echo "c5 rc=$?"
python bench.py --config c4 --no-extra > $O/r6_bench_c4.json 2>> $O/err.txt; echo "c4 rc=$?"
python bench.py --config c1 --no-extra --steps 200 --warmup 50 > $O/r6_bench_c1.json 2>> $O/err.txt; echo "c1 rc=$?"
python bench.py --no-ssg-output --steps 30 --warmup 5 --no-cpu-baseline > $O/r6_bench_c2_fused.json 2>> $O/err.txt; echo "c2 fused rc=$?"
python bench.py --no-ssg-output --config c5 --steps 10 --warmup 3 --no-cpu-baseline > $O/r6_bench_c5_fused.json 2>> $O/err.txt; echo "c5 fused rc=$?"
