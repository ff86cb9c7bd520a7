#!/bin/bash
# Same-box A/B of library builds (gpurun_ab/<tag>/; "work" = the working tree's): three alternations of
#   the headline step (steady-state block, no kernel table) + the per-kernel event times of one bench line
#   tools/r6_ab.sh [-t "<pytest -k expr>"] work sb w3 ...      -> gpurun_out/r6_ab_<tags>.txt
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
tests=""; if [ "$1" = "-t" ]; then tests="$2"; shift 2; fi
mkdir -p gpurun_ab/work gpurun_out; cp ssl_amd/csrc/libssg_hip*.so gpurun_ab/work/
out=gpurun_out/r6_ab_$(echo "$@" | tr ' ' '_').txt; : > $out
for rep in $(seq 1 ${REPS:-3}); do
  for t in "$@"; do
    cp gpurun_ab/$t/libssg_hip.so gpurun_ab/$t/libssg_hip_prof.so ssl_amd/csrc/
    if [ $rep = 1 ] && [ -n "$tests" ] && [ $t != work ]; then
      timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$tests" 2>&1 | tail -1 | sed "s/^/$t tests: /" | tee -a $out
    fi
    c2=$(python bench.py --no-cpu-baseline --no-module --no-extra --no-kernel-table --steps 100 --warmup 20 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | cut -d' ' -f2 | cut -c1-6)
    k=$(python bench.py --no-cpu-baseline --no-module --no-extra --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['roofline'].get('kernel_ms', {})
print(' '.join('%s %.3f' % (n.split('<')[0].replace('ssg_', '') + ('+' if 'merged' in n else ''), v) for n, v in k.items() if n.startswith('ssg_') or n.startswith('edge') or 'all' in n))")
    echo "$t c2 $c2 | $k" | tee -a $out
  done
done
cp gpurun_ab/work/libssg_hip*.so ssl_amd/csrc/
