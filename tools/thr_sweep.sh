#!/bin/bash
# step time at C2 for several dense thresholds (SSG_DENSE_THR), current kernels
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
for t in "$@"; do
  SSG_DENSE_THR=$t timeout 200 python bench.py --no-cpu-baseline --no-module 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('thr', sys.argv[1], round(d['ms_per_step'],4), {k.split('<')[0]+('m' if 'merged' in k else ''):round(v,3) for k,v in d['roofline']['kernel_ms'].items() if k.startswith('ssg_')})" $t
done
