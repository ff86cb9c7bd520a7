#!/bin/bash
# same-box A/B of environment switches on the working tree's build: tools/r4b_env_ab.sh "VAR=a" "VAR=b" ...
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/r4b
for rep in 1 2 3; do
  for e in "$@"; do
    env $e python bench.py --no-cpu-baseline --no-module --no-extra --steps 100 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['roofline'].get('kernel_ms', {})
print('%-34s' % '$e', 'c2 step %.4f' % d['ms_per_step'], ' '.join('%s %.3f' % (n.split('<')[0].split(' (')[0].replace('ssg_', ''), v) for n, v in k.items() if n.startswith('edge') or 'all' in n))"
    for c in c1 b1 c4; do env $e python tools/sparse_step.py $c 200 2>&1 | grep ms/step | sed "s/^/    /"; done
  done
done | tee gpurun_out/r4b/env_ab.txt
