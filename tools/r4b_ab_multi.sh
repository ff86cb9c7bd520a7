#!/bin/bash
# alternate several library builds (gpurun_ab/<tag>/) on one box: C2 bench line with the per-kernel table + sparse steps
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_ab/work gpurun_out/r4b; cp ssl_amd/csrc/libssg_hip*.so gpurun_ab/work/
for rep in 1 2 3; do
  for t in "$@"; do
    cp gpurun_ab/$t/libssg_hip.so gpurun_ab/$t/libssg_hip_prof.so ssl_amd/csrc/
    python bench.py --no-cpu-baseline --no-module --no-extra --steps 100 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['roofline'].get('kernel_ms', {})
print('%-6s' % '$t', 'c2 %.4f' % d['ms_per_step'], ' '.join('%s %.3f' % (n.split('<')[0].split(' (')[0].replace('ssg_', ''), v) for n, v in k.items()))"
    for c in c4 b4; do python tools/sparse_step.py $c 200 2>&1 | grep ms/step | sed "s/^/       /"; done
  done
done | tee gpurun_out/r4b/ab_multi.txt
cp gpurun_ab/work/libssg_hip*.so ssl_amd/csrc/
