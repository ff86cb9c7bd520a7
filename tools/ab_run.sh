#!/bin/bash
# Alternate two library builds (gpurun_ab/<tag>/, or "work" = the working tree's build) under the same bench command on
# ONE box: box-to-box and run-to-run spread is ~1-3 %, more than most kernel changes.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
A="$1"; B="$2"; shift 2
mkdir -p gpurun_ab/work; cp ssl_amd/csrc/libssg_hip*.so gpurun_ab/work/
for rep in 1 2 3; do
  for t in "$A" "$B"; do
    cp gpurun_ab/$t/libssg_hip.so gpurun_ab/$t/libssg_hip_prof.so ssl_amd/csrc/
    python bench.py --no-cpu-baseline --no-module --no-extra "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['roofline'].get('kernel_ms', {})
print('$t', 'step %.4f' % d['ms_per_step'], ' '.join('%s %.3f' % (n.split('<')[0].replace('ssg_', '') + ('+' if 'merged' in n else ''), v) for n, v in k.items() if n.startswith('ssg_') or n.startswith('edge') or 'all' in n))"
  done
done
cp gpurun_ab/work/libssg_hip*.so ssl_amd/csrc/
