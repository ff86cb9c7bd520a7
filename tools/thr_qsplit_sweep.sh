#!/bin/bash
# C2 step time over (dense threshold, waves per tile of the dense backward): SSG_DENSE_THR x SSG_BWD_QSPLIT
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
for q in ${QS:-1 2 3}; do for t in ${THR:-12 16 20 24 28}; do
  r=$(SSG_DENSE_THR=$t SSG_BWD_QSPLIT=$q python bench.py --config c2 --steps 60 --warmup 10 --no-cpu-baseline --no-module --no-extra 2>/dev/null)
  echo "$r" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']
print('thr $t qsplit $q: step %.4f  fwd_dense %.3f fwd_direct %.3f rows %.3f bwd_dense %.3f bwd_direct %.3f' % (d['ms_per_step'], k['ssg_fwd_dense<25,9,3>'], k['ssg_fwd_tiled<25,9> merged+single (2 launches)'], k['ssg_grad_rows<25,9>+finalize'], k['ssg_bwd_dense<25,9,3>'], k['ssg_bwd_tiled<25,9>']))"
done; done
