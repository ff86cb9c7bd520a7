#!/bin/bash
# channel-parallel direct forward: parity tests with the variants forced on, then off/auto timings -> gpurun_out/r4cs/
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
O=gpurun_out/r4cs; mkdir -p $O
SSG_FWD_CSPLIT=1 timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 -k "f2_paper or f3 or f5 or f9 or f10 or loss_step or c2_full or dense or kl_cond or gradient_of_one or region_and_list or ref_api or unchanged or drop_in or fused_step" 2>&1 | tail -3
for c in b1 b4 c4; do
  for m in 0 auto 1; do
    if [ $m = auto ]; then python tools/sparse_step.py $c 100; else SSG_FWD_CSPLIT=$m python tools/sparse_step.py $c 100; fi 2>&1 | grep ms/step | sed "s/^/csplit=$m  /"
  done
done | tee $O/times.txt
for m in 0 auto; do
  if [ $m = auto ]; then python bench.py --no-extra --no-cpu-baseline --no-module --steps 100 --warmup 20; else SSG_FWD_CSPLIT=$m python bench.py --no-extra --no-cpu-baseline --no-module --steps 100 --warmup 20; fi 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['roofline']['kernel_ms']
print('c2 csplit=$m step %.4f' % d['ms_per_step'], ' '.join('%s %.3f' % (n.split('<')[0].replace('ssg_', ''), v) for n, v in k.items()))"
done | tee -a $O/times.txt
